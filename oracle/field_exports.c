/* ORACLE (test infrastructure, NOT product code): ctypes-visible wrappers of gl64.h */
#include "oracle.h"
uint64_t orc_gl_mul(uint64_t a, uint64_t b) { return gl_mul(gl_canon(a), gl_canon(b)); }
uint64_t orc_gl_add(uint64_t a, uint64_t b) { return gl_add(gl_canon(a), gl_canon(b)); }
uint64_t orc_gl_sub(uint64_t a, uint64_t b) { return gl_sub(gl_canon(a), gl_canon(b)); }
uint64_t orc_gl_inv(uint64_t a) { return gl_inv(gl_canon(a)); }
uint64_t orc_gl_pow(uint64_t a, uint64_t e) { return gl_pow(gl_canon(a), e); }
uint64_t orc_gl_root_of_unity(unsigned k) { return gl_root_of_unity(k); }
void orc_gl2_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
  gl2_t r = gl2_mul(gl2_make(gl_canon(a[0]), gl_canon(a[1])), gl2_make(gl_canon(b[0]), gl_canon(b[1])));
  out[0] = r.c[0]; out[1] = r.c[1];
}
void orc_gl2_inv(const uint64_t a[2], uint64_t out[2]) {
  gl2_t r = gl2_inv(gl2_make(gl_canon(a[0]), gl_canon(a[1])));
  out[0] = r.c[0]; out[1] = r.c[1];
}
int orc_merkle_cap(const uint64_t *leaves, size_t nleaves, size_t leaf_len, unsigned cap_height, uint64_t *cap_out) {
  orc_merkle *t = orc_merkle_build(leaves, nleaves, leaf_len, cap_height);
  if (!t) return -1;
  for (size_t i = 0; i < ((size_t)4 << cap_height); i++) cap_out[i] = t->cap[i];
  orc_merkle_free(t);
  return 0;
}
