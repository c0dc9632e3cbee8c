/* ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of plonky2_field 0.1.1 `GoldilocksField` and its quadratic
 * extension (pinned: Electron-Labs/plonky2 @666f3151, /root/reference/Cargo.lock:2425-2427;
 * reached from the reference through `F = <C as GenericConfig<D>>::F`,
 * /root/reference/eth-lc-plonky2/src/main.rs:74-76).  The crate source is not
 * in this container; the arithmetic below is the published definition
 *  and is pinned numerically by tests/test_oracle_field.py
 * (reduction vs. 128-bit `%`, root-of-unity constant, Poseidon KATs that
 * exercise every operation).
 *
 * Convention inside the oracle: every value held in a uint64_t is CANONICAL
 * (< p).  Entry points canonicalise their inputs with gl_canon().
 */
#ifndef ORACLE_GL64_H
#define ORACLE_GL64_H
#include <stdint.h>
#include <stddef.h>

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL /* 2^32 - 1 = 2^64 mod p */
#define GL_GENERATOR 7ULL    /* multiplicative generator and coset shift */
#define GL_TWO_ADICITY 32
#define GL_ROOT_2_32 1753635133440165772ULL /* 7^((p-1)/2^32) */
#define GL_EXT_W 7ULL /* F_{p^2} = F_p[X]/(X^2 - 7) */

typedef unsigned __int128 gl_u128;

static inline uint64_t gl_canon(uint64_t x) { return x >= GL_P ? x - GL_P : x; }

static inline uint64_t gl_add(uint64_t a, uint64_t b) {
  uint64_t s = a + b;
  if (s < a || s >= GL_P) s -= GL_P;
  return s;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) {
  uint64_t d = a - b;
  if (a < b) d += GL_P;
  return d;
}
static inline uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }

/* x = lo + 2^64*hi_lo + 2^96*hi_hi ; 2^64 = EPS, 2^96 = -1 (mod p). */
static inline uint64_t gl_reduce128(gl_u128 x) {
  uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  uint64_t t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GL_EPS;
  uint64_t t1 = hi_lo * GL_EPS;
  uint64_t r = t0 + t1;
  if (r < t0) r += GL_EPS;
  return gl_canon(r);
}
static inline uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_reduce128((gl_u128)a * b); }
static inline uint64_t gl_sqr(uint64_t a) { return gl_mul(a, a); }

static inline uint64_t gl_pow(uint64_t b, uint64_t e) {
  uint64_t r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_sqr(b);
    e >>= 1;
  }
  return r;
}
static inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }

/* primitive 2^k-th root of unity: ROOT_2_32 ^ (2^(32-k)) */
static inline uint64_t gl_root_of_unity(unsigned k) {
  uint64_t r = GL_ROOT_2_32;
  for (unsigned i = k; i < GL_TWO_ADICITY; i++) r = gl_sqr(r);
  return r;
}

/* ---- quadratic extension, element = c[0] + c[1]*X ---- */
typedef struct { uint64_t c[2]; } gl2_t;

static inline gl2_t gl2_make(uint64_t a, uint64_t b) { gl2_t r = {{a, b}}; return r; }
static inline gl2_t gl2_from_base(uint64_t a) { return gl2_make(a, 0); }
static inline gl2_t gl2_add(gl2_t a, gl2_t b) { return gl2_make(gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1])); }
static inline gl2_t gl2_sub(gl2_t a, gl2_t b) { return gl2_make(gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1])); }
static inline gl2_t gl2_neg(gl2_t a) { return gl2_make(gl_neg(a.c[0]), gl_neg(a.c[1])); }
static inline gl2_t gl2_mul(gl2_t a, gl2_t b) {
  uint64_t c0 = gl_add(gl_mul(a.c[0], b.c[0]), gl_mul(GL_EXT_W, gl_mul(a.c[1], b.c[1])));
  uint64_t c1 = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
  return gl2_make(c0, c1);
}
static inline gl2_t gl2_scale(gl2_t a, uint64_t s) { return gl2_make(gl_mul(a.c[0], s), gl_mul(a.c[1], s)); }
static inline int gl2_eq(gl2_t a, gl2_t b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1]; }
static inline gl2_t gl2_inv(gl2_t a) {
  /* 1/(a0 + a1 X) = (a0 - a1 X) / (a0^2 - 7 a1^2) */
  uint64_t n = gl_sub(gl_sqr(a.c[0]), gl_mul(GL_EXT_W, gl_sqr(a.c[1])));
  uint64_t ni = gl_inv(n);
  return gl2_make(gl_mul(a.c[0], ni), gl_mul(gl_neg(a.c[1]), ni));
}
static inline gl2_t gl2_pow(gl2_t b, uint64_t e) {
  gl2_t r = gl2_from_base(1);
  while (e) {
    if (e & 1) r = gl2_mul(r, b);
    b = gl2_mul(b, b);
    e >>= 1;
  }
  return r;
}

static inline unsigned gl_log2(size_t n) { unsigned k = 0; while (((size_t)1 << k) < n) k++; return k; }
static inline size_t gl_bitrev(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}
#endif
