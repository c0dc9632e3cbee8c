/* ORACLE (test infrastructure, NOT product code).
 *
 * Restates plonky2_field 0.1.1 fft.rs / polynomial/mod.rs conventions
 * (pin /root/reference/Cargo.lock:2425-2427):
 *   fft        natural in -> natural out, v_i = sum_j c_j w^(ij)
 *   ifft       exact inverse (includes 1/n)
 *   lde(k)     zero-pad coefficients to n << k
 *   coset_fft  c_j <- c_j * shift^j, then fft
 *   coset_ifft ifft, then c_j <- c_j * shift^-j
 * Textbook bit-reverse + radix-2 DIT, one column at a time (the device path
 * uses a different decomposition; equality of results is the parity check).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static void bitrev_inplace(uint64_t *a, size_t n, unsigned lg) {
  for (size_t i = 0; i < n; i++) {
    size_t j = gl_bitrev(i, lg);
    if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
  }
}

static void fft_core(uint64_t *a, size_t n, uint64_t root) {
  unsigned lg = gl_log2(n);
  bitrev_inplace(a, n, lg);
  uint64_t *tw = (uint64_t *)malloc((n / 2 ? n / 2 : 1) * sizeof(uint64_t));
  tw[0] = 1;
  for (size_t i = 1; i < n / 2; i++) tw[i] = gl_mul(tw[i - 1], root);
  for (size_t len = 2; len <= n; len <<= 1) {
    size_t half = len / 2, step = n / len;
    for (size_t s = 0; s < n; s += len)
      for (size_t k = 0; k < half; k++) {
        uint64_t u = a[s + k], v = gl_mul(a[s + k + half], tw[k * step]);
        a[s + k] = gl_add(u, v);
        a[s + k + half] = gl_sub(u, v);
      }
  }
  free(tw);
}

void orc_fft(uint64_t *a, size_t n) {
  for (size_t i = 0; i < n; i++) a[i] = gl_canon(a[i]);
  if (n > 1) fft_core(a, n, gl_root_of_unity(gl_log2(n)));
}

void orc_ifft(uint64_t *a, size_t n) {
  for (size_t i = 0; i < n; i++) a[i] = gl_canon(a[i]);
  if (n > 1) fft_core(a, n, gl_inv(gl_root_of_unity(gl_log2(n))));
  uint64_t ninv = gl_inv((uint64_t)n % GL_P);
  for (size_t i = 0; i < n; i++) a[i] = gl_mul(a[i], ninv);
}

void orc_coset_fft(uint64_t *a, size_t n, uint64_t shift) {
  uint64_t s = 1;
  shift = gl_canon(shift);
  for (size_t i = 0; i < n; i++) { a[i] = gl_mul(gl_canon(a[i]), s); s = gl_mul(s, shift); }
  orc_fft(a, n);
}

void orc_coset_ifft(uint64_t *a, size_t n, uint64_t shift) {
  orc_ifft(a, n);
  uint64_t si = gl_inv(gl_canon(shift)), s = 1;
  for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], s); s = gl_mul(s, si); }
}

/* coeffs[n] -> values on shift*<w_{n<<rate_bits}>, natural order (lde + coset_fft) */
void orc_lde_coset_fft(const uint64_t *coeffs, size_t n, unsigned rate_bits, uint64_t shift, uint64_t *out) {
  size_t N = n << rate_bits;
  memcpy(out, coeffs, n * sizeof(uint64_t));
  memset(out + n, 0, (N - n) * sizeof(uint64_t));
  orc_coset_fft(out, N, shift);
}

/* batched column helpers (OpenMP over columns = rayon over polynomials) */
void orc_ifft_batch(uint64_t *cols, size_t ncols, size_t n) {
#pragma omp parallel for schedule(dynamic)
  for (size_t c = 0; c < ncols; c++) orc_ifft(cols + c * n, n);
}
void orc_fft_batch(uint64_t *cols, size_t ncols, size_t n) {
#pragma omp parallel for schedule(dynamic)
  for (size_t c = 0; c < ncols; c++) orc_fft(cols + c * n, n);
}
void orc_lde_batch(const uint64_t *coeffs, size_t ncols, size_t n, unsigned rate_bits, uint64_t shift, uint64_t *out) {
#pragma omp parallel for schedule(dynamic)
  for (size_t c = 0; c < ncols; c++) orc_lde_coset_fft(coeffs + c * n, n, rate_bits, shift, out + c * (n << rate_bits));
}
