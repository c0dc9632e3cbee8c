// Host-side Poseidon for circuit construction: the witness of a PoseidonGate row (plonky2 0.1.4 gates/poseidon.rs
// `PoseidonGenerator::run_once`).  A handful of rows per circuit (the in-circuit public-input hash); the prover's
// hashing runs in the HIP kernels.  Uses the portable arithmetic of csrc/poseidon.hpp (plain C++ outside hipcc).
#include "../csrc/poseidon.hpp"
#include <cstring>
#include "host_internal.hpp"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace lc {

namespace {
// The MDS layer on the host.  The recursive verifier's witness is a few thousand PoseidonGate rows, most of them a sequential
// sponge (the hash of the inner proof's public inputs), so the permutation's latency on one core is what counts: the 288 small
// multiplications of the layer run four to a vpmuludq where AVX2 is present (csrc/poseidon.hpp pos_mds is the portable form).
#if defined(__x86_64__)
__attribute__((target("avx2"))) void mds_avx2(lcp2::u64 s[12]) {  // lcp2::u64 (unsigned long long), NOT uint64_t: the callers' arrays are u64,
  // and a pointer cast between the two distinct 64-bit types lets the optimiser assume the stores below do not alias them
  static const uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  alignas(32) uint32_t lo[28], hi[28];
  for (int i = 0; i < 12; i++) { lo[i] = lo[i + 12] = (uint32_t)s[i]; hi[i] = hi[i + 12] = (uint32_t)(s[i] >> 32); }
  for (int i = 24; i < 28; i++) lo[i] = hi[i] = 0;
  __m256i al[3], ah[3];
  for (int v = 0; v < 3; v++) al[v] = ah[v] = _mm256_setzero_si256();
  for (int i = 0; i < 12; i++) {  // output r (lane r of the three vectors) takes C[i] * s[(i + r) % 12]
    const __m256i c = _mm256_set1_epi64x(C[i]);
    for (int v = 0; v < 3; v++) {
      const __m256i l = _mm256_cvtepu32_epi64(_mm_loadu_si128((const __m128i *)(lo + i + 4 * v)));
      const __m256i h = _mm256_cvtepu32_epi64(_mm_loadu_si128((const __m128i *)(hi + i + 4 * v)));
      al[v] = _mm256_add_epi64(al[v], _mm256_mul_epu32(l, c));
      ah[v] = _mm256_add_epi64(ah[v], _mm256_mul_epu32(h, c));
    }
  }
  alignas(32) uint64_t a[12], b[12];
  for (int v = 0; v < 3; v++) { _mm256_store_si256((__m256i *)(a + 4 * v), al[v]); _mm256_store_si256((__m256i *)(b + 4 * v), ah[v]); }
  a[0] += (uint64_t)lo[0] * 8u; b[0] += (uint64_t)hi[0] * 8u;  // the diagonal
  for (int r = 0; r < 12; r++) {  // a + b 2^32 with a, b < 2^42 (as in pos_mds)
    const uint64_t bh = b[r] >> 32, t = a[r] + ((bh << 32) - bh), u = b[r] << 32, v = t + u;
    s[r] = v < t ? v + 0xFFFFFFFFull : v;
  }
}
const bool HAVE_AVX2 = __builtin_cpu_supports("avx2");
#else
const bool HAVE_AVX2 = false;
void mds_avx2(lcp2::u64 *) {}
#endif
inline void host_mds(lcp2::u64 s[12]) {
  if (HAVE_AVX2) mds_avx2(s); else lcp2::pos_mds(s);
}
}  // namespace

const uint64_t *poseidon_round_constants() {
  static uint64_t rc[lcp2::POS_ROUNDS * lcp2::POS_W];
  static bool ready = false;
  if (!ready) { lcp2::pos_derive_round_constants((lcp2::u64 *)rc); ready = true; }
  return rc;
}

// The naive round form (add constants, S-box, MDS), recording the value that enters every S-box that has a wire.  The value
// entering lane 0's S-box in a partial round is the same in plonky2's fast-partial-round refactoring.
void poseidon_gate_row(const F in[12], bool swap, F row[135]) {
  using namespace lcp2;
  const uint64_t *rc = poseidon_round_constants();
  for (uint32_t i = 0; i < POS_GATE_WIRES; i++) row[i] = 0;
  u64 s[12];
  for (int i = 0; i < 12; i++) { row[POS_WIRE_INPUT + i] = in[i] % GL_P; s[i] = row[POS_WIRE_INPUT + i]; }
  row[POS_WIRE_SWAP] = swap ? 1 : 0;
  for (int i = 0; i < 4; i++) {
    const u64 delta = swap ? gl_sub(s[i + 4], s[i]) : 0;
    row[POS_WIRE_DELTA + i] = delta;
    const u64 l = gl_add(s[i], delta), r = gl_sub(s[i + 4], delta);
    s[i] = l; s[i + 4] = r;
  }
  for (int round = 0; round < POS_ROUNDS; round++) {
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], rc[12 * round + i]);
    const bool full = round < POS_FULL_HALF || round >= POS_FULL_HALF + POS_PARTIAL;
    if (full) {
      for (int i = 0; i < 12; i++) {
        if (round >= 1 && round < POS_FULL_HALF) row[pos_wire_full_sbox_0(round, i)] = s[i];
        if (round >= POS_FULL_HALF + POS_PARTIAL) row[pos_wire_full_sbox_1(round - POS_FULL_HALF - POS_PARTIAL, i)] = s[i];
        s[i] = gl_canon(pos_sbox(s[i]));
      }
    } else {
      row[POS_WIRE_PARTIAL + (round - POS_FULL_HALF)] = s[0];
      s[0] = gl_canon(pos_sbox(s[0]));
    }
    host_mds(s);
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
  }
  for (int i = 0; i < 12; i++) row[POS_WIRE_OUTPUT + i] = s[i];
}


// The outputs of a PoseidonGate row without its trace: swap, then the permutation.  This is what a generator downstream needs; the
// row's 135 cells are produced on the device from (inputs, swap) (lcp2_poseidon_gate_rows).  The recursive verifier's sponge over
// the inner proof's 25 216 public inputs is 3 152 of these in sequence, so the LATENCY of this function on one core is what the
// host side of a light-client proof costs.  Lazily reduced values (any u64 congruent to the element, one canonicalisation at the
// end); every linear layer is a row of 128-bit multiply-accumulates with one reduction per output; the partial rounds run three
// at a time exactly as in the hash kernels (csrc/poseidon.hpp: y3 = A^3 y + ..., 386 small products instead of 3 x 144), which
// also takes two thirds of the S-box -> MDS -> S-box dependency chains out of the critical path.
namespace {
typedef unsigned __int128 u128;
// 128 bits -> a lazy 64-bit value, without branches: the borrow and the carry of the two steps are data dependent coin flips, and
// as branches (what the portable form of csrc/gl64.hpp compiles to) they mispredict half the time - 26 ns per S-box instead of 8
inline lcp2::u64 red128(u128 v) {
  const lcp2::u64 lo = (lcp2::u64)v, hi = (lcp2::u64)(v >> 64), hi_hi = hi >> 32, hi_lo = hi & 0xFFFFFFFFull;
  lcp2::u64 t0, r;
  const bool borrow = __builtin_sub_overflow(lo, hi_hi, &t0);
  t0 -= (0 - (lcp2::u64)borrow) & 0xFFFFFFFFull;
  const bool carry = __builtin_add_overflow(t0, (hi_lo << 32) - hi_lo, &r);
  return r + ((0 - (lcp2::u64)carry) & 0xFFFFFFFFull);
}
inline lcp2::u64 mul_lazy(lcp2::u64 a, lcp2::u64 b) { return red128((u128)a * b); }
inline lcp2::u64 sbox_lazy(lcp2::u64 x) {  // x^7 on lazy values
  const lcp2::u64 x2 = mul_lazy(x, x), x4 = mul_lazy(x2, x2), x3 = mul_lazy(x, x2);
  return mul_lazy(x3, x4);
}
const lcp2::u64 *extended_round_constants() {
  static lcp2::u64 rc[lcp2::POS_RC_WORDS];
  static bool ready = false;
  if (!ready) {
    lcp2::pos_derive_round_constants(rc);
    lcp2::pos_extend_round_constants(rc);
    ready = true;
  }
  return rc;
}
// al + ah 2^32 (al, ah < 2^59) as a lazy 64-bit value (csrc/poseidon.hpp pos_mds)
inline lcp2::u64 fold_halves(lcp2::u64 al, lcp2::u64 ah) {
  const lcp2::u64 ahh = ah >> 32, t = al + ((ahh << 32) - ahh);
  lcp2::u64 v;
  const bool carry = __builtin_add_overflow(t, ah << 32, &v);
  return v + ((0 - (lcp2::u64)carry) & 0xFFFFFFFFull);
}

// ---- portable linear layers: rows of 128-bit multiply-accumulates
void mds_lazy_portable(const lcp2::u64 t[12], lcp2::u64 s[12], const lcp2::u64 *add) {
  static constexpr lcp2::u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  for (int r = 0; r < 12; r++) {
    u128 acc = add ? add[r] : 0;
    for (int i = 0; i < 12; i++) acc += (u128)t[(i + r) % 12] * C[i];
    if (r == 0) acc += (u128)t[0] * 8u;
    s[r] = red128(acc);
  }
}
// three partial rounds (csrc/poseidon.hpp): s = u of the first of them on entry, u of the round after the third on return
void partial3_portable(lcp2::u64 s[12], const lcp2::u64 *k) {
  using namespace lcp2;
  constexpr PosPartialTables T = pos_partial_tables();
  const u64 *y = s + 1;
  const u64 w0 = sbox_lazy(s[0]);
  u128 a1 = (u128)k[0] + (u128)w0 * T.m00, a2 = (u128)k[1] + (u128)w0 * T.ab, a3 = (u128)k[2] + (u128)w0 * T.aAb;
  for (int j = 0; j < 11; j++) { a1 += (u128)y[j] * T.a[j]; a2 += (u128)y[j] * T.aA[j]; a3 += (u128)y[j] * T.aA2[j]; }
  u128 acc[11];
  for (int i = 0; i < 11; i++) {  // the part of y3 that does not wait for the S-boxes below
    u128 v = (u128)k[3 + i] + (u128)w0 * T.A2b[i];
    for (int j = 0; j < 11; j++) v += (u128)y[j] * T.A3[i][j];
    acc[i] = v;
  }
  const u64 w1 = sbox_lazy(red128(a1));
  a2 += (u128)w1 * T.m00;
  const u64 w2 = sbox_lazy(red128(a2));
  a3 += (u128)w1 * T.ab + (u128)w2 * T.m00;
  s[0] = red128(a3);
  for (int i = 0; i < 11; i++) s[1 + i] = red128(acc[i] + (u128)w1 * T.Ab[i] + (u128)w2 * T.b[i]);
}

#if defined(__x86_64__)
// ---- AVX2 linear layers on 32-bit halves, four outputs to a vector: vpmuludq multiplies the low halves of its lanes, so a product
// with the low half needs no masking and one with the high half one shift; both sums of an output stay below 2^59 and are folded
// (al + ah 2^32 mod p) in the vector.  A 128-bit multiply-accumulate costs the scalar core 4 micro-ops per product, this 1/2.
__attribute__((target("avx2"))) inline __m256i fold_halves_avx2(__m256i al, __m256i ah) {
  const __m256i ahh = _mm256_srli_epi64(ah, 32);
  const __m256i t = _mm256_add_epi64(al, _mm256_sub_epi64(_mm256_slli_epi64(ahh, 32), ahh));
  const __m256i v = _mm256_add_epi64(t, _mm256_slli_epi64(ah, 32));
  const __m256i sign = _mm256_set1_epi64x((long long)0x8000000000000000ull);
  const __m256i carry = _mm256_cmpgt_epi64(_mm256_xor_si256(t, sign), _mm256_xor_si256(v, sign));  // v < t (unsigned)
  return _mm256_add_epi64(v, _mm256_and_si256(carry, _mm256_set1_epi64x(0xFFFFFFFFll)));
}
__attribute__((target("avx2"))) void mds_lazy_avx2(const lcp2::u64 t[12], lcp2::u64 s[12], const lcp2::u64 *add) {
  static const lcp2::u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  alignas(32) lcp2::u64 buf[24];
  for (int i = 0; i < 12; i++) buf[i] = buf[i + 12] = t[i];
  const __m256i mask = _mm256_set1_epi64x(0xFFFFFFFFll);
  __m256i al[3], ah[3];
  for (int v = 0; v < 3; v++) {
    if (add) {
      const __m256i k = _mm256_loadu_si256((const __m256i *)(add + 4 * v));
      al[v] = _mm256_and_si256(k, mask);
      ah[v] = _mm256_srli_epi64(k, 32);
    } else {
      al[v] = ah[v] = _mm256_setzero_si256();
    }
  }
  {  // the diagonal: 8 t[0] into output 0
    const __m256i d = _mm256_set_epi64x(0, 0, 0, (long long)t[0]), eight = _mm256_set1_epi64x(8);
    al[0] = _mm256_add_epi64(al[0], _mm256_mul_epu32(d, eight));
    ah[0] = _mm256_add_epi64(ah[0], _mm256_mul_epu32(_mm256_srli_epi64(d, 32), eight));
  }
  for (int i = 0; i < 12; i++) {  // output r (lane r of the three vectors) takes C[i] t[(i + r) % 12]
    const __m256i c = _mm256_set1_epi64x(C[i]);
    for (int v = 0; v < 3; v++) {
      const __m256i x = _mm256_loadu_si256((const __m256i *)(buf + i + 4 * v));
      al[v] = _mm256_add_epi64(al[v], _mm256_mul_epu32(x, c));
      ah[v] = _mm256_add_epi64(ah[v], _mm256_mul_epu32(_mm256_srli_epi64(x, 32), c));
    }
  }
  for (int v = 0; v < 3; v++) _mm256_storeu_si256((__m256i *)(s + 4 * v), fold_halves_avx2(al[v], ah[v]));
}
// constant vectors of the grouped partial rounds: [j][v] = entries (4v .. 4v+3, j); v = 0..2 the 11 rows of y3 (lane 11 zero),
// v = 3 the three rows (a, aA, aA^2) of u1, u2, u3
struct Partial3Vectors {
  alignas(32) lcp2::u64 y[11][4][4];
  alignas(32) lcp2::u64 w0[4][4], w1[3][4], w2[3][4];
  Partial3Vectors() {
    constexpr lcp2::PosPartialTables T = lcp2::pos_partial_tables();
    memset(this, 0, sizeof *this);
    for (int i = 0; i < 11; i++) {
      for (int j = 0; j < 11; j++) y[j][i / 4][i % 4] = T.A3[i][j];
      w0[i / 4][i % 4] = T.A2b[i]; w1[i / 4][i % 4] = T.Ab[i]; w2[i / 4][i % 4] = T.b[i];
    }
    for (int j = 0; j < 11; j++) { y[j][3][0] = T.a[j]; y[j][3][1] = T.aA[j]; y[j][3][2] = T.aA2[j]; }
    w0[3][0] = T.m00; w0[3][1] = T.ab; w0[3][2] = T.aAb;
  }
};
__attribute__((target("avx2"))) void partial3_avx2(lcp2::u64 s[12], const lcp2::u64 *k) {
  using namespace lcp2;
  constexpr PosPartialTables T = pos_partial_tables();
  static const Partial3Vectors P;
  const __m256i mask = _mm256_set1_epi64x(0xFFFFFFFFll);
  const u64 w0 = sbox_lazy(s[0]);
  alignas(32) u64 kk[16] = {k[3], k[4], k[5], k[6], k[7], k[8], k[9], k[10], k[11], k[12], k[13], 0, k[0], k[1], k[2], 0};
  __m256i al[4], ah[4];
  const __m256i w0v = _mm256_set1_epi64x((long long)w0), w0h = _mm256_srli_epi64(w0v, 32);
  for (int v = 0; v < 4; v++) {
    const __m256i c = _mm256_load_si256((const __m256i *)(kk + 4 * v)), m = _mm256_load_si256((const __m256i *)P.w0[v]);
    al[v] = _mm256_add_epi64(_mm256_and_si256(c, mask), _mm256_mul_epu32(w0v, m));
    ah[v] = _mm256_add_epi64(_mm256_srli_epi64(c, 32), _mm256_mul_epu32(w0h, m));
  }
  for (int j = 0; j < 11; j++) {
    const __m256i yv = _mm256_set1_epi64x((long long)s[1 + j]), yh = _mm256_srli_epi64(yv, 32);
    for (int v = 0; v < 4; v++) {
      const __m256i m = _mm256_load_si256((const __m256i *)P.y[j][v]);
      al[v] = _mm256_add_epi64(al[v], _mm256_mul_epu32(yv, m));
      ah[v] = _mm256_add_epi64(ah[v], _mm256_mul_epu32(yh, m));
    }
  }
  alignas(32) u64 ul[4], uh[4];  // the sums of u1, u2, u3 so far
  _mm256_store_si256((__m256i *)ul, al[3]);
  _mm256_store_si256((__m256i *)uh, ah[3]);
  const u64 w1 = sbox_lazy(fold_halves(ul[0], uh[0]));
  const u64 w1l = (u32)w1, w1h = w1 >> 32;
  const u64 w2 = sbox_lazy(fold_halves(ul[1] + w1l * T.m00, uh[1] + w1h * T.m00));
  const u64 w2l = (u32)w2, w2h = w2 >> 32;
  s[0] = fold_halves(ul[2] + w1l * T.ab + w2l * T.m00, uh[2] + w1h * T.ab + w2h * T.m00);
  const __m256i w1v = _mm256_set1_epi64x((long long)w1), w1hv = _mm256_srli_epi64(w1v, 32);
  const __m256i w2v = _mm256_set1_epi64x((long long)w2), w2hv = _mm256_srli_epi64(w2v, 32);
  alignas(32) u64 out[12];
  for (int v = 0; v < 3; v++) {
    const __m256i m1 = _mm256_load_si256((const __m256i *)P.w1[v]), m2 = _mm256_load_si256((const __m256i *)P.w2[v]);
    const __m256i l = _mm256_add_epi64(al[v], _mm256_add_epi64(_mm256_mul_epu32(w1v, m1), _mm256_mul_epu32(w2v, m2)));
    const __m256i h = _mm256_add_epi64(ah[v], _mm256_add_epi64(_mm256_mul_epu32(w1hv, m1), _mm256_mul_epu32(w2hv, m2)));
    _mm256_store_si256((__m256i *)(out + 4 * v), fold_halves_avx2(l, h));
  }
  for (int i = 0; i < 11; i++) s[1 + i] = out[i];
}
#else
void mds_lazy_avx2(const lcp2::u64 *, lcp2::u64 *, const lcp2::u64 *) {}
void partial3_avx2(lcp2::u64 *, const lcp2::u64 *) {}
#endif
inline void mds_lazy(const lcp2::u64 t[12], lcp2::u64 s[12], const lcp2::u64 *add) {
  if (HAVE_AVX2) mds_lazy_avx2(t, s, add); else mds_lazy_portable(t, s, add);
}
inline void partial3(lcp2::u64 s[12], const lcp2::u64 *k) {
  if (HAVE_AVX2) partial3_avx2(s, k); else partial3_portable(s, k);
}
}  // namespace

// for the tests: the same function on the portable path
void poseidon_gate_outputs_impl(const F in[12], bool swap, F out[12], bool portable);
void poseidon_gate_outputs_impl(const F in[12], bool swap, F out[12], bool portable) {
  using namespace lcp2;
  const u64 *rc = extended_round_constants();
  const bool vec = HAVE_AVX2 && !portable;
  u64 s[12], t[12];
  for (int i = 0; i < 12; i++) s[i] = gl_add_nc(in[i], rc[i]);  // u of round 0: the state after its constant layer
  if (swap)
    for (int i = 0; i < 4; i++) {  // (a + c_i, b + c_j) swapped means (b + c_i, a + c_j)
      const u64 a = gl_add_nc(in[i + 4], rc[i]), b = gl_add_nc(in[i], rc[i + 4]);
      s[i] = a; s[i + 4] = b;
    }
  int round = 0;
  for (int r = 0; r < POS_FULL_HALF; r++, round++) {
    for (int i = 0; i < 12; i++) t[i] = sbox_lazy(s[i]);
    if (vec) mds_lazy_avx2(t, s, rc + 12 * (round + 1)); else mds_lazy_portable(t, s, rc + 12 * (round + 1));
  }
  for (int g = 0; g < POS_GROUPS; g++, round += POS_GROUP) {
    const u64 *k = rc + POS_ROUNDS * POS_W + POS_GROUP_CONSTS * g;
    if (vec) partial3_avx2(s, k); else partial3_portable(s, k);
  }
  for (; round < POS_FULL_HALF + POS_PARTIAL; round++) {  // the partial round the groups leave over
    for (int i = 0; i < 12; i++) t[i] = s[i];
    t[0] = sbox_lazy(s[0]);
    if (vec) mds_lazy_avx2(t, s, rc + 12 * (round + 1)); else mds_lazy_portable(t, s, rc + 12 * (round + 1));
  }
  for (int r = 0; r < POS_FULL_HALF; r++, round++) {
    for (int i = 0; i < 12; i++) t[i] = sbox_lazy(s[i]);
    const u64 *add = round + 1 < POS_ROUNDS ? rc + 12 * (round + 1) : nullptr;
    if (vec) mds_lazy_avx2(t, s, add); else mds_lazy_portable(t, s, add);
  }
  for (int i = 0; i < 12; i++) out[i] = gl_canon(s[i]);
}
void poseidon_gate_outputs(const F in[12], bool swap, F out[12]) { poseidon_gate_outputs_impl(in, swap, out, false); }

}  // namespace lc
