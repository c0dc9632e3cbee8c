// Host-side Poseidon for circuit construction: the witness of a PoseidonGate row (plonky2 0.1.4 gates/poseidon.rs
// `PoseidonGenerator::run_once`).  A handful of rows per circuit (the in-circuit public-input hash); the prover's
// hashing runs in the HIP kernels.  Uses the portable arithmetic of csrc/poseidon.hpp (plain C++ outside hipcc).
#include "../csrc/poseidon.hpp"
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
__attribute__((target("avx2"))) void mds_avx2(uint64_t s[12]) {
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
void mds_avx2(uint64_t *) {}
#endif
inline void host_mds(uint64_t s[12]) {
  if (HAVE_AVX2) mds_avx2(s); else lcp2::pos_mds((lcp2::u64 *)s);
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
    host_mds((uint64_t *)s);
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
  }
  for (int i = 0; i < 12; i++) row[POS_WIRE_OUTPUT + i] = s[i];
}

}  // namespace lc
