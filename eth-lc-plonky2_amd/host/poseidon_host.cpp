// Host-side Poseidon for circuit construction: the witness of a PoseidonGate row (plonky2 0.1.4 gates/poseidon.rs
// `PoseidonGenerator::run_once`).  A handful of rows per circuit (the in-circuit public-input hash); the prover's
// hashing runs in the HIP kernels.  Uses the portable arithmetic of csrc/poseidon.hpp (plain C++ outside hipcc).
#include "../csrc/poseidon.hpp"
#include "host_internal.hpp"

namespace lc {

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
    pos_mds(s);
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
  }
  for (int i = 0; i < 12; i++) row[POS_WIRE_OUTPUT + i] = s[i];
}

}  // namespace lc
