// Internal declarations of the host circuit layer.
#pragma once
#include <algorithm>
#include <cstring>
#include <functional>
#include "lc_plonky2.hpp"

namespace lc {

constexpr int SHA_ADD_OPS = 3;
constexpr uint32_t ARITH_OPS = 20;
constexpr uint32_t SHA_ROWS_PER_HASH = 48 + 128 + 3 + 128 + 3;  // 310

struct GateSetLayout {
  std::vector<std::pair<uint32_t, uint32_t>> groups;
  uint32_t num_selectors = 0, num_regs = 1;
  std::vector<lcp2_gate> gates;
  std::vector<uint32_t> code;
  std::vector<uint64_t> imm;
};
GateSetLayout build_gate_set(uint32_t max_degree);

// PoseidonGate wire layout (plonky2 gates/poseidon.rs) and the host-side permutation that fills a row (poseidon_host.cpp)
constexpr uint32_t POS_WIRE_INPUT = 0, POS_WIRE_OUTPUT = 12, POS_WIRE_SWAP = 24, POS_WIRE_DELTA = 25, POS_WIRE_FULL_0 = 29,
                   POS_WIRE_PARTIAL = 65, POS_WIRE_FULL_1 = 87, POS_GATE_WIRES = 135;
inline uint32_t pos_wire_full_sbox_0(uint32_t round, uint32_t i) { return POS_WIRE_FULL_0 + 12 * (round - 1) + i; }  // rounds 1..3
inline uint32_t pos_wire_full_sbox_1(uint32_t round, uint32_t i) { return POS_WIRE_FULL_1 + 12 * round + i; }
const uint64_t *poseidon_round_constants();                    // 360 values
void poseidon_gate_row(const F in[12], bool swap, F row[135]); // PoseidonGenerator::run_once: every wire of one row
void poseidon_gate_outputs(const F in[12], bool swap, F out[12]); // the 12 output wires alone (the row itself is generated on the device)
void poseidon_gate_outputs_impl(const F in[12], bool swap, F out[12], bool portable);  // the same, optionally without AVX2 (tests)
extern const uint32_t GATE_DEGREE[G_COUNT];

// ---- Goldilocks on the host (circuit construction / witness generation only)
inline F f_add(F a, F b) { F s = a + b; return (s < a || s >= GOLDILOCKS_P) ? s - GOLDILOCKS_P : s; }
inline F f_mul(F a, F b) {  // 2^64 = 2^32 - 1 and 2^96 = -1 (mod p): no 128-bit division
  const unsigned __int128 m = (unsigned __int128)a * b;
  const uint64_t lo = (uint64_t)m, hi = (uint64_t)(m >> 64), hi_hi = hi >> 32, hi_lo = hi & 0xFFFFFFFFull;
  uint64_t t = lo - hi_hi;
  if (lo < hi_hi) t -= 0xFFFFFFFFull;
  const uint64_t u = (hi_lo << 32) - hi_lo;
  uint64_t r = t + u;
  if (r < t) r += 0xFFFFFFFFull;
  return r >= GOLDILOCKS_P ? r - GOLDILOCKS_P : r;
}
inline F f_pow(F b, uint64_t e) { F r = 1; while (e) { if (e & 1) r = f_mul(r, b); b = f_mul(b, b); e >>= 1; } return r; }
inline F f_root_of_unity(unsigned bits) { return f_pow(f_pow(7, (GOLDILOCKS_P - 1) >> 32), 1ull << (32 - bits)); }

extern const uint32_t SHA_K[64];
extern const uint32_t SHA_IV[8];

// one generator step, evaluated in creation order by generate_witness
struct Op {
  enum Kind { CONST, ARITH, SHA, BITS, POSEIDON, INV, EXT_INV, SPLIT32, HINT } kind;
  // BITS: x -> its c0 low bits in `internal` (split_le), fails if x does not fit
  // POSEIDON: in[0..12), swap flag x -> internal[0..12) on row first_row
  // INV: x -> out = 1 / x (0 for 0) ; EXT_INV: (x, y) -> internal[0..2) ; SPLIT32: x -> internal = {low 32 bits, high 32 bits}
  // HINT: hint_in -> internal through hint_fn (a generator without constraints of its own, e.g. the quotient and remainder of a
  //       BigUint division; the gadget that adds it constrains the results)
  uint32_t out = 0, x = 0, y = 0, z = 0;  // CONST: out ; ARITH: x, y, z -> out
  F c0 = 0, c1 = 0;                        // CONST: c0 = value
  // SHA: message words in[16] -> digest out8[8]; internal words by row
  std::array<uint32_t, 16> in{};
  std::array<uint32_t, 8> out8{};
  uint32_t first_row = 0;
  std::vector<uint32_t> hint_in;
  std::function<void(const std::vector<F> &, std::vector<F> &)> hint_fn;
  std::vector<uint32_t> internal;          // vars: sched W[16..64) (48), then per compression c, per round t: t1, a_new, e_new (2*64*3), then mid[8]
};

struct CellBinding { uint32_t row, col, var; };
// which thread ("lane") runs which host generator in generate_witness_gpu (builder.cpp plan_host_lanes)
struct HostLanes {
  std::vector<uint8_t> lane_of_op;  // index into Impl::ops
  unsigned lanes = 1;
};

// What generate_witness_gpu learns at the first proof of a circuit and reuses afterwards (the structure of a witness generation -
// which hashes are ready together, where their message words come from, which bound cells the host writes - does not depend on
// the witness values; plonky2 builds the same kind of tables in build()):
struct ShaBatchPlan {
  std::vector<lcp2_sha_job> jobs;            // in device order (by level); in_src >= 0: index into the batch's word list
  std::vector<uint32_t> word_vars;           // variable (root) whose value is word k of the batch
  std::vector<uint32_t> level_start;
  std::vector<std::array<uint32_t, 8>> out;  // digest variables of job k
  std::vector<uint32_t> ops_left;            // the SHA generators still waiting after this batch (indices into Impl::ops)
};
struct GpuWitnessPlan {
  bool ready = false;
  std::vector<ShaBatchPlan> batches;
  std::vector<CellBinding> host_cells;       // the bound cells of the rows the host writes cell by cell (var = root)
};

struct CircuitData::Impl {
  CircuitConfig config;
  uint32_t nrows = 0;                     // used rows (before padding)
  std::vector<uint32_t> gate_of_row;
  std::vector<std::array<F, 2>> row_consts;
  std::vector<CellBinding> cells;
  std::vector<uint32_t> parent;           // union-find over variables
  std::vector<Op> ops;
  std::vector<uint32_t> public_inputs;    // variables
  lcp2_ctx *ctx = nullptr;
  void *d_wires = nullptr;                // device witness matrix [num_wires][n], zeroed once
  lcp2_circuit *gpu = nullptr;
  lcp2_circuit *verifier = nullptr;
  HostLanes host_lanes;                   // planned at the first prove
  GpuWitnessPlan gpu_plan;
  // the value table of a witness generation (builder.cpp Values), kept between proofs: a slot is valid when its stamp carries the
  // current epoch, so a new proof costs one increment instead of clearing 9 bytes per variable (3.6 M variables in the 2^22-row
  // circuit: ~3 ms).  One witness generation at a time per CircuitData (prove() is not const).
  // the host-written cells of a proof as the list lcp2_scatter_cells takes: rows and columns filled once (GpuWitnessPlan::host_cells),
  // the values refreshed per proof; pinned (lcp2_host_register) so that its upload is a DMA
  std::vector<lcp2_cell> cell_buf;
  bool cell_buf_pinned = false;
  mutable std::vector<F> value_store;
  mutable std::vector<uint32_t> stamp_store;
  mutable uint32_t epoch = 0;
  uint32_t find(uint32_t v) const { while (parent[v] != v) v = parent[v]; return v; }
};

}  // namespace lc
