// Host-side mirror of the reference's circuit-building surface, in C++ (the image has no Rust toolchain).
//
// Mirrors, name for name, what eth-lc-plonky2 uses from plonky2 / plonky2_crypto and what it defines itself:
//   CircuitBuilder::{new, add_virtual_target, add_virtual_target_arr, constant, mul_add, connect,
//                    register_public_input, num_gates, print_gate_counts, build}     (plonky2 plonk/circuit_builder.rs)
//   CircuitBuilderHash / CircuitBuilderHashSha2::{add_virtual_hash256_target, connect_hash256, two_to_one_sha256},
//   CircuitBuilderU32::{constant_u32, zero_u32, connect_u32}, WitnessHash::set_hash256_target   (plonky2_crypto)
//   PartialWitness::{set_target, set_target_arr}, CircuitData::{prove, verify}       (src/main.rs:226-233)
//   the gadgets of src/merkle_tree_gadget.rs, src/sync_committee_pubkeys.rs and src/targets.rs (gadgets.hpp)
// Differences that the C ABI forces are documented in DESIGN.md: gates are gate programs, the SHA-256 gate
// layout is this repository's own (plonky2_crypto's is not visible).  Public inputs are bound as plonky2 binds them:
// build() hashes them in-circuit with PoseidonGate rows and connects the digest to the PublicInputGate.
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/lcp2.h"

namespace lc {

using F = uint64_t;  // GoldilocksField, canonical
constexpr F GOLDILOCKS_P = 0xFFFFFFFF00000001ull;

struct Target { uint32_t id = 0xFFFFFFFFu; };
struct BoolTarget { Target target; };
struct U32Target { Target t; };
using Hash256Target = std::array<U32Target, 8>;

// prove() failed: the witness is inconsistent (plonky2 returns Err; the reference's tests unwrap() -> panic)
struct UnsatisfiedError : std::runtime_error { using std::runtime_error::runtime_error; };
struct VerifyError : std::runtime_error { using std::runtime_error::runtime_error; };

struct CircuitConfig {
  uint32_t num_wires = 135, num_routed_wires = 80, num_constants = 2, rate_bits = 3, cap_height = 4, num_challenges = 2,
           max_quotient_degree_factor = 8, proof_of_work_bits = 16, num_query_rounds = 28;
  static CircuitConfig standard_recursion_config() { return CircuitConfig(); }
};

// gate kinds of the own layout (sorted by degree then name, as plonky2 sorts its gate set)
enum GateKind : uint32_t { G_NOOP = 0, G_CONSTANT, G_PUBLIC_INPUT, G_SHA_ADD, G_ARITHMETIC, G_SHA_ROUND_A, G_SHA_ROUND_E, G_SHA_SCHED, G_POSEIDON, G_COUNT };
const char *gate_name(uint32_t kind);

class PartialWitness {
 public:
  void set_target(Target t, F value);
  template <size_t N> void set_target_arr(const std::array<Target, N> &ts, const std::vector<F> &vals) {
    for (size_t i = 0; i < N; i++) set_target(ts[i], vals[i]);
  }
  void set_u32_target(U32Target t, uint32_t v) { set_target(t.t, v); }
  void set_hash256_target(const Hash256Target &t, const uint8_t value[32]);  // 8 big-endian u32 limbs
  void set_bool_target(BoolTarget t, bool v) { set_target(t.target, v ? 1 : 0); }
  const std::vector<std::pair<uint32_t, F>> &entries() const { return entries_; }

 private:
  std::vector<std::pair<uint32_t, F>> entries_;
};

struct ProofWithPublicInputs {
  std::vector<uint64_t> proof;          // flat layout of include/lcp2.h
  std::vector<F> public_inputs;
};

class CircuitData;

class CircuitBuilder {
 public:
  explicit CircuitBuilder(const CircuitConfig &config);
  ~CircuitBuilder();
  Target add_virtual_target();
  template <size_t N> std::array<Target, N> add_virtual_target_arr() {
    std::array<Target, N> a;
    for (auto &t : a) t = add_virtual_target();
    return a;
  }
  BoolTarget add_virtual_bool_target_safe();
  Hash256Target add_virtual_hash256_target();
  Target constant(F v);
  Target zero() { return constant(0); }
  Target one() { return constant(1); }
  U32Target constant_u32(uint32_t v) { return U32Target{constant(v)}; }
  U32Target zero_u32() { return constant_u32(0); }
  Target mul_add(Target a, Target b, Target c);  // a * b + c
  Target mul(Target a, Target b);
  Target add(Target a, Target b);
  Target sub(Target a, Target b);
  Target add_many(const std::vector<Target> &terms);
  Target arithmetic(F const_0, Target x, Target y, F const_1, Target z);  // const_0 * x * y + const_1 * z: one ArithmeticGate operation
  Target mul_const(F c, Target x) { return arithmetic(c, x, one(), 0, zero()); }
  Target add_const(Target x, F c) { return arithmetic(1, x, one(), c, one()); }
  Target inverse(Target x);                              // 1 / x; x = 0 cannot be proved (x * inv = 1)
  BoolTarget is_equal(Target a, Target b);               // builder.is_equal
  BoolTarget and_(BoolTarget a, BoolTarget b) { return BoolTarget{mul(a.target, b.target)}; }
  BoolTarget or_(BoolTarget a, BoolTarget b) { return BoolTarget{arithmetic(GOLDILOCKS_P - 1, a.target, b.target, 1, add(a.target, b.target))}; }
  // one PoseidonGate row: the permutation of `inputs`, with inputs[0..4) and [4..8) exchanged first when swap is set
  // (plonky2 gates/poseidon.rs; what hash_n_to_hash_no_pad, the Merkle-path and the Challenger gadgets are made of)
  std::array<Target, 12> poseidon(const std::array<Target, 12> &inputs, BoolTarget swap);
  std::array<Target, 12> poseidon(const std::array<Target, 12> &inputs) { return poseidon(inputs, BoolTarget{zero()}); }
  // generators without constraints (the caller constrains the results): the inverse of x0 + x1 X in F[X]/(X^2 - 7) (0 for 0),
  // and the canonical value of x as low / high 32-bit halves
  std::array<Target, 2> hint_ext_inverse(Target x0, Target x1);
  std::array<Target, 2> hint_split_32(Target x);
  // a generator of the caller's: outputs = fn(inputs), no constraints (SimpleGenerator without a gate)
  std::vector<Target> hint(const std::vector<Target> &inputs, size_t num_outputs, std::function<void(const std::vector<F> &, std::vector<F> &)> fn);
  // the CANONICAL value of x as 32 low bits and hi_bits high bits (x = lo + 2^32 hi, both range checked, and hi = 2^32 - 1 forces
  // lo = 0: no second decomposition exists).  hi_bits < 32 also bounds x < 2^(32 + hi_bits).
  struct CanonicalSplit { Target lo, hi; std::vector<BoolTarget> bits; };
  CanonicalSplit split_canonical(Target x, uint32_t hi_bits = 32);
  BoolTarget not_(BoolTarget b);                         // builder.not(b) = 1 - b
  Target select(BoolTarget b, Target x, Target y);       // builder._if / select: b ? x : y
  void assert_bool(BoolTarget b);                        // b * b = b
  // builder.split_le(x, nbits): little-endian bits of x, each constrained boolean, recomposition constrained equal to x
  // (so x < 2^nbits is enforced); le_sum: sum of bits[first .. first + count) * 2^i
  std::vector<BoolTarget> split_le(Target x, size_t nbits);
  Target le_sum(const std::vector<BoolTarget> &bits, size_t first = 0, size_t count = (size_t)-1);
  void connect(Target a, Target b);
  void connect_u32(U32Target a, U32Target b) { connect(a.t, b.t); }
  void connect_hash256(const Hash256Target &a, const Hash256Target &b) { for (int i = 0; i < 8; i++) connect(a[i].t, b[i].t); }
  Hash256Target two_to_one_sha256(const Hash256Target &left, const Hash256Target &right);
  void register_public_input(Target t);
  void register_public_inputs(const std::vector<Target> &ts) { for (auto t : ts) register_public_input(t); }
  size_t num_gates() const;
  void print_gate_counts(int min_delta) const;
  // consumes the builder (the reference moves it): selectors, copy-constraint permutation, gate programs
  std::unique_ptr<CircuitData> build();

  struct Impl;
  Impl *impl() { return impl_.get(); }

 private:
  std::unique_ptr<Impl> impl_;
};

// what build() produced, in the form lcp2_circuit_create consumes (kept by value so tests can hand the same
// description to another prover, e.g. the CPU oracle)
struct CircuitDescription {
  lcp2_params params{};
  std::vector<uint64_t> constants_sigmas;  // [num_constants + num_routed][n]
  std::vector<uint64_t> k_is;
  uint32_t num_selectors = 0;
  std::vector<lcp2_gate> gates;
  std::vector<uint32_t> code;
  std::vector<uint64_t> imm;
  uint32_t num_public_inputs = 0, num_regs = 1;
  lcp2_circuit_desc c_desc() const;
};

class CircuitData {
 public:
  ~CircuitData();
  const CircuitDescription &description() const { return desc_; }
  uint32_t degree_bits() const { return desc_.params.degree_bits; }
  uint32_t quotient_degree_factor() const { return desc_.params.quotient_degree_factor; }
  // generate_partial_witness: runs the generators in creation order, checks every copy constraint and returns the
  // full wire matrix [num_wires][n] (column-major) and the public inputs; throws UnsatisfiedError on a conflict.
  void generate_witness(const PartialWitness &pw, std::vector<uint64_t> &wires, std::vector<F> &public_inputs) const;
  // attaches the MI355X backend: lcp2_circuit_create on `ctx` (commits the preprocessed polynomials)
  void attach_gpu(lcp2_ctx *ctx);
  // VerifierOnlyCircuitData { constants_sigmas_cap, circuit_digest } of the attached circuit (what a recursive verifier is given)
  void verifier_only_data(uint64_t digest[4], std::vector<uint64_t> &constants_sigmas_cap) const;
  ProofWithPublicInputs prove(const PartialWitness &pw);          // data.prove(pw): needs attach_gpu (no CPU prover here)
  // the device half of generate_partial_witness: SHA-256 rows are computed and written in HBM (K10), the few
  // remaining cells are scattered from the host; leaves the witness in the circuit's device buffer
  void generate_witness_gpu(const PartialWitness &pw, std::vector<F> &public_inputs);
  void read_device_witness(std::vector<uint64_t> &wires) const;   // test helper: copy of the device witness matrix
  void verify(const ProofWithPublicInputs &proof) const;          // data.verify(proof): host only
  struct Impl;
  const Impl *impl_for_tools() const { return impl_.get(); }  // diagnostics (op_stats)

 private:
  friend class CircuitBuilder;
  CircuitData() = default;
  CircuitDescription desc_;
  std::unique_ptr<Impl> impl_;
};

}  // namespace lc
