// The recursive verifier gadget: a circuit that checks a proof of another circuit of this library.
//
// Mirrors what the reference uses from plonky2 for its BLS-signature proof (src/targets.rs:468-482, src/main.rs:172-176):
//   builder.add_virtual_proof_with_pis(&common_data)          -> add_virtual_proof_with_pis
//   builder.add_virtual_verifier_data(cap_height)             -> add_virtual_verifier_data   (constant_verifier_data: as constants)
//   builder.verify_proof::<C>(&proof, &verifier_data, &common) -> verify_proof
//   pw.set_proof_with_pis_target / set_verifier_data_target    -> same names
// and restates, in circuit form, exactly the checks of this library's host verifier (csrc/verifier.hip, itself following plonky2
// 0.1.4 plonk/verifier.rs, plonk/vanishing_poly.rs, plonk/get_challenges.rs, fri/verifier.rs; recursive forms:
// plonk/recursive_verifier.rs, fri/recursive_verifier.rs [RECALL]):
//   the Fiat-Shamir transcript (Challenger over PoseidonGate rows), the vanishing identity at zeta (the inner circuit's gate
//   programs interpreted over extension-field targets, permutation argument, Z_H(zeta) t(zeta)), the proof of work, and per query
//   round the four initial Merkle proofs, the batched opening at x, every FRI layer (consistency, coset interpolation at beta,
//   Merkle proof) and the final polynomial.
// plonky2 builds this from dedicated gates (ReducingGate, ArithmeticExtensionGate, RandomAccessGate, CosetInterpolationGate ...);
// here it is made of the host layer's ArithmeticGate operations, bit decompositions and PoseidonGate rows, so it costs more rows
// (about 2^13 for an inner circuit of 2^7 rows) and none of it needs new device code.
#pragma once
#include "lc_plonky2.hpp"

namespace lc {

struct ExtensionTarget { Target c0, c1; };  // c0 + c1 X in F[X] / (X^2 - 7)

// plonky2 CommonCircuitData: what the shape of a proof and its checks depend on (no commitment in it)
struct CommonCircuitData {
  lcp2_params params{};
  std::vector<uint64_t> k_is;
  uint32_t num_selectors = 0, num_public_inputs = 0;
  std::vector<lcp2_gate> gates;
  std::vector<uint32_t> code;
  std::vector<uint64_t> imm;
  static CommonCircuitData of(const CircuitDescription &d);
};

// plonky2 VerifierCircuitTarget { constants_sigmas_cap, circuit_digest }
struct VerifierCircuitTarget {
  std::vector<Target> constants_sigmas_cap;  // 4 << cap_height
  std::array<Target, 4> circuit_digest;
};

// plonky2 ProofWithPublicInputsTarget: the flat proof of include/lcp2.h, one target per word
struct ProofWithPublicInputsTarget {
  std::vector<Target> proof;
  std::vector<Target> public_inputs;
};

ProofWithPublicInputsTarget add_virtual_proof_with_pis(CircuitBuilder &builder, const CommonCircuitData &common_data);
VerifierCircuitTarget add_virtual_verifier_data(CircuitBuilder &builder, uint32_t cap_height);
VerifierCircuitTarget constant_verifier_data(CircuitBuilder &builder, const uint64_t circuit_digest[4], const std::vector<uint64_t> &constants_sigmas_cap);
void verify_proof(CircuitBuilder &builder, const ProofWithPublicInputsTarget &proof_with_pis, const VerifierCircuitTarget &inner_verifier_data,
                  const CommonCircuitData &inner_common_data);
void set_proof_with_pis_target(PartialWitness &witness, const ProofWithPublicInputsTarget &target, const ProofWithPublicInputs &proof);
void set_verifier_data_target(PartialWitness &witness, const VerifierCircuitTarget &target, const uint64_t circuit_digest[4],
                              const std::vector<uint64_t> &constants_sigmas_cap);

// the pieces, usable on their own (plonky2: hash_n_to_hash_no_pad, verify_merkle_proof_to_cap_with_cap_index, RecursiveChallenger)
std::array<Target, 4> hash_n_to_hash_no_pad(CircuitBuilder &builder, const std::vector<Target> &inputs);
void verify_merkle_proof_to_cap(CircuitBuilder &builder, const std::vector<Target> &leaf_data, const std::vector<BoolTarget> &leaf_index_bits,
                                const std::vector<Target> &siblings, const std::vector<Target> &merkle_cap);

class RecursiveChallenger {
 public:
  explicit RecursiveChallenger(CircuitBuilder &b);
  void observe_element(Target t);
  void observe_elements(const std::vector<Target> &ts) { for (Target t : ts) observe_element(t); }
  Target get_challenge();
  ExtensionTarget get_extension_challenge() { Target a = get_challenge(), b = get_challenge(); return ExtensionTarget{a, b}; }

 private:
  void duplex();
  CircuitBuilder &b_;
  std::array<Target, 12> sponge_;
  std::vector<Target> input_, output_;
};

}  // namespace lc
