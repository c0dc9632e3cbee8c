// The reference's gadget API, function for function (same names, argument order and meaning):
//   src/merkle_tree_gadget.rs      MerkleTreeSha256Target, VerifyMerkleProofTarget(+Conditional), compute_next_layer,
//                                  add_virtual_merkle_tree_sha256_target, add_verify_merkle_proof_target(+conditional),
//                                  set_verify_merkle_proof_target, set_partial_merkle_tree_sha256_target
//   src/sync_committee_pubkeys.rs  SyncCommitteeTarget, add_virtual_sync_committee_target, read_u32_be, ssz_sync_committee
//   src/targets.rs                 SigningRootTarget, BeaconBlockHeaderTarget, ContractStateTarget and their
//                                  add_virtual_* / set_* functions (the SHA-256-only sub-circuits)
#pragma once
#include "lc_plonky2.hpp"
#include "recursion.hpp"

namespace lc {

constexpr size_t SYNC_COMMITTEE_SIZE = 512;       // src/sync_committee_pubkeys.rs:6
constexpr size_t LOG2_SYNC_COMMITTEE_SIZE = 9;    // :7
constexpr size_t G1_PUBKEY_SIZE = 48;             // :8
constexpr size_t FINALIZED_HEADER_INDEX = 105;    // src/targets.rs:25
constexpr size_t FINALIZED_HEADER_HEIGHT = 6;     // :26
constexpr size_t SYNC_COMMITTEE_HEIGHT = 5;       // :27
constexpr size_t SYNC_COMMITTEE_INDEX = 55;       // :28

struct MerkleTreeSha256Target { Hash256Target root; std::vector<Hash256Target> leaves; };
struct VerifyMerkleProofTarget { Hash256Target leaf; std::vector<Hash256Target> proof; Hash256Target root; };
struct VerifyMerkleProofConditionalTarget { Hash256Target leaf; std::vector<Hash256Target> proof; Hash256Target root; BoolTarget v; };
struct SyncCommitteeTarget { std::vector<std::array<Target, G1_PUBKEY_SIZE>> pubkeys; std::array<Target, G1_PUBKEY_SIZE> aggregate_pubkey; };
struct SigningRootTarget { Hash256Target signing_root, header_root, domain; };
struct BeaconBlockHeaderTarget { Hash256Target header_root, slot, proposer_index, parent_root, state_root, body_root; };
struct ContractStateTarget {
  Hash256Target cur_state, new_state, cur_header, cur_slot, cur_sync_committee_i, cur_sync_committee_ii, new_header, new_slot,
      new_sync_committee_i, new_sync_committee_ii;
};

constexpr size_t FINALITY_THRESHOLD = 342;      // src/targets.rs:29
constexpr size_t N_SLOTS_PER_PERIOD = 8192;     // src/targets.rs:30
// src/utils.rs:93-113 BigUintHash256ConnectTarget for a u64: the Hash256Target of an SSZ uint64 (32 bytes, little endian,
// read as 8 big-endian u32 limbs) tied bit by bit to the integer it encodes.  The reference keeps the integer as a
// 256-bit BigUintTarget; here it is one field element, the 192 upper bits of the encoding are constrained to zero and
// bit 63 as well (value < 2^63 < p), which every SSZ slot satisfies.
struct SlotConnectTarget {
  Hash256Target h256;
  Target value;
  std::vector<BoolTarget> bits;  // 64, little endian
};
struct FindSyncCommitteeTarget {  // src/targets.rs:37-45
  SlotConnectTarget attested_slot, cur_slot;
  BoolTarget is_attested_from_next_period;
  Hash256Target cur_sync_committee_i, cur_sync_committee_ii, sync_committee_for_attested_slot;
};
struct UpdateValidityTarget {  // src/targets.rs:64-69
  Target cur_slot, finalized_slot, participation;
};
struct VerifySyncCommitteeTarget {
  BoolTarget is_attested_from_next_period;
  Hash256Target cur_sync_committee_i, cur_sync_committee_ii, new_sync_committee_i, new_sync_committee_ii, finalized_state_root;
  std::vector<Hash256Target> new_sync_committee_ii_branch;
};
// public inputs of the reference's BLS-signature proof, in its order (src/targets.rs:471-482): the 32 signing-root bytes, the 96
// signature bytes, then per committee member its 48 pubkey bytes and its participation bit
constexpr size_t BLS_PROOF_PUBLIC_INPUTS = 32 + 96 + SYNC_COMMITTEE_SIZE * (G1_PUBKEY_SIZE + 1);  // 25 216

// src/targets.rs:84-119 ProofTarget (the BigUint slot copies are the u64 form above).  bls_proof / bls_verifier_data are present
// when add_virtual_proof_target was given the common data of a BLS-signature proof to verify recursively.
struct ProofTarget {
  bool has_bls_proof = false;
  ProofWithPublicInputsTarget bls_proof;
  VerifierCircuitTarget bls_verifier_data;
  std::array<Target, 32> signing_root_bytes;
  Hash256Target attested_header_root, domain, attested_slot, attested_proposer_index, attested_parent_root, attested_state_root,
      attested_body_root, finalized_header_root;
  std::vector<Hash256Target> finality_branch;
  Hash256Target finalized_slot, finalized_proposer_index, finalized_parent_root, finalized_state_root, finalized_body_root, cur_state,
      cur_slot, cur_header, cur_sync_committee_i, cur_sync_committee_ii, new_state, new_sync_committee_i, new_sync_committee_ii;
  std::vector<BoolTarget> sync_committee_bits;
  std::vector<Hash256Target> new_sync_committee_ii_branch;
  SyncCommitteeTarget sync_committee;
  std::array<Target, 96> signature_bytes;
  BoolTarget is_attested_from_next_period;  // derived in-circuit by find_sync_committee
  Target participation;                      // sum of the sync committee bits, compared with FINALITY_THRESHOLD by update_validity
};

std::vector<Hash256Target> compute_next_layer(CircuitBuilder &builder, size_t layer_size, const std::vector<Hash256Target> &prev_layer);
MerkleTreeSha256Target add_virtual_merkle_tree_sha256_target(CircuitBuilder &builder, size_t height);
VerifyMerkleProofTarget add_verify_merkle_proof_target(CircuitBuilder &builder, size_t leaf_index, size_t height);
VerifyMerkleProofConditionalTarget add_verify_merkle_proof_conditional_target(CircuitBuilder &builder, size_t leaf_index, size_t height);
void set_verify_merkle_proof_target(PartialWitness &witness, const uint8_t leaf[32], const std::vector<std::array<uint8_t, 32>> &proof,
                                    const uint8_t root[32], const VerifyMerkleProofTarget &target);
void set_partial_merkle_tree_sha256_target(PartialWitness &witness, const std::vector<std::array<uint8_t, 32>> &leaves,
                                           const MerkleTreeSha256Target &target);

SyncCommitteeTarget add_virtual_sync_committee_target(CircuitBuilder &builder);
U32Target read_u32_be(CircuitBuilder &builder, const Target *arr, size_t index);
Hash256Target ssz_sync_committee(CircuitBuilder &builder, const SyncCommitteeTarget &sync_committee);

SigningRootTarget add_virtual_signing_root_target(CircuitBuilder &builder);
BeaconBlockHeaderTarget add_virtual_beacon_block_header_target(CircuitBuilder &builder);
void set_beacon_block_header_target(PartialWitness &witness, const uint8_t header_root[32], uint64_t slot, uint64_t proposer_index,
                                    const uint8_t parent_root[32], const uint8_t state_root[32], const uint8_t body_root[32],
                                    const BeaconBlockHeaderTarget &target);
ContractStateTarget add_virtual_contract_state_target(CircuitBuilder &builder);
VerifySyncCommitteeTarget add_virtual_verify_sync_committe_target(CircuitBuilder &builder);
SlotConnectTarget add_virtual_biguint_hash256_connect_target(CircuitBuilder &builder);
// src/targets.rs:184-235: periods = slot / 8192 from the slots' bits; the attested period must be the current one or the
// next (d = attested_period - cur_period is constrained boolean: the reference's two is_equal + or + assert), the signing
// committee is i for the current period and ii for the next
FindSyncCommitteeTarget add_virtual_find_sync_committee_target(CircuitBuilder &builder);
// src/targets.rs:304-332: cur_slot <= finalized_slot and participation > FINALITY_THRESHOLD (range checks of the differences)
UpdateValidityTarget add_virtual_update_validity_target(CircuitBuilder &builder);
// src/targets.rs:391-683.  bls_sig_cd = nullptr: the recursive BLS verifier (:468-482) is left out (BASELINE configs[2]).
// With the common data of an inner proof that has BLS_PROOF_PUBLIC_INPUTS public inputs, :468-482 is built as in the reference:
// add_virtual_proof_with_pis + add_virtual_verifier_data + verify_proof, and the inner public inputs are connected to the signing
// root bytes, the signature bytes, the committee's pubkey bytes and the participation bits.
ProofTarget add_virtual_proof_target(CircuitBuilder &builder, const CommonCircuitData *bls_sig_cd = nullptr);
// the tail of set_proof_target in the reference (set_proof_with_pis_target + set_verifier_data_target for the BLS proof)
void set_bls_proof_target(PartialWitness &witness, const ProofTarget &target, const ProofWithPublicInputs &bls_proof, const uint64_t circuit_digest[4],
                          const std::vector<uint64_t> &constants_sigmas_cap);

// A STAND-IN for the circuit whose proof the reference verifies here (starky_bls12_381's aggregate_proof, src/main.rs:170): it
// has the same public inputs in the same order and constrains only that the participation flags are boolean - it says NOTHING
// about the BLS signature.  It exists so that the recursive half of the light-client circuit (proof target, verifier data,
// verify_proof, the 25 216 connections) can be built, proved and measured; the BLS12-381 verifier itself is out of scope.
struct BlsStatementStandIn {
  std::unique_ptr<CircuitData> data;
  std::vector<Target> public_inputs;  // BLS_PROOF_PUBLIC_INPUTS
};
BlsStatementStandIn build_bls_statement_stand_in();
void set_bls_statement_stand_in(PartialWitness &witness, const BlsStatementStandIn &circuit, const uint8_t signing_root[32], const uint8_t signature[96],
                                const uint8_t sync_committee_pubkeys[][48], const std::vector<bool> &sync_committee_bits);
// src/targets.rs:771-898 (same argument order; the BLS proof / verifier data arguments are dropped)
void set_proof_target(PartialWitness &witness, const uint8_t signing_root[32], const uint8_t domain[32], uint64_t attested_slot,
                      uint64_t attested_proposer_index, const uint8_t attested_header_root[32], const uint8_t attested_parent_root[32],
                      const uint8_t attested_state_root[32], const uint8_t attested_body_root[32], uint64_t finalized_slot,
                      uint64_t finalized_proposer_index, const uint8_t finalized_header_root[32], const uint8_t finalized_parent_root[32],
                      const uint8_t finalized_state_root[32], const uint8_t finalized_body_root[32], const uint8_t finality_branch[6][32],
                      const uint8_t cur_state[32], const uint8_t new_state[32], uint64_t cur_slot, const uint8_t cur_header[32],
                      const uint8_t cur_sync_committee_i[32], const uint8_t cur_sync_committee_ii[32], const uint8_t new_sync_committee_i[32],
                      const uint8_t new_sync_committee_ii[32], const std::vector<bool> &sync_committee_bits,
                      const uint8_t new_sync_committee_ii_branch[5][32], const uint8_t sync_committee_pubkeys[][48],
                      const uint8_t sync_committee_aggregate[48], const uint8_t signature[96], const ProofTarget &target);

}  // namespace lc
