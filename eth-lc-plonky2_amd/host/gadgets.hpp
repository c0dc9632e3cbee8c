// The reference's gadget API, function for function (same names, argument order and meaning):
//   src/merkle_tree_gadget.rs      MerkleTreeSha256Target, VerifyMerkleProofTarget(+Conditional), compute_next_layer,
//                                  add_virtual_merkle_tree_sha256_target, add_verify_merkle_proof_target(+conditional),
//                                  set_verify_merkle_proof_target, set_partial_merkle_tree_sha256_target
//   src/sync_committee_pubkeys.rs  SyncCommitteeTarget, add_virtual_sync_committee_target, read_u32_be, ssz_sync_committee
//   src/targets.rs                 SigningRootTarget, BeaconBlockHeaderTarget, ContractStateTarget and their
//                                  add_virtual_* / set_* functions (the SHA-256-only sub-circuits)
#pragma once
#include "lc_plonky2.hpp"

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

}  // namespace lc
