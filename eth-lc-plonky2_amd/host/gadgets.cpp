// Gadgets of the reference, restated on the C++ CircuitBuilder (see gadgets.hpp for the source lines).
#include "gadgets.hpp"
#include <stdexcept>

namespace lc {

// src/merkle_tree_gadget.rs:28-40
std::vector<Hash256Target> compute_next_layer(CircuitBuilder &builder, size_t layer_size, const std::vector<Hash256Target> &prev_layer) {
  std::vector<Hash256Target> out;
  for (size_t i = 0; i < layer_size; i++) out.push_back(builder.two_to_one_sha256(prev_layer[2 * i], prev_layer[2 * i + 1]));
  return out;
}

// src/merkle_tree_gadget.rs:42-59
MerkleTreeSha256Target add_virtual_merkle_tree_sha256_target(CircuitBuilder &builder, size_t height) {
  const size_t num_leaves = (size_t)1 << height;
  std::vector<Hash256Target> leaves;
  for (size_t i = 0; i < num_leaves; i++) leaves.push_back(builder.add_virtual_hash256_target());
  std::vector<Hash256Target> layer = leaves;
  for (size_t i = 1; i < height; i++) layer = compute_next_layer(builder, (size_t)1 << (height - i), layer);
  if (layer.size() != 2) throw std::runtime_error("Error buiding merkle tree");
  Hash256Target root = builder.two_to_one_sha256(layer[0], layer[1]);
  return MerkleTreeSha256Target{root, leaves};
}

// src/merkle_tree_gadget.rs:61-87
VerifyMerkleProofTarget add_verify_merkle_proof_target(CircuitBuilder &builder, size_t leaf_index, size_t height) {
  Hash256Target root = builder.add_virtual_hash256_target();
  std::vector<Hash256Target> proof;
  Hash256Target leaf = builder.add_virtual_hash256_target();
  size_t curr_index = leaf_index;
  Hash256Target next_hash = leaf;
  for (size_t i = 0; i < height; i++) {
    proof.push_back(builder.add_virtual_hash256_target());
    if (curr_index % 2 == 0) next_hash = builder.two_to_one_sha256(next_hash, proof[i]);
    else next_hash = builder.two_to_one_sha256(proof[i], next_hash);
    curr_index /= 2;
  }
  builder.connect_hash256(next_hash, root);
  return VerifyMerkleProofTarget{leaf, proof, root};
}

// src/merkle_tree_gadget.rs:89-130
VerifyMerkleProofConditionalTarget add_verify_merkle_proof_conditional_target(CircuitBuilder &builder, size_t leaf_index, size_t height) {
  Hash256Target root = builder.add_virtual_hash256_target();
  std::vector<Hash256Target> proof;
  Hash256Target leaf = builder.add_virtual_hash256_target();
  Hash256Target next_hash_v = builder.add_virtual_hash256_target();
  Hash256Target root_v = builder.add_virtual_hash256_target();
  BoolTarget v = builder.add_virtual_bool_target_safe();
  size_t curr_index = leaf_index;
  Hash256Target next_hash = leaf;
  for (size_t i = 0; i < height; i++) {
    proof.push_back(builder.add_virtual_hash256_target());
    if (curr_index % 2 == 0) next_hash = builder.two_to_one_sha256(next_hash, proof[i]);
    else next_hash = builder.two_to_one_sha256(proof[i], next_hash);
    curr_index /= 2;
  }
  for (int i = 0; i < 8; i++) {
    Target temp1 = builder.mul(v.target, next_hash[i].t);
    Target temp2 = builder.mul(v.target, root[i].t);
    builder.connect(next_hash_v[i].t, temp1);
    builder.connect(root_v[i].t, temp2);
  }
  builder.connect_hash256(next_hash_v, root_v);
  return VerifyMerkleProofConditionalTarget{leaf, proof, root, v};
}

// src/merkle_tree_gadget.rs:132-150
void set_verify_merkle_proof_target(PartialWitness &witness, const uint8_t leaf[32], const std::vector<std::array<uint8_t, 32>> &proof,
                                    const uint8_t root[32], const VerifyMerkleProofTarget &target) {
  if (proof.size() != target.proof.size()) throw std::runtime_error("Incorrect number of proof elements");
  witness.set_hash256_target(target.leaf, leaf);
  for (size_t i = 0; i < target.proof.size(); i++) witness.set_hash256_target(target.proof[i], proof[i].data());
  witness.set_hash256_target(target.root, root);
}

// src/merkle_tree_gadget.rs:152-165
void set_partial_merkle_tree_sha256_target(PartialWitness &witness, const std::vector<std::array<uint8_t, 32>> &leaves,
                                           const MerkleTreeSha256Target &target) {
  if (leaves.size() != target.leaves.size()) throw std::runtime_error("Not correct number of leaf values provided");
  for (size_t i = 0; i < target.leaves.size(); i++) witness.set_hash256_target(target.leaves[i], leaves[i].data());
}

// src/sync_committee_pubkeys.rs:15-29
SyncCommitteeTarget add_virtual_sync_committee_target(CircuitBuilder &builder) {
  SyncCommitteeTarget t;
  for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++) t.pubkeys.push_back(builder.add_virtual_target_arr<G1_PUBKEY_SIZE>());
  t.aggregate_pubkey = builder.add_virtual_target_arr<G1_PUBKEY_SIZE>();
  return t;
}

// src/sync_committee_pubkeys.rs:31-45
U32Target read_u32_be(CircuitBuilder &builder, const Target *arr, size_t index) {
  Target two_pow_8 = builder.constant(1u << 8);
  Target two_pow_16 = builder.constant(1u << 16);
  Target two_pow_24 = builder.constant(1u << 24);
  Target u32_be = builder.mul_add(arr[index + 2], two_pow_8, arr[index + 3]);
  u32_be = builder.mul_add(arr[index + 1], two_pow_16, u32_be);
  u32_be = builder.mul_add(arr[index], two_pow_24, u32_be);
  return U32Target{u32_be};
}

// src/sync_committee_pubkeys.rs:47-87
Hash256Target ssz_sync_committee(CircuitBuilder &builder, const SyncCommitteeTarget &sync_committee) {
  auto pack = [&](const std::array<Target, G1_PUBKEY_SIZE> &pk, const Hash256Target &leaf0, const Hash256Target &leaf1) {
    for (size_t idx = 0; idx < 8; idx++) builder.connect_u32(read_u32_be(builder, pk.data(), idx * 4), leaf0[idx]);
    for (size_t idx = 0; idx < 8; idx++) {
      if (idx < 4) builder.connect_u32(read_u32_be(builder, pk.data(), idx * 4 + 32), leaf1[idx]);
      else builder.connect_u32(builder.constant_u32(0), leaf1[idx]);
    }
  };
  MerkleTreeSha256Target pubkey_merkle_tree = add_virtual_merkle_tree_sha256_target(builder, LOG2_SYNC_COMMITTEE_SIZE + 1);
  for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++) pack(sync_committee.pubkeys[i], pubkey_merkle_tree.leaves[2 * i], pubkey_merkle_tree.leaves[2 * i + 1]);
  MerkleTreeSha256Target aggregate_merkle_tree = add_virtual_merkle_tree_sha256_target(builder, 1);
  pack(sync_committee.aggregate_pubkey, aggregate_merkle_tree.leaves[0], aggregate_merkle_tree.leaves[1]);
  MerkleTreeSha256Target sync_committee_merkle_tree = add_virtual_merkle_tree_sha256_target(builder, 1);
  builder.connect_hash256(pubkey_merkle_tree.root, sync_committee_merkle_tree.leaves[0]);
  builder.connect_hash256(aggregate_merkle_tree.root, sync_committee_merkle_tree.leaves[1]);
  return sync_committee_merkle_tree.root;
}

static void zero_leaves(CircuitBuilder &builder, const MerkleTreeSha256Target &tree, size_t from) {
  U32Target zero_u32 = builder.zero_u32();
  for (size_t i = from; i < tree.leaves.size(); i++)
    for (auto &limb : tree.leaves[i]) builder.connect_u32(limb, zero_u32);
}

// src/targets.rs:121-145
SigningRootTarget add_virtual_signing_root_target(CircuitBuilder &builder) {
  Hash256Target header_root = builder.add_virtual_hash256_target();
  Hash256Target domain = builder.add_virtual_hash256_target();
  Hash256Target signing_root = builder.add_virtual_hash256_target();
  MerkleTreeSha256Target merkle_tree_target = add_virtual_merkle_tree_sha256_target(builder, 1);
  builder.connect_hash256(merkle_tree_target.leaves[0], header_root);
  builder.connect_hash256(merkle_tree_target.leaves[1], domain);
  zero_leaves(builder, merkle_tree_target, 2);
  builder.connect_hash256(merkle_tree_target.root, signing_root);
  return SigningRootTarget{signing_root, header_root, domain};
}

// src/targets.rs:147-181
BeaconBlockHeaderTarget add_virtual_beacon_block_header_target(CircuitBuilder &builder) {
  Hash256Target slot = builder.add_virtual_hash256_target();
  Hash256Target proposer_index = builder.add_virtual_hash256_target();
  Hash256Target parent_root = builder.add_virtual_hash256_target();
  Hash256Target state_root = builder.add_virtual_hash256_target();
  Hash256Target body_root = builder.add_virtual_hash256_target();
  Hash256Target header_root = builder.add_virtual_hash256_target();
  MerkleTreeSha256Target merkle_tree_target = add_virtual_merkle_tree_sha256_target(builder, 3);
  builder.connect_hash256(merkle_tree_target.leaves[0], slot);
  builder.connect_hash256(merkle_tree_target.leaves[1], proposer_index);
  builder.connect_hash256(merkle_tree_target.leaves[2], parent_root);
  builder.connect_hash256(merkle_tree_target.leaves[3], state_root);
  builder.connect_hash256(merkle_tree_target.leaves[4], body_root);
  zero_leaves(builder, merkle_tree_target, 5);
  builder.connect_hash256(merkle_tree_target.root, header_root);
  return BeaconBlockHeaderTarget{header_root, slot, proposer_index, parent_root, state_root, body_root};
}

// src/targets.rs:685-707 (u64 -> 32-byte little-endian leaf :819-821)
void set_beacon_block_header_target(PartialWitness &witness, const uint8_t header_root[32], uint64_t slot, uint64_t proposer_index,
                                    const uint8_t parent_root[32], const uint8_t state_root[32], const uint8_t body_root[32],
                                    const BeaconBlockHeaderTarget &target) {
  uint8_t slot_bytes[32] = {0}, proposer_bytes[32] = {0};
  for (int i = 0; i < 8; i++) { slot_bytes[i] = (uint8_t)(slot >> (8 * i)); proposer_bytes[i] = (uint8_t)(proposer_index >> (8 * i)); }
  witness.set_hash256_target(target.header_root, header_root);
  witness.set_hash256_target(target.slot, slot_bytes);
  witness.set_hash256_target(target.proposer_index, proposer_bytes);
  witness.set_hash256_target(target.parent_root, parent_root);
  witness.set_hash256_target(target.state_root, state_root);
  witness.set_hash256_target(target.body_root, body_root);
}

// src/targets.rs:334-389
ContractStateTarget add_virtual_contract_state_target(CircuitBuilder &builder) {
  Hash256Target cur_state = builder.add_virtual_hash256_target();
  Hash256Target new_state = builder.add_virtual_hash256_target();
  Hash256Target cur_slot = builder.add_virtual_hash256_target();
  Hash256Target cur_header = builder.add_virtual_hash256_target();
  Hash256Target cur_sync_committee_i = builder.add_virtual_hash256_target();
  Hash256Target cur_sync_committee_ii = builder.add_virtual_hash256_target();
  Hash256Target new_slot = builder.add_virtual_hash256_target();
  Hash256Target new_header = builder.add_virtual_hash256_target();
  Hash256Target new_sync_committee_i = builder.add_virtual_hash256_target();
  Hash256Target new_sync_committee_ii = builder.add_virtual_hash256_target();
  MerkleTreeSha256Target cur_merkle_tree_target = add_virtual_merkle_tree_sha256_target(builder, 2);
  builder.connect_hash256(cur_merkle_tree_target.leaves[0], cur_slot);
  builder.connect_hash256(cur_merkle_tree_target.leaves[1], cur_header);
  builder.connect_hash256(cur_merkle_tree_target.leaves[2], cur_sync_committee_i);
  builder.connect_hash256(cur_merkle_tree_target.leaves[3], cur_sync_committee_ii);
  zero_leaves(builder, cur_merkle_tree_target, 4);
  MerkleTreeSha256Target new_merkle_tree_target = add_virtual_merkle_tree_sha256_target(builder, 2);
  builder.connect_hash256(new_merkle_tree_target.leaves[0], new_slot);
  builder.connect_hash256(new_merkle_tree_target.leaves[1], new_header);
  builder.connect_hash256(new_merkle_tree_target.leaves[2], new_sync_committee_i);
  builder.connect_hash256(new_merkle_tree_target.leaves[3], new_sync_committee_ii);
  zero_leaves(builder, new_merkle_tree_target, 4);
  builder.connect_hash256(cur_merkle_tree_target.root, cur_state);
  builder.connect_hash256(new_merkle_tree_target.root, new_state);
  return ContractStateTarget{cur_state, new_state, cur_header, cur_slot, cur_sync_committee_i, cur_sync_committee_ii,
                             new_header, new_slot, new_sync_committee_i, new_sync_committee_ii};
}

}  // namespace lc
