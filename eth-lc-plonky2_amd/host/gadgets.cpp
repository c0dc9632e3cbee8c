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

// src/targets.rs:237-302
VerifySyncCommitteeTarget add_virtual_verify_sync_committe_target(CircuitBuilder &builder) {
  VerifySyncCommitteeTarget t;
  t.is_attested_from_next_period = builder.add_virtual_bool_target_safe();
  t.cur_sync_committee_i = builder.add_virtual_hash256_target();
  t.cur_sync_committee_ii = builder.add_virtual_hash256_target();
  t.new_sync_committee_i = builder.add_virtual_hash256_target();
  t.new_sync_committee_ii = builder.add_virtual_hash256_target();
  t.finalized_state_root = builder.add_virtual_hash256_target();
  for (size_t i = 0; i < SYNC_COMMITTEE_HEIGHT; i++) t.new_sync_committee_ii_branch.push_back(builder.add_virtual_hash256_target());
  VerifyMerkleProofTarget branch = add_verify_merkle_proof_target(builder, SYNC_COMMITTEE_INDEX, SYNC_COMMITTEE_HEIGHT);
  builder.connect_hash256(branch.leaf, t.new_sync_committee_ii);
  for (size_t i = 0; i < SYNC_COMMITTEE_HEIGHT; i++) builder.connect_hash256(branch.proof[i], t.new_sync_committee_ii_branch[i]);
  builder.connect_hash256(branch.root, t.finalized_state_root);
  BoolTarget not_next = builder.not_(t.is_attested_from_next_period);
  for (int i = 0; i < 8; i++) {  // cur_i == new_i unless attested from the next period
    Target a = builder.mul(t.cur_sync_committee_i[i].t, not_next.target);
    Target b = builder.mul(t.new_sync_committee_i[i].t, not_next.target);
    builder.connect(a, b);
  }
  for (int i = 0; i < 8; i++) {  // new_i == cur_ii if attested from the next period
    Target a = builder.mul(t.cur_sync_committee_ii[i].t, t.is_attested_from_next_period.target);
    Target b = builder.mul(t.new_sync_committee_i[i].t, t.is_attested_from_next_period.target);
    builder.connect(a, b);
  }
  return t;
}

// src/utils.rs:93-113 for a u64 (see gadgets.hpp)
SlotConnectTarget add_virtual_biguint_hash256_connect_target(CircuitBuilder &builder) {
  SlotConnectTarget t;
  t.h256 = builder.add_virtual_hash256_target();
  for (int limb = 0; limb < 2; limb++) {
    // bit k of the big-endian u32 limb; integer bit 8a + j of this limb's 4 bytes is limb bit 24 - 8a + j
    std::vector<BoolTarget> be = builder.split_le(t.h256[limb].t, 32);
    for (int a = 0; a < 4; a++)
      for (int j = 0; j < 8; j++) t.bits.push_back(be[24 - 8 * a + j]);
  }
  for (int limb = 2; limb < 8; limb++) builder.connect(t.h256[limb].t, builder.zero());
  builder.connect(t.bits[63].target, builder.zero());
  t.value = builder.le_sum(t.bits, 0, 63);
  return t;
}

FindSyncCommitteeTarget add_virtual_find_sync_committee_target(CircuitBuilder &builder) {
  FindSyncCommitteeTarget t;
  t.attested_slot = add_virtual_biguint_hash256_connect_target(builder);
  t.cur_slot = add_virtual_biguint_hash256_connect_target(builder);
  t.cur_sync_committee_i = builder.add_virtual_hash256_target();
  t.cur_sync_committee_ii = builder.add_virtual_hash256_target();
  // N_SLOTS_PER_PERIOD = 2^13: the period is the slot without its 13 low bits
  Target attested_period = builder.le_sum(t.attested_slot.bits, 13, 50), cur_period = builder.le_sum(t.cur_slot.bits, 13, 50);
  t.is_attested_from_next_period = BoolTarget{builder.sub(attested_period, cur_period)};
  builder.assert_bool(t.is_attested_from_next_period);  // attested period is the current one (0) or the next (1)
  for (int i = 0; i < 8; i++)
    t.sync_committee_for_attested_slot[i] = U32Target{builder.select(t.is_attested_from_next_period, t.cur_sync_committee_ii[i].t, t.cur_sync_committee_i[i].t)};
  return t;
}

UpdateValidityTarget add_virtual_update_validity_target(CircuitBuilder &builder) {
  UpdateValidityTarget t;
  t.cur_slot = builder.add_virtual_target();
  t.finalized_slot = builder.add_virtual_target();
  t.participation = builder.add_virtual_target();
  builder.split_le(builder.sub(t.finalized_slot, t.cur_slot), 63);  // cur_slot <= finalized_slot (both < 2^63)
  Target over = builder.sub(t.participation, builder.constant((F)(FINALITY_THRESHOLD + 1)));
  builder.split_le(over, 10);                                       // participation > FINALITY_THRESHOLD (at most 512)
  return t;
}

ProofTarget add_virtual_proof_target(CircuitBuilder &builder, const CommonCircuitData *bls_sig_cd) {
  ProofTarget p;
  p.signing_root_bytes = builder.add_virtual_target_arr<32>();
  Hash256Target signing_root = builder.add_virtual_hash256_target();
  for (size_t idx = 0; idx < 8; idx++) builder.connect_u32(read_u32_be(builder, p.signing_root_bytes.data(), idx * 4), signing_root[idx]);
  p.domain = builder.add_virtual_hash256_target();
  p.attested_header_root = builder.add_virtual_hash256_target();
  p.attested_slot = builder.add_virtual_hash256_target();
  p.attested_proposer_index = builder.add_virtual_hash256_target();
  p.attested_parent_root = builder.add_virtual_hash256_target();
  p.attested_state_root = builder.add_virtual_hash256_target();
  p.attested_body_root = builder.add_virtual_hash256_target();
  p.finalized_header_root = builder.add_virtual_hash256_target();
  p.finalized_slot = builder.add_virtual_hash256_target();
  p.finalized_proposer_index = builder.add_virtual_hash256_target();
  p.finalized_parent_root = builder.add_virtual_hash256_target();
  p.finalized_state_root = builder.add_virtual_hash256_target();
  p.finalized_body_root = builder.add_virtual_hash256_target();
  for (size_t i = 0; i < FINALIZED_HEADER_HEIGHT; i++) p.finality_branch.push_back(builder.add_virtual_hash256_target());
  p.cur_state = builder.add_virtual_hash256_target();
  p.cur_slot = builder.add_virtual_hash256_target();
  p.cur_header = builder.add_virtual_hash256_target();
  p.cur_sync_committee_i = builder.add_virtual_hash256_target();
  p.cur_sync_committee_ii = builder.add_virtual_hash256_target();
  p.new_state = builder.add_virtual_hash256_target();
  p.new_sync_committee_i = builder.add_virtual_hash256_target();
  p.new_sync_committee_ii = builder.add_virtual_hash256_target();
  for (size_t i = 0; i < SYNC_COMMITTEE_HEIGHT; i++) p.new_sync_committee_ii_branch.push_back(builder.add_virtual_hash256_target());
  p.sync_committee = add_virtual_sync_committee_target(builder);
  Hash256Target sync_committee_ssz = ssz_sync_committee(builder, p.sync_committee);
  std::vector<Target> bit_targets;
  for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++) { p.sync_committee_bits.push_back(builder.add_virtual_bool_target_safe()); bit_targets.push_back(p.sync_committee_bits.back().target); }
  p.signature_bytes = builder.add_virtual_target_arr<96>();
  p.participation = builder.add_many(bit_targets);

  SigningRootTarget signing_root_target = add_virtual_signing_root_target(builder);
  BeaconBlockHeaderTarget attested = add_virtual_beacon_block_header_target(builder);
  BeaconBlockHeaderTarget finalized = add_virtual_beacon_block_header_target(builder);
  VerifyMerkleProofTarget finality_branch_target = add_verify_merkle_proof_target(builder, FINALIZED_HEADER_INDEX, FINALIZED_HEADER_HEIGHT);
  ContractStateTarget contract_state_target = add_virtual_contract_state_target(builder);
  FindSyncCommitteeTarget find_sync_committee_target = add_virtual_find_sync_committee_target(builder);
  VerifySyncCommitteeTarget verify_sync_committe_target = add_virtual_verify_sync_committe_target(builder);
  UpdateValidityTarget update_validity_target = add_virtual_update_validity_target(builder);
  SlotConnectTarget finalized_slot_connect = add_virtual_biguint_hash256_connect_target(builder);

  // *** signing root ***
  builder.connect_hash256(signing_root_target.signing_root, signing_root);
  builder.connect_hash256(signing_root_target.header_root, p.attested_header_root);
  builder.connect_hash256(signing_root_target.domain, p.domain);
  if (bls_sig_cd) {  // src/targets.rs:468-482
    if (bls_sig_cd->num_public_inputs != BLS_PROOF_PUBLIC_INPUTS) throw std::runtime_error("add_virtual_proof_target: the BLS proof must have 25 216 public inputs");
    p.has_bls_proof = true;
    p.bls_proof = add_virtual_proof_with_pis(builder, *bls_sig_cd);
    p.bls_verifier_data = add_virtual_verifier_data(builder, bls_sig_cd->params.cap_height);
    verify_proof(builder, p.bls_proof, p.bls_verifier_data, *bls_sig_cd);
    for (size_t idx = 0; idx < 32; idx++) builder.connect(p.bls_proof.public_inputs[idx], p.signing_root_bytes[idx]);
    for (size_t idx = 0; idx < 96; idx++) builder.connect(p.bls_proof.public_inputs[idx + 32], p.signature_bytes[idx]);
    for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++) {
      for (size_t j = 0; j < G1_PUBKEY_SIZE; j++) builder.connect(p.bls_proof.public_inputs[32 + 96 + i * (G1_PUBKEY_SIZE + 1) + j], p.sync_committee.pubkeys[i][j]);
      builder.connect(p.bls_proof.public_inputs[32 + 96 + i * (G1_PUBKEY_SIZE + 1) + G1_PUBKEY_SIZE], p.sync_committee_bits[i].target);
    }
  }
  // *** attested block header ***
  builder.connect_hash256(attested.body_root, p.attested_body_root);
  builder.connect_hash256(attested.header_root, p.attested_header_root);
  builder.connect_hash256(attested.parent_root, p.attested_parent_root);
  builder.connect_hash256(attested.proposer_index, p.attested_proposer_index);
  builder.connect_hash256(attested.slot, p.attested_slot);
  builder.connect_hash256(attested.state_root, p.attested_state_root);
  // *** finality branch ***
  builder.connect_hash256(finality_branch_target.leaf, p.finalized_header_root);
  builder.connect_hash256(finality_branch_target.root, p.attested_state_root);
  for (size_t i = 0; i < FINALIZED_HEADER_HEIGHT; i++) builder.connect_hash256(finality_branch_target.proof[i], p.finality_branch[i]);
  // *** finalized block header ***
  builder.connect_hash256(finalized.body_root, p.finalized_body_root);
  builder.connect_hash256(finalized.header_root, p.finalized_header_root);
  builder.connect_hash256(finalized.parent_root, p.finalized_parent_root);
  builder.connect_hash256(finalized.proposer_index, p.finalized_proposer_index);
  builder.connect_hash256(finalized.slot, p.finalized_slot);
  builder.connect_hash256(finalized.state_root, p.finalized_state_root);
  // *** sync committee ***   (targets.rs:541-542, 627-635: the slots enter through their Hash256 encodings)
  builder.connect_hash256(find_sync_committee_target.cur_slot.h256, p.cur_slot);
  builder.connect_hash256(find_sync_committee_target.attested_slot.h256, p.attested_slot);
  builder.connect_hash256(find_sync_committee_target.cur_sync_committee_i, p.cur_sync_committee_i);
  builder.connect_hash256(find_sync_committee_target.cur_sync_committee_ii, p.cur_sync_committee_ii);
  builder.connect_hash256(find_sync_committee_target.sync_committee_for_attested_slot, sync_committee_ssz);
  // *** update sync committee ***
  builder.connect(find_sync_committee_target.is_attested_from_next_period.target, verify_sync_committe_target.is_attested_from_next_period.target);
  builder.connect_hash256(verify_sync_committe_target.cur_sync_committee_i, p.cur_sync_committee_i);
  builder.connect_hash256(verify_sync_committe_target.cur_sync_committee_ii, p.cur_sync_committee_ii);
  builder.connect_hash256(verify_sync_committe_target.new_sync_committee_i, p.new_sync_committee_i);
  builder.connect_hash256(verify_sync_committe_target.new_sync_committee_ii, p.new_sync_committee_ii);
  builder.connect_hash256(verify_sync_committe_target.finalized_state_root, p.attested_state_root);
  for (size_t i = 0; i < SYNC_COMMITTEE_HEIGHT; i++) builder.connect_hash256(verify_sync_committe_target.new_sync_committee_ii_branch[i], p.new_sync_committee_ii_branch[i]);
  p.is_attested_from_next_period = find_sync_committee_target.is_attested_from_next_period;
  // *** update validity ***   (targets.rs:589-598, 637-640)
  builder.connect_hash256(finalized_slot_connect.h256, p.finalized_slot);
  builder.connect(update_validity_target.cur_slot, find_sync_committee_target.cur_slot.value);
  builder.connect(update_validity_target.finalized_slot, finalized_slot_connect.value);
  builder.connect(update_validity_target.participation, p.participation);
  // *** contract state ***
  builder.connect_hash256(contract_state_target.cur_state, p.cur_state);
  builder.connect_hash256(contract_state_target.new_state, p.new_state);
  builder.connect_hash256(contract_state_target.cur_header, p.cur_header);
  builder.connect_hash256(contract_state_target.cur_slot, p.cur_slot);
  builder.connect_hash256(contract_state_target.cur_sync_committee_i, p.cur_sync_committee_i);
  builder.connect_hash256(contract_state_target.cur_sync_committee_ii, p.cur_sync_committee_ii);
  builder.connect_hash256(contract_state_target.new_header, p.finalized_header_root);
  builder.connect_hash256(contract_state_target.new_slot, p.finalized_slot);
  builder.connect_hash256(contract_state_target.new_sync_committee_i, p.new_sync_committee_i);
  builder.connect_hash256(contract_state_target.new_sync_committee_ii, p.new_sync_committee_ii);
  return p;
}

static void u64_le_bytes(uint64_t v, uint8_t out[32]) { for (int i = 0; i < 32; i++) out[i] = i < 8 ? (uint8_t)(v >> (8 * i)) : 0; }

void set_proof_target(PartialWitness &witness, const uint8_t signing_root[32], const uint8_t domain[32], uint64_t attested_slot,
                      uint64_t attested_proposer_index, const uint8_t attested_header_root[32], const uint8_t attested_parent_root[32],
                      const uint8_t attested_state_root[32], const uint8_t attested_body_root[32], uint64_t finalized_slot,
                      uint64_t finalized_proposer_index, const uint8_t finalized_header_root[32], const uint8_t finalized_parent_root[32],
                      const uint8_t finalized_state_root[32], const uint8_t finalized_body_root[32], const uint8_t finality_branch[6][32],
                      const uint8_t cur_state[32], const uint8_t new_state[32], uint64_t cur_slot, const uint8_t cur_header[32],
                      const uint8_t cur_sync_committee_i[32], const uint8_t cur_sync_committee_ii[32], const uint8_t new_sync_committee_i[32],
                      const uint8_t new_sync_committee_ii[32], const std::vector<bool> &sync_committee_bits,
                      const uint8_t new_sync_committee_ii_branch[5][32], const uint8_t sync_committee_pubkeys[][48],
                      const uint8_t sync_committee_aggregate[48], const uint8_t signature[96], const ProofTarget &target) {
  uint8_t tmp[32];
  witness.set_hash256_target(target.attested_header_root, attested_header_root);
  witness.set_hash256_target(target.domain, domain);
  witness.set_target_arr(target.signing_root_bytes, std::vector<F>(signing_root, signing_root + 32));
  witness.set_hash256_target(target.attested_parent_root, attested_parent_root);
  witness.set_hash256_target(target.attested_state_root, attested_state_root);
  witness.set_hash256_target(target.attested_body_root, attested_body_root);
  u64_le_bytes(attested_slot, tmp); witness.set_hash256_target(target.attested_slot, tmp);
  u64_le_bytes(attested_proposer_index, tmp); witness.set_hash256_target(target.attested_proposer_index, tmp);
  witness.set_hash256_target(target.finalized_header_root, finalized_header_root);
  witness.set_hash256_target(target.finalized_parent_root, finalized_parent_root);
  witness.set_hash256_target(target.finalized_state_root, finalized_state_root);
  witness.set_hash256_target(target.finalized_body_root, finalized_body_root);
  u64_le_bytes(finalized_slot, tmp); witness.set_hash256_target(target.finalized_slot, tmp);
  u64_le_bytes(finalized_proposer_index, tmp); witness.set_hash256_target(target.finalized_proposer_index, tmp);
  for (size_t i = 0; i < FINALIZED_HEADER_HEIGHT; i++) witness.set_hash256_target(target.finality_branch[i], finality_branch[i]);
  u64_le_bytes(cur_slot, tmp); witness.set_hash256_target(target.cur_slot, tmp);
  witness.set_hash256_target(target.cur_state, cur_state);
  witness.set_hash256_target(target.cur_header, cur_header);
  witness.set_hash256_target(target.cur_sync_committee_i, cur_sync_committee_i);
  witness.set_hash256_target(target.cur_sync_committee_ii, cur_sync_committee_ii);
  witness.set_hash256_target(target.new_state, new_state);
  witness.set_hash256_target(target.new_sync_committee_i, new_sync_committee_i);
  witness.set_hash256_target(target.new_sync_committee_ii, new_sync_committee_ii);
  for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++) witness.set_bool_target(target.sync_committee_bits[i], sync_committee_bits[i]);
  for (size_t i = 0; i < SYNC_COMMITTEE_HEIGHT; i++) witness.set_hash256_target(target.new_sync_committee_ii_branch[i], new_sync_committee_ii_branch[i]);
  for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++)
    witness.set_target_arr(target.sync_committee.pubkeys[i], std::vector<F>(sync_committee_pubkeys[i], sync_committee_pubkeys[i] + G1_PUBKEY_SIZE));
  witness.set_target_arr(target.sync_committee.aggregate_pubkey, std::vector<F>(sync_committee_aggregate, sync_committee_aggregate + G1_PUBKEY_SIZE));
  witness.set_target_arr(target.signature_bytes, std::vector<F>(signature, signature + 96));
}

void set_bls_proof_target(PartialWitness &witness, const ProofTarget &target, const ProofWithPublicInputs &bls_proof, const uint64_t circuit_digest[4],
                          const std::vector<uint64_t> &constants_sigmas_cap) {
  if (!target.has_bls_proof) throw std::runtime_error("set_bls_proof_target: the circuit was built without the recursive verifier");
  set_proof_with_pis_target(witness, target.bls_proof, bls_proof);
  set_verifier_data_target(witness, target.bls_verifier_data, circuit_digest, constants_sigmas_cap);
}

BlsStatementStandIn build_bls_statement_stand_in() {
  BlsStatementStandIn c;
  CircuitBuilder builder(CircuitConfig::standard_recursion_config());
  for (size_t i = 0; i < BLS_PROOF_PUBLIC_INPUTS; i++) c.public_inputs.push_back(builder.add_virtual_target());
  for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++) builder.assert_bool(BoolTarget{c.public_inputs[32 + 96 + i * (G1_PUBKEY_SIZE + 1) + G1_PUBKEY_SIZE]});
  builder.register_public_inputs(c.public_inputs);
  c.data = builder.build();
  return c;
}
void set_bls_statement_stand_in(PartialWitness &witness, const BlsStatementStandIn &c, const uint8_t signing_root[32], const uint8_t signature[96],
                                const uint8_t pubkeys[][48], const std::vector<bool> &bits) {
  for (size_t i = 0; i < 32; i++) witness.set_target(c.public_inputs[i], signing_root[i]);
  for (size_t i = 0; i < 96; i++) witness.set_target(c.public_inputs[32 + i], signature[i]);
  for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++) {
    for (size_t j = 0; j < G1_PUBKEY_SIZE; j++) witness.set_target(c.public_inputs[32 + 96 + i * (G1_PUBKEY_SIZE + 1) + j], pubkeys[i][j]);
    witness.set_target(c.public_inputs[32 + 96 + i * (G1_PUBKEY_SIZE + 1) + G1_PUBKEY_SIZE], bits[i] ? 1 : 0);
  }
}

}  // namespace lc
