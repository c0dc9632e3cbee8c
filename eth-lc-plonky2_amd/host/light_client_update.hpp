// LC-update ingestion: the data side of src/main.rs:56-175 and src/utils.rs:128-237 -- parse a light
// client update (the beacon-API V1_5 layout the reference fetches over RPC, or the layout of the two fixture files
// src/light_client_update_period_63{3,4}.json), native SSZ hash_tree_root of the containers the circuit re-computes,
// compute_domain / compute_signing_root, and the witness assembly that main.rs does before set_proof_target.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>
#include "gadgets.hpp"

namespace lc {

using H256 = std::array<uint8_t, 32>;

void sha256_two_to_one_native(const uint8_t left[32], const uint8_t right[32], uint8_t out[32]);  // sha256(left || right)

struct BeaconBlockHeader {  // eth_types::eth2::BeaconBlockHeader
  uint64_t slot = 0, proposer_index = 0;
  H256 parent_root{}, state_root{}, body_root{};
  H256 tree_hash_root() const;
};
struct SyncCommittee {
  std::vector<std::array<uint8_t, G1_PUBKEY_SIZE>> pubkeys;
  std::array<uint8_t, G1_PUBKEY_SIZE> aggregate_pubkey{};
  H256 tree_hash_root() const;  // what ssz_sync_committee computes in-circuit (src/sync_committee_pubkeys.rs:47-87)
};
struct SyncAggregate {
  std::vector<bool> sync_committee_bits;  // LSB-first per byte (src/utils.rs:112-126)
  std::array<uint8_t, 96> sync_committee_signature{};
};
struct LightClientUpdate {
  BeaconBlockHeader attested_header, finalized_header;
  std::vector<H256> finality_branch;
  SyncCommittee next_sync_committee;
  std::vector<H256> next_sync_committee_branch;
  SyncAggregate sync_aggregate;
  uint64_t signature_slot = 0;
};

enum class UpdateLayout { AUTO, V1_5, FIXTURE };
// throws std::runtime_error on malformed input (the reference unwrap()s)
LightClientUpdate parse_light_client_update(const std::string &json_text, UpdateLayout layout = UpdateLayout::AUTO);

struct NetworkConfig {  // eth2_utility NetworkConfig::new(&Network::Mainnet) as used at src/main.rs:79-84
  H256 genesis_validators_root{};
  struct Fork { uint64_t epoch; uint8_t version[4]; };
  std::vector<Fork> forks;  // ascending epochs
  static NetworkConfig mainnet();
  void fork_version_by_slot(uint64_t slot, uint8_t out[4]) const;
};
extern const uint8_t DOMAIN_SYNC_COMMITTEE[4];
H256 compute_domain(const uint8_t domain_type[4], const uint8_t fork_version[4], const H256 &genesis_validators_root);
H256 compute_signing_root(const H256 &object_root, const H256 &domain);  // src/utils.rs:229-237
H256 contract_state_root(uint64_t slot, const H256 &header, const H256 &sync_committee_i, const H256 &sync_committee_ii);

// Everything src/main.rs:84-175 derives from the previous and the current update, then set_proof_target.
// prev supplies the contract's current state (its finalized header and both committees) and the signing committee.
struct LightClientStep {
  H256 cur_state, new_state, signing_root, domain, attested_header_root, finalized_header_root;
  bool is_attested_from_next_period = false;
  size_t participation = 0;
};
LightClientStep set_light_client_step(PartialWitness &witness, const ProofTarget &target, const LightClientUpdate &prev,
                                      const LightClientUpdate &cur, const NetworkConfig &network);

}  // namespace lc
