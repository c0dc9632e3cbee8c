#include "light_client_update.hpp"
#include <cctype>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include "host_internal.hpp"

namespace lc {

// ---------------------------------------------------------------- native SHA-256 (FIPS 180-4) on whole 64-byte messages
static void sha256_compress_native(uint32_t st[8], const uint8_t block[64]) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = (uint32_t)block[4 * i] << 24 | (uint32_t)block[4 * i + 1] << 16 | (uint32_t)block[4 * i + 2] << 8 | block[4 * i + 3];
  auto rotr = [](uint32_t x, int r) { return (x >> r) | (x << (32 - r)); };
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
  for (int i = 0; i < 64; i++) {
    uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + w[i];
    uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
void sha256_two_to_one_native(const uint8_t left[32], const uint8_t right[32], uint8_t out[32]) {
  uint32_t st[8];
  memcpy(st, SHA_IV, sizeof st);
  uint8_t block[64];
  memcpy(block, left, 32);
  memcpy(block + 32, right, 32);
  sha256_compress_native(st, block);
  memset(block, 0, 64);
  block[0] = 0x80;
  block[62] = 0x02;  // message length 512 bits
  sha256_compress_native(st, block);
  for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16); out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i]; }
}
static H256 h2(const H256 &l, const H256 &r) { H256 o; sha256_two_to_one_native(l.data(), r.data(), o.data()); return o; }
static H256 merkleize(std::vector<H256> layer) {  // power-of-two leaf count
  while (layer.size() > 1) {
    std::vector<H256> up(layer.size() / 2);
    for (size_t i = 0; i < up.size(); i++) up[i] = h2(layer[2 * i], layer[2 * i + 1]);
    layer.swap(up);
  }
  return layer[0];
}
static H256 u64_leaf(uint64_t v) { H256 o{}; for (int i = 0; i < 8; i++) o[i] = (uint8_t)(v >> (8 * i)); return o; }

H256 BeaconBlockHeader::tree_hash_root() const {
  return merkleize({u64_leaf(slot), u64_leaf(proposer_index), parent_root, state_root, body_root, H256{}, H256{}, H256{}});
}
static H256 pubkey_leaf(const std::array<uint8_t, G1_PUBKEY_SIZE> &pk) {  // hash_tree_root(Bytes48): two 32-byte chunks
  H256 lo, hi{};
  memcpy(lo.data(), pk.data(), 32);
  memcpy(hi.data(), pk.data() + 32, 16);
  return h2(lo, hi);
}
H256 SyncCommittee::tree_hash_root() const {
  if (pubkeys.size() != SYNC_COMMITTEE_SIZE) throw std::runtime_error("sync committee: expected 512 pubkeys");
  std::vector<H256> leaves;
  for (auto &pk : pubkeys) leaves.push_back(pubkey_leaf(pk));
  return h2(merkleize(leaves), pubkey_leaf(aggregate_pubkey));
}

const uint8_t DOMAIN_SYNC_COMMITTEE[4] = {7, 0, 0, 0};
H256 compute_domain(const uint8_t domain_type[4], const uint8_t fork_version[4], const H256 &genesis_validators_root) {
  H256 version{};
  memcpy(version.data(), fork_version, 4);
  H256 fork_data_root = h2(version, genesis_validators_root);  // hash_tree_root(ForkData)
  H256 d;
  memcpy(d.data(), domain_type, 4);
  memcpy(d.data() + 4, fork_data_root.data(), 28);
  return d;
}
H256 compute_signing_root(const H256 &object_root, const H256 &domain) { return h2(object_root, domain); }
H256 contract_state_root(uint64_t slot, const H256 &header, const H256 &sc_i, const H256 &sc_ii) {
  return h2(h2(u64_leaf(slot), header), h2(sc_i, sc_ii));
}

static H256 hex32(const char *s) {
  H256 o;
  for (int i = 0; i < 32; i++) { unsigned v; sscanf(s + 2 * i, "%2x", &v); o[i] = (uint8_t)v; }
  return o;
}
NetworkConfig NetworkConfig::mainnet() {
  NetworkConfig c;
  c.genesis_validators_root = hex32("4b363db94e286120d76eb905340fdd4e54bfe9f06bf33ff6cf5ad27f511bfe95");
  c.forks = {{0, {0, 0, 0, 0}}, {74240, {1, 0, 0, 0}}, {144896, {2, 0, 0, 0}}, {194048, {3, 0, 0, 0}}, {269568, {4, 0, 0, 0}}};
  return c;
}
void NetworkConfig::fork_version_by_slot(uint64_t slot, uint8_t out[4]) const {
  const uint64_t epoch = slot / 32;
  const Fork *f = &forks[0];
  for (auto &k : forks) if (epoch >= k.epoch) f = &k;
  memcpy(out, f->version, 4);
}

// ---------------------------------------------------------------- minimal JSON reader (objects, arrays, strings, numbers, literals)
namespace {
struct JVal {
  enum Kind { NUL, STR, NUM, ARR, OBJ, LIT } kind = NUL;
  std::string s;  // STR: text, NUM/LIT: token
  std::vector<std::unique_ptr<JVal>> a;
  std::map<std::string, std::unique_ptr<JVal>> o;
  const JVal *get(const std::string &k) const { auto it = o.find(k); return it == o.end() ? nullptr : it->second.get(); }
  const JVal &at(const std::string &k) const {
    const JVal *v = kind == OBJ ? get(k) : nullptr;
    if (!v) throw std::runtime_error("light client update json: missing field '" + k + "'");
    return *v;
  }
};
struct JParser {
  const std::string &t;
  size_t i = 0;
  explicit JParser(const std::string &text) : t(text) {}
  void ws() { while (i < t.size() && isspace((unsigned char)t[i])) i++; }
  [[noreturn]] void fail(const char *what) { throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(i)); }
  std::string str() {
    if (t[i] != '"') fail("expected string");
    std::string r;
    for (i++; i < t.size() && t[i] != '"'; i++) {
      if (t[i] == '\\') { if (++i >= t.size()) fail("bad escape"); r += t[i] == 'n' ? '\n' : t[i] == 't' ? '\t' : t[i]; }
      else r += t[i];
    }
    if (i >= t.size()) fail("unterminated string");
    i++;
    return r;
  }
  std::unique_ptr<JVal> value() {
    ws();
    if (i >= t.size()) fail("unexpected end");
    auto v = std::make_unique<JVal>();
    char c = t[i];
    if (c == '{') {
      v->kind = JVal::OBJ;
      i++; ws();
      if (t[i] == '}') { i++; return v; }
      for (;;) {
        ws();
        std::string k = str();
        ws();
        if (t[i] != ':') fail("expected ':'");
        i++;
        v->o[k] = value();
        ws();
        if (t[i] == ',') { i++; continue; }
        if (t[i] == '}') { i++; break; }
        fail("expected ',' or '}'");
      }
    } else if (c == '[') {
      v->kind = JVal::ARR;
      i++; ws();
      if (t[i] == ']') { i++; return v; }
      for (;;) {
        v->a.push_back(value());
        ws();
        if (t[i] == ',') { i++; continue; }
        if (t[i] == ']') { i++; break; }
        fail("expected ',' or ']'");
      }
    } else if (c == '"') {
      v->kind = JVal::STR;
      v->s = str();
    } else {
      size_t j = i;
      while (j < t.size() && (isalnum((unsigned char)t[j]) || t[j] == '-' || t[j] == '+' || t[j] == '.')) j++;
      if (j == i) fail("unexpected character");
      v->s = t.substr(i, j - i);
      v->kind = (isdigit((unsigned char)v->s[0]) || v->s[0] == '-') ? JVal::NUM : JVal::LIT;
      i = j;
    }
    return v;
  }
};

uint64_t as_u64(const JVal &v) {  // the beacon API quotes integers, the fixture files do not
  if (v.kind != JVal::STR && v.kind != JVal::NUM) throw std::runtime_error("light client update json: expected an integer");
  size_t pos = 0;
  uint64_t r = std::stoull(v.s, &pos, 10);
  if (pos != v.s.size()) throw std::runtime_error("light client update json: bad integer '" + v.s + "'");
  return r;
}
std::vector<uint8_t> as_bytes(const JVal &v, size_t expect) {
  if (v.kind != JVal::STR) throw std::runtime_error("light client update json: expected a hex string");
  const std::string &s = v.s;
  size_t off = (s.size() >= 2 && s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) ? 2 : 0;
  if ((s.size() - off) % 2) throw std::runtime_error("light client update json: odd hex length");
  std::vector<uint8_t> out((s.size() - off) / 2);
  auto nib = [](char c) -> int { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; };
  for (size_t k = 0; k < out.size(); k++) {
    int hi = nib(s[off + 2 * k]), lo = nib(s[off + 2 * k + 1]);
    if (hi < 0 || lo < 0) throw std::runtime_error("light client update json: bad hex digit");
    out[k] = (uint8_t)(hi << 4 | lo);
  }
  if (expect && out.size() != expect) throw std::runtime_error("light client update json: expected " + std::to_string(expect) + " bytes, got " + std::to_string(out.size()));
  return out;
}
H256 as_h256(const JVal &v) { H256 o; auto b = as_bytes(v, 32); memcpy(o.data(), b.data(), 32); return o; }
BeaconBlockHeader as_header(const JVal &v) {
  BeaconBlockHeader h;
  h.slot = as_u64(v.at("slot"));
  h.proposer_index = as_u64(v.at("proposer_index"));
  h.parent_root = as_h256(v.at("parent_root"));
  h.state_root = as_h256(v.at("state_root"));
  h.body_root = as_h256(v.at("body_root"));
  return h;
}
std::vector<H256> as_branch(const JVal &v) {
  if (v.kind != JVal::ARR) throw std::runtime_error("light client update json: expected an array of roots");
  std::vector<H256> r;
  for (auto &e : v.a) r.push_back(as_h256(*e));
  return r;
}
SyncCommittee as_committee(const JVal &v) {
  SyncCommittee c;
  const JVal &pks = v.at("pubkeys");
  if (pks.kind != JVal::ARR) throw std::runtime_error("light client update json: pubkeys is not an array");
  for (auto &e : pks.a) { std::array<uint8_t, G1_PUBKEY_SIZE> pk; auto b = as_bytes(*e, G1_PUBKEY_SIZE); memcpy(pk.data(), b.data(), G1_PUBKEY_SIZE); c.pubkeys.push_back(pk); }
  auto agg = as_bytes(v.at("aggregate_pubkey"), G1_PUBKEY_SIZE);
  memcpy(c.aggregate_pubkey.data(), agg.data(), G1_PUBKEY_SIZE);
  return c;
}
SyncAggregate as_aggregate(const JVal &v) {
  SyncAggregate a;
  auto bits = as_bytes(v.at("sync_committee_bits"), SYNC_COMMITTEE_SIZE / 8);
  for (size_t k = 0; k < SYNC_COMMITTEE_SIZE; k++) a.sync_committee_bits.push_back((bits[k / 8] >> (k % 8)) & 1);
  auto sig = as_bytes(v.at("sync_committee_signature"), 96);
  memcpy(a.sync_committee_signature.data(), sig.data(), 96);
  return a;
}
}  // namespace

LightClientUpdate parse_light_client_update(const std::string &text, UpdateLayout layout) {
  JParser P(text);
  std::unique_ptr<JVal> root = P.value();
  P.ws();
  if (P.i != text.size()) P.fail("trailing characters");
  const JVal *top = root.get();
  if (top->kind == JVal::ARR && top->a.size() == 1) top = top->a[0].get();  // the RPC answers with a list of updates
  if (top->kind != JVal::OBJ) throw std::runtime_error("light client update json: top level is not an object");
  if (layout == UpdateLayout::AUTO) layout = top->get("data") ? UpdateLayout::V1_5 : UpdateLayout::FIXTURE;
  LightClientUpdate u;
  if (layout == UpdateLayout::V1_5) {  // src/utils.rs:128-227
    const JVal &d = top->at("data");
    u.attested_header = as_header(d.at("attested_header").at("beacon"));
    u.finalized_header = as_header(d.at("finalized_header").at("beacon"));
    u.finality_branch = as_branch(d.at("finality_branch"));
    u.next_sync_committee = as_committee(d.at("next_sync_committee"));
    u.next_sync_committee_branch = as_branch(d.at("next_sync_committee_branch"));
    u.sync_aggregate = as_aggregate(d.at("sync_aggregate"));
    if (d.get("signature_slot")) u.signature_slot = as_u64(d.at("signature_slot"));
  } else {  // serde layout of eth_types::eth2::LightClientUpdate (the two fixture files)
    u.attested_header = as_header(top->at("attested_beacon_header"));
    const JVal &fu = top->at("finality_update");
    u.finalized_header = as_header(fu.at("header_update").at("beacon_header"));
    u.finality_branch = as_branch(fu.at("finality_branch"));
    const JVal &su = top->at("sync_committee_update");
    u.next_sync_committee = as_committee(su.at("next_sync_committee"));
    u.next_sync_committee_branch = as_branch(su.at("next_sync_committee_branch"));
    u.sync_aggregate = as_aggregate(top->at("sync_aggregate"));
    if (top->get("signature_slot")) u.signature_slot = as_u64(top->at("signature_slot"));
  }
  if (u.finality_branch.size() != FINALIZED_HEADER_HEIGHT) throw std::runtime_error("light client update: finality branch must have 6 nodes");
  if (u.next_sync_committee_branch.size() != SYNC_COMMITTEE_HEIGHT) throw std::runtime_error("light client update: next sync committee branch must have 5 nodes");
  if (u.next_sync_committee.pubkeys.size() != SYNC_COMMITTEE_SIZE) throw std::runtime_error("light client update: next sync committee must have 512 pubkeys");
  return u;
}

LightClientStep set_light_client_step(PartialWitness &witness, const ProofTarget &target, const LightClientUpdate &prev,
                                      const LightClientUpdate &cur, const NetworkConfig &network) {
  LightClientStep st;
  uint8_t version[4];
  network.fork_version_by_slot(cur.attested_header.slot, version);
  st.domain = compute_domain(DOMAIN_SYNC_COMMITTEE, version, network.genesis_validators_root);
  st.attested_header_root = cur.attested_header.tree_hash_root();
  st.finalized_header_root = cur.finalized_header.tree_hash_root();
  st.signing_root = compute_signing_root(st.attested_header_root, st.domain);
  // the contract's state before the step (src/main.rs:113-131): slot/header of prev's finalized block, committee i = the
  // sibling of prev.next_sync_committee in its branch (= prev's current committee root), committee ii = root(prev.next)
  const uint64_t cur_slot = prev.finalized_header.slot;
  const H256 cur_header = prev.finalized_header.tree_hash_root();
  const H256 cur_i = prev.next_sync_committee_branch[0], cur_ii = prev.next_sync_committee.tree_hash_root();
  const H256 new_i = cur.next_sync_committee_branch[0], new_ii = cur.next_sync_committee.tree_hash_root();
  st.cur_state = contract_state_root(cur_slot, cur_header, cur_i, cur_ii);
  st.new_state = contract_state_root(cur.finalized_header.slot, st.finalized_header_root, new_i, new_ii);
  const uint64_t attested_period = cur.attested_header.slot / 8192, cur_period = cur_slot / 8192;
  st.is_attested_from_next_period = attested_period == cur_period + 1;
  for (bool b : cur.sync_aggregate.sync_committee_bits) st.participation += b;
  uint8_t finality_branch[6][32], sc_branch[5][32];
  for (int i = 0; i < 6; i++) memcpy(finality_branch[i], cur.finality_branch[i].data(), 32);
  for (int i = 0; i < 5; i++) memcpy(sc_branch[i], cur.next_sync_committee_branch[i].data(), 32);
  // the signing committee is prev.next_sync_committee (src/main.rs:133-150)
  std::vector<uint8_t> pubkeys(SYNC_COMMITTEE_SIZE * G1_PUBKEY_SIZE);
  for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++) memcpy(&pubkeys[i * G1_PUBKEY_SIZE], prev.next_sync_committee.pubkeys[i].data(), G1_PUBKEY_SIZE);
  set_proof_target(witness, st.signing_root.data(), st.domain.data(), cur.attested_header.slot, cur.attested_header.proposer_index,
                   st.attested_header_root.data(), cur.attested_header.parent_root.data(), cur.attested_header.state_root.data(),
                   cur.attested_header.body_root.data(), cur.finalized_header.slot, cur.finalized_header.proposer_index,
                   st.finalized_header_root.data(), cur.finalized_header.parent_root.data(), cur.finalized_header.state_root.data(),
                   cur.finalized_header.body_root.data(), finality_branch, st.cur_state.data(), st.new_state.data(), cur_slot, cur_header.data(),
                   cur_i.data(), cur_ii.data(), new_i.data(), new_ii.data(), cur.sync_aggregate.sync_committee_bits, sc_branch,
                   reinterpret_cast<const uint8_t(*)[G1_PUBKEY_SIZE]>(pubkeys.data()), prev.next_sync_committee.aggregate_pubkey.data(),
                   cur.sync_aggregate.sync_committee_signature.data(), target);
  return st;
}

}  // namespace lc
