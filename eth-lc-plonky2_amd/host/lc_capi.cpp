// C entry points over the host layer's light-client workloads (lc_capi.h): the flow of examples/lc_prover.cpp = the reference's
// main() (eth-lc-plonky2/src/main.rs:56-233), in-process for callers that are not C++.
#include "lc_capi.h"
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include "light_client_update.hpp"

using namespace lc;

namespace {
thread_local std::string g_error;
int fail(int status, const std::string &why) { g_error = why; return status; }
double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
}  // namespace

struct lch_session {
  std::unique_ptr<CircuitData> data;
  BlsStatementStandIn bls;           // kept alive: the inner circuit (and its device workspace) of LCH_BLS_PROOF_STAND_IN
  PartialWitness pw;
  lch_info info{};
  std::vector<uint64_t> expected;    // the public inputs a correct proof carries
};

extern "C" const char *lch_last_error(void) { return g_error.c_str(); }

extern "C" int lch_light_client_step_create(lcp2_ctx *ctx, const char *prev_json, const char *cur_json, uint32_t flags, uint32_t extra, lch_session **out) {
  if (!ctx || !prev_json || !cur_json || !out) return fail(LCP2_E_INVALID, "null argument");
  if (flags & ~(LCH_BLS_PROOF_STAND_IN | LCH_SYNC_COMMITTEE_ONLY)) return fail(LCP2_E_INVALID, "unknown flag");
  *out = nullptr;
  try {
    std::unique_ptr<lch_session> s(new lch_session());
    const LightClientUpdate prev = parse_light_client_update(prev_json), cur = parse_light_client_update(cur_json);
    auto words_of = [&](const H256 &h) {
      for (int w = 0; w < 8; w++) s->expected.push_back((uint32_t)h[4 * w] << 24 | (uint32_t)h[4 * w + 1] << 16 | (uint32_t)h[4 * w + 2] << 8 | h[4 * w + 3]);
    };
    auto set_committee = [&](const SyncCommitteeTarget &sc, const SyncCommittee &c) {
      for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++) s->pw.set_target_arr(sc.pubkeys[i], std::vector<F>(c.pubkeys[i].begin(), c.pubkeys[i].end()));
      s->pw.set_target_arr(sc.aggregate_pubkey, std::vector<F>(c.aggregate_pubkey.begin(), c.aggregate_pubkey.end()));
    };
    auto t0 = std::chrono::steady_clock::now();
    if (flags & LCH_SYNC_COMMITTEE_ONLY) {  // the reference's test_ssz_sync_committee (src/sync_committee_pubkeys.rs:100-653)
      CircuitBuilder builder(CircuitConfig::standard_recursion_config());
      SyncCommitteeTarget sc = add_virtual_sync_committee_target(builder);
      Hash256Target root = ssz_sync_committee(builder, sc);
      for (auto &limb : root) builder.register_public_input(limb.t);
      s->info.num_gates = builder.num_gates();
      s->data = builder.build();
      set_committee(sc, cur.next_sync_committee);
      words_of(cur.next_sync_committee.tree_hash_root());
    } else {
      CommonCircuitData bls_cd;
      const bool bls = (flags & LCH_BLS_PROOF_STAND_IN) != 0;
      if (bls) {  // src/main.rs:170: the BLS-signature proof first; its common data shapes the recursive verifier
        s->bls = build_bls_statement_stand_in();
        bls_cd = CommonCircuitData::of(s->bls.data->description());
        t0 = std::chrono::steady_clock::now();
      }
      CircuitBuilder builder(CircuitConfig::standard_recursion_config());
      ProofTarget target = add_virtual_proof_target(builder, bls ? &bls_cd : nullptr);
      for (auto &limb : target.cur_state) builder.register_public_input(limb.t);  // src/main.rs:180-187
      for (auto &limb : target.new_state) builder.register_public_input(limb.t);
      std::vector<SyncCommitteeTarget> more;
      for (uint32_t k = 0; k < extra; k++) {
        more.push_back(add_virtual_sync_committee_target(builder));
        ssz_sync_committee(builder, more.back());
      }
      s->info.num_gates = builder.num_gates();
      s->data = builder.build();
      const LightClientStep st = set_light_client_step(s->pw, target, prev, cur, NetworkConfig::mainnet());
      for (const SyncCommitteeTarget &sc : more) set_committee(sc, prev.next_sync_committee);  // the extra trees hash the signing committee again
      words_of(st.cur_state);
      words_of(st.new_state);
      s->info.build_ms = ms_since(t0);
      if (bls) {
        PartialWitness bpw;
        std::vector<uint8_t> pubkeys(SYNC_COMMITTEE_SIZE * G1_PUBKEY_SIZE);
        for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++) memcpy(&pubkeys[i * G1_PUBKEY_SIZE], prev.next_sync_committee.pubkeys[i].data(), G1_PUBKEY_SIZE);
        set_bls_statement_stand_in(bpw, s->bls, st.signing_root.data(), cur.sync_aggregate.sync_committee_signature.data(),
                                   reinterpret_cast<const uint8_t(*)[G1_PUBKEY_SIZE]>(pubkeys.data()), cur.sync_aggregate.sync_committee_bits);
        s->bls.data->attach_gpu(ctx);
        auto t1 = std::chrono::steady_clock::now();
        ProofWithPublicInputs inner = s->bls.data->prove(bpw);
        s->info.inner_prove_ms = ms_since(t1);
        s->bls.data->verify(inner);
        uint64_t digest[4];
        std::vector<uint64_t> cap;
        s->bls.data->verifier_only_data(digest, cap);
        set_bls_proof_target(s->pw, target, inner, digest, cap);
        s->info.inner_degree_bits = s->bls.data->degree_bits();
        s->info.inner_public_inputs = (uint32_t)inner.public_inputs.size();
      }
    }
    if (!s->info.build_ms) s->info.build_ms = ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    s->data->attach_gpu(ctx);
    s->info.attach_ms = ms_since(t0);
    s->info.degree_bits = s->data->degree_bits();
    s->info.num_public_inputs = s->data->description().num_public_inputs;
    s->info.proof_words = lcp2_proof_words(&s->data->description().params);
    *out = s.release();
    return LCP2_OK;
  } catch (const UnsatisfiedError &e) {
    return fail(LCP2_E_UNSAT, e.what());
  } catch (const std::bad_alloc &) {
    return fail(LCP2_E_OOM, "out of host memory");
  } catch (const std::exception &e) {
    return fail(LCP2_E_INVALID, e.what());
  }
}

extern "C" void lch_destroy(lch_session *s) { delete s; }

extern "C" int lch_get_info(const lch_session *s, lch_info *out) {
  if (!s || !out) return fail(LCP2_E_INVALID, "null argument");
  *out = s->info;
  return LCP2_OK;
}

extern "C" int lch_prove(lch_session *s, uint64_t *proof, size_t proof_words, uint64_t *public_inputs, size_t num_public_inputs) {
  if (!s || !proof || (!public_inputs && num_public_inputs)) return fail(LCP2_E_INVALID, "null argument");
  if (proof_words != s->info.proof_words || num_public_inputs != s->info.num_public_inputs) return fail(LCP2_E_INVALID, "buffer lengths do not match lch_get_info");
  try {
    const ProofWithPublicInputs p = s->data->prove(s->pw);
    memcpy(proof, p.proof.data(), proof_words * 8);
    if (num_public_inputs) memcpy(public_inputs, p.public_inputs.data(), num_public_inputs * 8);
    return LCP2_OK;
  } catch (const UnsatisfiedError &e) {
    return fail(LCP2_E_UNSAT, e.what());
  } catch (const std::exception &e) {
    return fail(LCP2_E_HIP, e.what());
  }
}

extern "C" int lch_verify(const lch_session *s, const uint64_t *proof, size_t proof_words, const uint64_t *public_inputs, size_t num_public_inputs) {
  if (!s || !proof || (!public_inputs && num_public_inputs)) return fail(LCP2_E_INVALID, "null argument");
  if (proof_words != s->info.proof_words || num_public_inputs != s->info.num_public_inputs) return fail(LCP2_E_INVALID, "buffer lengths do not match lch_get_info");
  try {
    ProofWithPublicInputs p;
    p.proof.assign(proof, proof + proof_words);
    p.public_inputs.assign(public_inputs, public_inputs + num_public_inputs);
    s->data->verify(p);
    return LCP2_OK;
  } catch (const VerifyError &e) {
    return fail(LCP2_E_VERIFY, e.what());
  } catch (const std::exception &e) {
    return fail(LCP2_E_INVALID, e.what());
  }
}

extern "C" int lch_expected_public_inputs(const lch_session *s, uint64_t *out, size_t count) {
  if (!s || !out || count != s->expected.size()) return fail(LCP2_E_INVALID, "count does not match");
  memcpy(out, s->expected.data(), count * 8);
  return LCP2_OK;
}
