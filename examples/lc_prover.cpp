// The reference's main() (eth-lc-plonky2/src/main.rs:30-233) on this backend: two consecutive light-client updates in,
// one proof of the contract-state transition out.  The RPC fetch of main.rs:33-56 is replaced by two files (the beacon
// API V1_5 layout or the layout of the reference's fixture files).
//   lc_prover <prev_update.json> <cur_update.json> [--witness-only] [--device N] [--repeat K] [--extra-committees C] [--bls-proof-stand-in]
// Without --bls-proof-stand-in the recursive BLS verifier (src/targets.rs:468-482) is left out.  With it the circuit is the
// reference's in full shape: a proof with the BLS proof's 25 216 public inputs is produced first (of the STAND-IN statement
// circuit of host/gadgets.hpp - it proves nothing about the signature; the BLS12-381 verifier is out of scope) and the
// light-client circuit verifies it recursively and ties its public inputs to the signing root, signature, committee and bits.
// --sync-committee-only proves the SyncCommitteeSSZ gadget alone (BASELINE configs[1]; the reference's test_ssz_sync_committee,
// src/sync_committee_pubkeys.rs:100-653): the 512 pubkeys and the aggregate key of <cur_update>'s next_sync_committee hashed to
// their SSZ root (1 025 two_to_one_sha256), the root a public input checked against the native SSZ root.
// --extra-committees C adds C more SyncCommitteeSSZ gadgets (1 025 two_to_one_sha256 = 317 750 rows each) on the update's own
// committee: with C = 6 the circuit has 7 x 1 025 + 32 hashes and 2^22 rows, the reference's scale, made of real gadgets.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include "../eth-lc-plonky2_amd/host/light_client_update.hpp"

using namespace lc;

static std::string slurp(const char *path) {
  std::ifstream f(path);
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
static std::string hex(const H256 &h) {
  static const char *d = "0123456789abcdef";
  std::string s = "0x";
  for (uint8_t b : h) { s += d[b >> 4]; s += d[b & 15]; }
  return s;
}
static double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <prev_update.json> <cur_update.json> [--witness-only] [--device N] [--repeat K] [--extra-committees C] [--bls-proof-stand-in] [--sync-committee-only]\n", argv[0]); return 2; }
  bool witness_only = false, bls = false, ssz_only = false;
  int device = 0, repeat = 1, extra = 0;
  for (int i = 3; i < argc; i++) {
    if (!strcmp(argv[i], "--witness-only")) witness_only = true;
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--extra-committees") && i + 1 < argc) extra = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--bls-proof-stand-in")) bls = true;
    else if (!strcmp(argv[i], "--sync-committee-only")) ssz_only = true;
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  try {
    auto t0 = std::chrono::steady_clock::now();
    const LightClientUpdate prev = parse_light_client_update(slurp(argv[1])), cur = parse_light_client_update(slurp(argv[2]));
    printf("updates parsed in %.1f ms: prev finalized slot %llu, attested slot %llu, finalized slot %llu\n", ms_since(t0),
           (unsigned long long)prev.finalized_header.slot, (unsigned long long)cur.attested_header.slot, (unsigned long long)cur.finalized_header.slot);

    lcp2_ctx *ctx = nullptr;
    if (!witness_only) {
      int rc = lcp2_ctx_create(device, nullptr, &ctx);
      if (rc != LCP2_OK) { fprintf(stderr, "lcp2_ctx_create(device %d): %s\n", device, lcp2_status_str(rc)); return 3; }
    }
    if (ssz_only) {  // BASELINE configs[1]
      if (witness_only) { fprintf(stderr, "--sync-committee-only needs the GPU\n"); return 2; }
      t0 = std::chrono::steady_clock::now();
      CircuitBuilder builder(CircuitConfig::standard_recursion_config());
      SyncCommitteeTarget sc = add_virtual_sync_committee_target(builder);
      Hash256Target root = ssz_sync_committee(builder, sc);
      for (auto &limb : root) builder.register_public_input(limb.t);
      const size_t gates = builder.num_gates();
      auto data = builder.build();
      printf("circuit built in %.1f ms: %zu gates, degree_bits %u\n", ms_since(t0), gates, data->degree_bits());
      PartialWitness pw;
      for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++)
        pw.set_target_arr(sc.pubkeys[i], std::vector<F>(cur.next_sync_committee.pubkeys[i].begin(), cur.next_sync_committee.pubkeys[i].end()));
      pw.set_target_arr(sc.aggregate_pubkey, std::vector<F>(cur.next_sync_committee.aggregate_pubkey.begin(), cur.next_sync_committee.aggregate_pubkey.end()));
      const H256 want = cur.next_sync_committee.tree_hash_root();
      data->attach_gpu(ctx);
      const bool prof = getenv("LCP2_PROF") != nullptr;
      for (int k = 0; k < repeat; k++) {
        if (prof && k == repeat - 1) { lcp2_prof_enable(ctx, 1); lcp2_prof_reset(ctx); }
        t0 = std::chrono::steady_clock::now();
        ProofWithPublicInputs proof = data->prove(pw);
        const double prove_ms = ms_since(t0);
        t0 = std::chrono::steady_clock::now();
        data->verify(proof);
        for (int w = 0; w < 8; w++) {  // the public inputs are the root's eight big-endian words
          const uint32_t word = (uint32_t)want[4 * w] << 24 | (uint32_t)want[4 * w + 1] << 16 | (uint32_t)want[4 * w + 2] << 8 | want[4 * w + 3];
          if (proof.public_inputs.size() != 8 || proof.public_inputs[w] != word) { fprintf(stderr, "error: the proved root is not the native SSZ root\n"); return 1; }
        }
        printf("proof %d: proved in %.1f ms (witness generation included), verified in %.1f ms, %zu proof words, %zu public inputs, root %s\n", k, prove_ms,
               ms_since(t0), proof.proof.size(), proof.public_inputs.size(), hex(want).c_str());
      }
      if (prof) {
        static const char *names[LCP2_K_COUNT] = {"intt", "lde", "leaf_hash", "merkle", "perm_z", "quotient", "openings", "fri", "pow", "sha256_witness", "other"};
        for (int f = 0; f < LCP2_K_COUNT; f++) {
          double ms = 0, bytes = 0;
          uint64_t launches = 0;
          lcp2_prof_get(ctx, f, &ms, &launches, &bytes);
          if (launches) printf("  %-15s %8.3f ms  (%llu scopes)\n", names[f], ms, (unsigned long long)launches);
        }
      }
      data.reset();
      lcp2_ctx_destroy(ctx);
      return 0;
    }
    // src/main.rs:170: the BLS-signature proof first; its common data shapes the recursive verifier
    BlsStatementStandIn bls_circuit;
    CommonCircuitData bls_cd;
    if (bls) {
      if (witness_only) { fprintf(stderr, "--bls-proof-stand-in needs the GPU (the inner proof has to be produced)\n"); return 2; }
      bls_circuit = build_bls_statement_stand_in();
      bls_cd = CommonCircuitData::of(bls_circuit.data->description());
    }
    t0 = std::chrono::steady_clock::now();
    CircuitBuilder builder(CircuitConfig::standard_recursion_config());
    ProofTarget target = add_virtual_proof_target(builder, bls ? &bls_cd : nullptr);
    for (auto &limb : target.cur_state) builder.register_public_input(limb.t);  // src/main.rs:180-187
    for (auto &limb : target.new_state) builder.register_public_input(limb.t);
    std::vector<SyncCommitteeTarget> more;
    for (int k = 0; k < extra; k++) {
      more.push_back(add_virtual_sync_committee_target(builder));
      ssz_sync_committee(builder, more.back());
    }
    const size_t gates = builder.num_gates();
    auto data = builder.build();
    printf("circuit built in %.1f ms: %zu gates, degree_bits %u\n", ms_since(t0), gates, data->degree_bits());

    PartialWitness pw;
    const LightClientStep st = set_light_client_step(pw, target, prev, cur, NetworkConfig::mainnet());
    for (const SyncCommitteeTarget &sc : more) {  // the extra trees hash the signing committee again
      for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++)
        pw.set_target_arr(sc.pubkeys[i], std::vector<F>(prev.next_sync_committee.pubkeys[i].begin(), prev.next_sync_committee.pubkeys[i].end()));
      pw.set_target_arr(sc.aggregate_pubkey, std::vector<F>(prev.next_sync_committee.aggregate_pubkey.begin(), prev.next_sync_committee.aggregate_pubkey.end()));
    }
    if (bls) {
      t0 = std::chrono::steady_clock::now();
      PartialWitness bpw;
      std::vector<uint8_t> pubkeys(SYNC_COMMITTEE_SIZE * G1_PUBKEY_SIZE);
      for (size_t i = 0; i < SYNC_COMMITTEE_SIZE; i++) memcpy(&pubkeys[i * G1_PUBKEY_SIZE], prev.next_sync_committee.pubkeys[i].data(), G1_PUBKEY_SIZE);
      set_bls_statement_stand_in(bpw, bls_circuit, st.signing_root.data(), cur.sync_aggregate.sync_committee_signature.data(),
                                 reinterpret_cast<const uint8_t(*)[G1_PUBKEY_SIZE]>(pubkeys.data()), cur.sync_aggregate.sync_committee_bits);
      bls_circuit.data->attach_gpu(ctx);
      const double build_ms = ms_since(t0);
      t0 = std::chrono::steady_clock::now();
      ProofWithPublicInputs inner = bls_circuit.data->prove(bpw);
      const double prove_ms = ms_since(t0);
      bls_circuit.data->verify(inner);
      uint64_t digest[4];
      std::vector<uint64_t> cap;
      bls_circuit.data->verifier_only_data(digest, cap);
      set_bls_proof_target(pw, target, inner, digest, cap);
      printf("inner proof (stand-in for the BLS-signature proof): 2^%u rows, %zu public inputs, build %.1f ms, inner prove %.1f ms, %zu words\n",
             bls_circuit.data->degree_bits(), inner.public_inputs.size(), build_ms, prove_ms, inner.proof.size());
    }
    printf("cur_state %s\nnew_state %s\nsigning_root %s\nparticipation %zu/512, attested from next period: %s\n", hex(st.cur_state).c_str(),
           hex(st.new_state).c_str(), hex(st.signing_root).c_str(), st.participation, st.is_attested_from_next_period ? "yes" : "no");

    if (witness_only) {  // no GPU: run the generators only (a conflicting witness throws, as prove() would return Err)
      t0 = std::chrono::steady_clock::now();
      std::vector<uint64_t> wires;
      std::vector<F> pis;
      data->generate_witness(pw, wires, pis);
      printf("witness generated on the host in %.1f ms (%zu public inputs); no proof without a GPU\n", ms_since(t0), pis.size());
      return 0;
    }
    t0 = std::chrono::steady_clock::now();
    data->attach_gpu(ctx);
    printf("constants/sigmas committed on the GPU in %.1f ms\n", ms_since(t0));
    const bool prof = getenv("LCP2_PROF") != nullptr;  // HIP-event time per kernel family of the last proof
    for (int k = 0; k < repeat; k++) {
      if (prof && k == repeat - 1) { lcp2_prof_enable(ctx, 1); lcp2_prof_reset(ctx); }
      t0 = std::chrono::steady_clock::now();
      ProofWithPublicInputs proof = data->prove(pw);
      const double prove_ms = ms_since(t0);
      t0 = std::chrono::steady_clock::now();
      data->verify(proof);
      printf("proof %d: proved in %.1f ms (witness generation included), verified in %.1f ms, %zu proof words, %zu public inputs\n", k, prove_ms,
             ms_since(t0), proof.proof.size(), proof.public_inputs.size());
    }
    if (prof) {
      static const char *names[LCP2_K_COUNT] = {"intt", "lde", "leaf_hash", "merkle", "perm_z", "quotient", "openings", "fri", "pow", "sha256_witness", "other"};
      double sum = 0;
      for (int f = 0; f < LCP2_K_COUNT; f++) {
        double ms = 0, bytes = 0;
        uint64_t launches = 0;
        lcp2_prof_get(ctx, f, &ms, &launches, &bytes);
        sum += ms;
        if (launches) printf("  %-15s %8.3f ms  (%llu scopes)\n", names[f], ms, (unsigned long long)launches);
      }
      printf("  %-15s %8.3f ms\n", "sum of kernels", sum);
    }
    data.reset();
    bls_circuit.data.reset();
    lcp2_ctx_destroy(ctx);
    return 0;
  } catch (const std::exception &e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
