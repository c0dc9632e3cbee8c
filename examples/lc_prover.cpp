// The reference's main() (eth-lc-plonky2/src/main.rs:30-233) on this backend: two consecutive light-client updates in,
// one proof of the contract-state transition out.  The RPC fetch of main.rs:33-56 is replaced by two files (the beacon
// API V1_5 layout or the layout of the reference's fixture files); the recursive BLS verifier is stubbed.
//   lc_prover <prev_update.json> <cur_update.json> [--witness-only] [--device N] [--repeat K] [--extra-committees C]
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
  if (argc < 3) { fprintf(stderr, "usage: %s <prev_update.json> <cur_update.json> [--witness-only] [--device N] [--repeat K]\n", argv[0]); return 2; }
  bool witness_only = false;
  int device = 0, repeat = 1, extra = 0;
  for (int i = 3; i < argc; i++) {
    if (!strcmp(argv[i], "--witness-only")) witness_only = true;
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--extra-committees") && i + 1 < argc) extra = atoi(argv[++i]);
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  try {
    auto t0 = std::chrono::steady_clock::now();
    const LightClientUpdate prev = parse_light_client_update(slurp(argv[1])), cur = parse_light_client_update(slurp(argv[2]));
    printf("updates parsed in %.1f ms: prev finalized slot %llu, attested slot %llu, finalized slot %llu\n", ms_since(t0),
           (unsigned long long)prev.finalized_header.slot, (unsigned long long)cur.attested_header.slot, (unsigned long long)cur.finalized_header.slot);

    t0 = std::chrono::steady_clock::now();
    CircuitBuilder builder(CircuitConfig::standard_recursion_config());
    ProofTarget target = add_virtual_proof_target(builder);
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
    lcp2_ctx *ctx = nullptr;
    int rc = lcp2_ctx_create(device, nullptr, &ctx);
    if (rc != LCP2_OK) { fprintf(stderr, "lcp2_ctx_create(device %d): %s\n", device, lcp2_status_str(rc)); return 3; }
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
    lcp2_ctx_destroy(ctx);
    return 0;
  } catch (const std::exception &e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
