// Host-side protocol pieces shared by the prover orchestration and the verifier:
// Fiat-Shamir challenger (plonky2 iop/challenger.rs), Poseidon sponge modes used on the host
// (a few hundred permutations per proof: SURVEY.md row a14), and the flat proof layout.
#pragma once
#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>
#include "../../include/lcp2.h"
#include "poseidon.hpp"

namespace lcp2 {

class HostPoseidon {
 public:
  static HostPoseidon &get() {
    static HostPoseidon h;
    return h;
  }
  void permute(u64 s[12]) const { pos_permute(s, rc_); }
  void hash_no_pad(const u64 *in, size_t len, u64 out[4]) const {
    u64 s[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
      size_t c = std::min<size_t>(8, len - off);
      for (size_t i = 0; i < c; i++) s[i] = gl_canon(in[off + i]);
      permute(s);
    }
    memcpy(out, s, 32);
  }
  void hash_or_noop(const u64 *in, size_t len, u64 out[4]) const {
    if (len <= 4) { for (size_t i = 0; i < 4; i++) out[i] = i < len ? gl_canon(in[i]) : 0; }
    else hash_no_pad(in, len, out);
  }
  void two_to_one(const u64 l[4], const u64 r[4], u64 out[4]) const {
    u64 s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
    permute(s);
    memcpy(out, s, 32);
  }
  // verify_merkle_proof_to_cap
  bool merkle_verify(const u64 *leaf, size_t leaf_len, u64 index, const u64 *siblings, u32 nsib, const u64 *cap) const {
    u64 cur[4], nxt[4];
    hash_or_noop(leaf, leaf_len, cur);
    for (u32 k = 0; k < nsib; k++) {
      if (index & 1) two_to_one(siblings + 4 * k, cur, nxt); else two_to_one(cur, siblings + 4 * k, nxt);
      memcpy(cur, nxt, 32);
      index >>= 1;
    }
    return memcmp(cur, cap + 4 * index, 32) == 0;
  }

 private:
  HostPoseidon() { pos_derive_round_constants(rc_); }
  u64 rc_[POS_ROUNDS * POS_W];
};

// Challenger<F, PoseidonHash>: duplex sponge in overwrite mode, rate 8, challenges popped from the back
class HostChallenger {
 public:
  void observe(u64 x) {
    nout_ = 0;
    in_[nin_++] = gl_canon(x);
    if (nin_ == 8) duplex();
  }
  void observe_n(const u64 *x, size_t n) { for (size_t i = 0; i < n; i++) observe(x[i]); }
  u64 get() {
    if (nin_ || !nout_) duplex();
    return out_[--nout_];
  }
  gl2 get_ext() { u64 a = get(), b = get(); return gl2_make(a, b); }
  // fri_proof_of_work: the duplex state with the pending inputs written, and the slot of the witness
  void pow_state(u64 state[12], u32 &pos) const {
    memcpy(state, s_, sizeof s_);
    for (int i = 0; i < nin_; i++) state[i] = in_[i];
    pos = (u32)nin_;
  }
  // plonky2's Challenger { sponge_state, input_buffer, output_buffer } for the staged C ABI (lcp2_challenger)
  void save(u64 sponge[12], u64 input[8], uint32_t &input_len, u64 output[8], uint32_t &output_len) const {
    memcpy(sponge, s_, sizeof s_); memcpy(input, in_, sizeof in_); memcpy(output, out_, sizeof out_);
    input_len = (uint32_t)nin_; output_len = (uint32_t)nout_;
  }
  void load(const u64 sponge[12], const u64 input[8], uint32_t input_len, const u64 output[8], uint32_t output_len) {
    memcpy(s_, sponge, sizeof s_); memcpy(in_, input, sizeof in_); memcpy(out_, output, sizeof out_);
    nin_ = (int)input_len; nout_ = (int)output_len;
  }

 private:
  void duplex() {
    for (int i = 0; i < nin_; i++) s_[i] = in_[i];
    nin_ = 0;
    HostPoseidon::get().permute(s_);
    memcpy(out_, s_, 64);
    nout_ = 8;
  }
  u64 s_[12] = {0}, in_[8] = {0}, out_[8] = {0};
  int nin_ = 0, nout_ = 0;
};

// Flat proof layout in u64 words; field order of plonky2's Proof / FriProof (plonk/proof.rs, fri/proof.rs)
struct ProofLayout {
  size_t capw, lgN;
  size_t wires_cap, zs_cap, quot_cap;
  size_t op_constants, op_sigmas, op_wires, op_zs, op_zs_next, op_pp, op_quot;
  size_t fri_caps, queries, query_words;
  size_t q_init_off[4], q_init_cols[4], q_init_sib;
  size_t q_step_off[LCP2_MAX_FRI_LAYERS], q_step_sib[LCP2_MAX_FRI_LAYERS];
  size_t final_poly, final_len, pow_witness, total;

  explicit ProofLayout(const lcp2_params &p) {
    const size_t CH = p.num_challenges, NR = p.num_routed_wires, NC = p.num_constants, W = p.num_wires, Q = p.quotient_degree_factor;
    const size_t npp = (NR + Q - 1) / Q - 1;
    capw = (size_t)4 << p.cap_height;
    lgN = p.degree_bits + p.rate_bits;
    size_t o = 0;
    wires_cap = o; o += capw; zs_cap = o; o += capw; quot_cap = o; o += capw;
    op_constants = o; o += 2 * NC; op_sigmas = o; o += 2 * NR; op_wires = o; o += 2 * W;
    op_zs = o; o += 2 * CH; op_zs_next = o; o += 2 * CH; op_pp = o; o += 2 * CH * npp; op_quot = o; o += 2 * CH * Q;
    fri_caps = o; o += p.num_fri_layers * capw;
    const size_t cols[4] = {NC + NR, W, CH * (1 + npp), CH * Q};
    size_t q = 0;
    q_init_sib = lgN - p.cap_height;
    for (int i = 0; i < 4; i++) { q_init_off[i] = q; q_init_cols[i] = cols[i]; q += cols[i] + 4 * q_init_sib; }
    size_t lg = lgN;
    for (u32 l = 0; l < LCP2_MAX_FRI_LAYERS; l++) { q_step_off[l] = 0; q_step_sib[l] = 0; }
    for (u32 l = 0; l < p.num_fri_layers; l++) {
      lg -= p.fri_arity_bits[l];
      q_step_off[l] = q;
      q_step_sib[l] = lg - p.cap_height;
      q += ((size_t)2 << p.fri_arity_bits[l]) + 4 * q_step_sib[l];
    }
    query_words = q;
    queries = o; o += q * p.num_query_rounds;
    size_t fl = p.degree_bits;
    for (u32 l = 0; l < p.num_fri_layers; l++) fl -= p.fri_arity_bits[l];
    final_len = (size_t)1 << fl;
    final_poly = o; o += 2 * final_len;
    pow_witness = o; o += 1;
    total = o;
  }
};

// what the verifier needs from a circuit (host data only)
struct VerifierView {
  const lcp2_params *p;
  u32 npi, num_selectors, num_gates;
  const lcp2_gate *gates;
  const uint32_t *code;
  const u64 *imm, *k_is, *digest, *cs_cap;
};

}  // namespace lcp2
