// data.verify(proof): host-only verifier (reference call site eth-lc-plonky2/src/main.rs:233,
// src/unit_tests.rs:34).  Restates plonky2 0.1.4 plonk/verifier.rs::verify_with_challenges,
// plonk/vanishing_poly.rs::eval_vanishing_poly, plonk/get_challenges.rs and fri/verifier.rs
// (verify_fri_proof, fri_combine_initial, compute_evaluation).  Milliseconds of scalar work: it stays
// on the host exactly as in the reference; no device call is made here.
#include <vector>
#include "host_protocol.hpp"
#include "internal.hpp"

namespace lcp2 {
VerifierView verifier_view(const lcp2_circuit *c);
}
using namespace lcp2;

namespace {
inline gl2 rd2(const u64 *p) { return gl2_make(p[0], p[1]); }
inline gl2 base2(u64 x) { return gl2_make(x, 0); }

// gate programs over the extension field (evaluation at zeta)
inline gl2 sbox7_ext(gl2 x) { gl2 x2 = gl2_mul(x, x), x4 = gl2_mul(x2, x2), x3 = gl2_mul(x2, x); return gl2_mul(x3, x4); }
void eval_gates_ext(const VerifierView &v, const gl2 *wires, const gl2 *consts, const u64 *pis, const u64 *alphas, gl2 *out) {
  static const u64 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  const u32 CH = v.p->num_challenges;
  for (u32 k = 0; k < CH; k++) out[k] = base2(0);
  gl2 regs[64];
  for (u32 g = 0; g < v.num_gates; g++) {
    const lcp2_gate &G = v.gates[g];
    const bool fwd = (G.flags & LCP2_GATE_EMIT_FORWARD) != 0;
    gl2 acc[4] = {base2(0), base2(0), base2(0), base2(0)};
    u64 apow[4] = {1, 1, 1, 1};  // forward gates: running powers of alpha (the prover kernel uses the 1 / alpha Horner form)
    for (u32 pc = G.code_offset; pc < G.code_offset + G.code_len; pc++) {
      const u32 w0 = v.code[2 * pc], w1 = v.code[2 * pc + 1];
      const u32 op = w0 & 0xF, dst = (w0 >> 8) & 0xFF, ka = (w0 >> 16) & 0xF, kb = (w0 >> 20) & 0xF, ia = w1 & 0xFFFF, ib = w1 >> 16;
      auto fetch = [&](u32 k, u32 i) -> gl2 {
        switch (k) {
          case 0: return regs[i];
          case 1: return wires[i];
          case 2: return consts[v.num_selectors + i];
          case 3: return base2(v.imm[i]);
          default: return base2(pis[i]);
        }
      };
      if (op == LCP2_OP_PMDS) {
        gl2 in[12];
        for (int i = 0; i < 12; i++) in[i] = regs[ia + i];
        for (int r = 0; r < 12; r++) {
          gl2 t = base2(v.imm[ib + r]);
          if (r == 0) t = gl2_add(t, gl2_scale(in[0], 8));
          for (int i = 0; i < 12; i++) t = gl2_add(t, gl2_scale(in[(i + r) % 12], CIRC[i]));
          regs[dst + r] = t;
        }
        continue;
      }
      gl2 a = fetch(ka, ia);
      if (op == LCP2_OP_EMIT || op == LCP2_OP_EMITBOOL) {
        if (op == LCP2_OP_EMITBOOL) a = gl2_sub(gl2_mul(a, a), a);
        for (u32 k = 0; k < CH; k++) {
          if (fwd) { acc[k] = gl2_add(acc[k], gl2_scale(a, apow[k])); apow[k] = gl_mul(apow[k], alphas[k]); }
          else acc[k] = gl2_add(gl2_scale(acc[k], alphas[k]), a);
        }
        continue;
      }
      if (op == LCP2_OP_SBOX) { regs[dst] = sbox7_ext(a); continue; }
      gl2 b = fetch(kb, ib);
      switch (op) {
        case LCP2_OP_ADD: regs[dst] = gl2_add(a, b); break;
        case LCP2_OP_SUB: regs[dst] = gl2_sub(a, b); break;
        case LCP2_OP_MUL: regs[dst] = gl2_mul(a, b); break;
        case LCP2_OP_XOR: { gl2 ab = gl2_mul(a, b); regs[dst] = gl2_sub(gl2_sub(gl2_add(a, b), ab), ab); break; }
        case LCP2_OP_DBLADD: regs[dst] = gl2_add(gl2_add(a, a), b); break;
        default: regs[dst] = gl2_add(regs[dst], gl2_mul(a, b)); break;  // LCP2_OP_MULADD
      }
    }
    gl2 s = consts[G.selector_index], f = base2(1);
    for (u32 j = G.group_start; j < G.group_end; j++)
      if (j != G.selector_value) f = gl2_mul(f, gl2_sub(base2(j), s));
    if (v.num_selectors > 1) f = gl2_mul(f, gl2_sub(base2(0xFFFFFFFFull), s));
    for (u32 k = 0; k < CH; k++) out[k] = gl2_add(out[k], gl2_mul(f, acc[k]));
  }
}

// compute_evaluation: value at beta of the degree < arity interpolant through the coset of x
gl2 fri_compute_evaluation(u64 x, u64 within, u32 arity_bits, const gl2 *evals_bitrev, gl2 beta) {
  const u32 arity = 1u << arity_bits;
  const u64 g = gl_root_of_unity(arity_bits);
  gl2 ev[32];
  for (u32 i = 0; i < arity; i++) ev[bitrev32(i, arity_bits)] = evals_bitrev[i];
  const u64 coset_start = gl_mul(x, gl_pow(g, arity - bitrev32((u32)within, arity_bits)));
  u64 pts[32], y = 1;
  for (u32 i = 0; i < arity; i++) { pts[i] = gl_mul(coset_start, y); y = gl_mul(y, g); }
  gl2 acc = base2(0);
  for (u32 i = 0; i < arity; i++) {
    gl2 num = base2(1);
    u64 den = 1;
    for (u32 j = 0; j < arity; j++) {
      if (j == i) continue;
      num = gl2_mul(num, gl2_sub_base(beta, pts[j]));
      den = gl_mul(den, gl_sub(pts[i], pts[j]));
    }
    acc = gl2_add(acc, gl2_mul(ev[i], gl2_scale(num, gl_inv(den))));
  }
  return acc;
}

int verify_impl(const VerifierView &v, const u64 *proof, const u64 *pis_in) {
  const lcp2_params &p = *v.p;
  const u64 n = 1ull << p.degree_bits, N = n << p.rate_bits;
  const u32 W = p.num_wires, NR = p.num_routed_wires, NC = p.num_constants, CH = p.num_challenges, Q = p.quotient_degree_factor;
  const u32 nchunks = (NR + Q - 1) / Q, npp = nchunks - 1, lgN = p.degree_bits + p.rate_bits;
  const ProofLayout L(p);
  const HostPoseidon &H = HostPoseidon::get();
  for (size_t i = 0; i < L.total; i++)
    if (proof[i] >= GL_P) return 1;
  std::vector<u64> pis(std::max<u32>(v.npi, 1), 0);
  for (u32 i = 0; i < v.npi; i++) pis[i] = gl_canon(pis_in[i]);
  u64 pi_hash[4];
  H.hash_no_pad(pis.data(), v.npi, pi_hash);

  // ---- get_challenges
  HostChallenger ch;
  ch.observe_n(v.digest, 4);
  ch.observe_n(pi_hash, 4);
  ch.observe_n(proof + L.wires_cap, L.capw);
  u64 betas[4], gammas[4], alphas[4];
  for (u32 k = 0; k < CH; k++) betas[k] = ch.get();
  for (u32 k = 0; k < CH; k++) gammas[k] = ch.get();
  ch.observe_n(proof + L.zs_cap, L.capw);
  for (u32 k = 0; k < CH; k++) alphas[k] = ch.get();
  ch.observe_n(proof + L.quot_cap, L.capw);
  const gl2 zeta = ch.get_ext();
  ch.observe_n(proof + L.op_constants, 2 * (NC + NR + W));
  ch.observe_n(proof + L.op_zs, 2 * CH);
  ch.observe_n(proof + L.op_pp, 2 * CH * npp);
  ch.observe_n(proof + L.op_quot, 2 * CH * Q);
  ch.observe_n(proof + L.op_zs_next, 2 * CH);
  const gl2 fri_alpha = ch.get_ext();
  gl2 fri_betas[LCP2_MAX_FRI_LAYERS];
  for (u32 l = 0; l < p.num_fri_layers; l++) { ch.observe_n(proof + L.fri_caps + l * L.capw, L.capw); fri_betas[l] = ch.get_ext(); }
  ch.observe_n(proof + L.final_poly, 2 * L.final_len);
  ch.observe(proof[L.pow_witness]);
  if ((ch.get() >> (64 - p.proof_of_work_bits)) != 0) return 2;

  // ---- vanishing(zeta) = Z_H(zeta) * t(zeta)
  std::vector<gl2> ow(W), oc(NC + NR);
  for (u32 j = 0; j < W; j++) ow[j] = rd2(proof + L.op_wires + 2 * j);
  for (u32 j = 0; j < NC + NR; j++) oc[j] = rd2(proof + L.op_constants + 2 * j);
  gl2 zeta_n = zeta;
  for (u32 i = 0; i < p.degree_bits; i++) zeta_n = gl2_mul(zeta_n, zeta_n);
  const gl2 one = base2(1);
  const gl2 zh = gl2_sub(zeta_n, one);
  {
    const gl2 l0 = gl2_eq(zeta, one) ? one : gl2_mul(zh, gl2_inv(gl2_scale(gl2_sub(zeta, one), n % GL_P)));
    std::vector<gl2> terms;
    for (u32 k = 0; k < CH; k++) terms.push_back(gl2_mul(l0, gl2_sub(rd2(proof + L.op_zs + 2 * k), one)));
    for (u32 k = 0; k < CH; k++) {
      gl2 prev = rd2(proof + L.op_zs + 2 * k);
      for (u32 c = 0; c < nchunks; c++) {
        gl2 pn = one, pd = one;
        for (u32 j = c * Q; j < NR && j < (c + 1) * Q; j++) {
          pn = gl2_mul(pn, gl2_add_base(gl2_add(ow[j], gl2_scale(gl2_scale(zeta, v.k_is[j]), betas[k])), gammas[k]));
          pd = gl2_mul(pd, gl2_add_base(gl2_add(ow[j], gl2_scale(oc[NC + j], betas[k])), gammas[k]));
        }
        gl2 next = c < npp ? rd2(proof + L.op_pp + 2 * (k * npp + c)) : rd2(proof + L.op_zs_next + 2 * k);
        terms.push_back(gl2_sub(gl2_mul(prev, pn), gl2_mul(next, pd)));
        prev = next;
      }
    }
    gl2 gates[4];
    eval_gates_ext(v, ow.data(), oc.data(), pi_hash, alphas, gates);
    for (u32 k = 0; k < CH; k++) {
      gl2 acc = gates[k];
      for (size_t t = terms.size(); t-- > 0;) acc = gl2_add(gl2_scale(acc, alphas[k]), terms[t]);
      gl2 tq = base2(0);
      for (u32 j = Q; j-- > 0;) tq = gl2_add(gl2_mul(tq, zeta_n), rd2(proof + L.op_quot + 2 * (k * Q + j)));
      if (!gl2_eq(acc, gl2_mul(zh, tq))) return 3;
    }
  }
  // ---- FRI
  gl2 red0 = base2(0), red1 = base2(0);
  {
    std::vector<gl2> vals;
    for (u32 j = 0; j < NC + NR + W; j++) vals.push_back(rd2(proof + L.op_constants + 2 * j));
    for (u32 j = 0; j < CH; j++) vals.push_back(rd2(proof + L.op_zs + 2 * j));
    for (u32 j = 0; j < CH * npp; j++) vals.push_back(rd2(proof + L.op_pp + 2 * j));
    for (u32 j = 0; j < CH * Q; j++) vals.push_back(rd2(proof + L.op_quot + 2 * j));
    for (size_t j = vals.size(); j-- > 0;) red0 = gl2_add(gl2_mul(red0, fri_alpha), vals[j]);
    for (u32 j = CH; j-- > 0;) red1 = gl2_add(gl2_mul(red1, fri_alpha), rd2(proof + L.op_zs_next + 2 * j));
  }
  const gl2 g_zeta = gl2_scale(zeta, gl_root_of_unity(p.degree_bits));
  const gl2 alpha_ch = gl2_pow(fri_alpha, CH);
  const u64 *caps[4] = {v.cs_cap, proof + L.wires_cap, proof + L.zs_cap, proof + L.quot_cap};
  for (u32 q = 0; q < p.num_query_rounds; q++) {
    u64 x_index = ch.get() % N;
    const u64 *R = proof + L.queries + (size_t)q * L.query_words;
    for (int o = 0; o < 4; o++)
      if (!H.merkle_verify(R + L.q_init_off[o], L.q_init_cols[o], x_index, R + L.q_init_off[o] + L.q_init_cols[o], (u32)L.q_init_sib, caps[o]))
        return 4;
    u64 subgroup_x = gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(lgN), bitrev32((u32)x_index, lgN)));
    gl2 r0 = base2(0);
    for (int o = 3; o >= 0; o--)
      for (size_t j = L.q_init_cols[o]; j-- > 0;) r0 = gl2_add_base(gl2_mul(r0, fri_alpha), R[L.q_init_off[o] + j]);
    gl2 sum = gl2_mul(gl2_sub(r0, red0), gl2_inv(gl2_sub(base2(subgroup_x), zeta)));
    gl2 r1 = base2(0);
    for (u32 j = CH; j-- > 0;) r1 = gl2_add_base(gl2_mul(r1, fri_alpha), R[L.q_init_off[2] + j]);
    sum = gl2_add(gl2_mul(sum, alpha_ch), gl2_mul(gl2_sub(r1, red1), gl2_inv(gl2_sub(base2(subgroup_x), g_zeta))));
    gl2 old_eval = sum;
    u64 xi = x_index;
    for (u32 l = 0; l < p.num_fri_layers; l++) {
      const u32 ab = p.fri_arity_bits[l], arity = 1u << ab;
      gl2 evals[32];
      for (u32 j = 0; j < arity; j++) evals[j] = rd2(R + L.q_step_off[l] + 2 * j);
      const u64 coset_index = xi >> ab, within = xi & (arity - 1);
      if (!gl2_eq(evals[within], old_eval)) return 5;
      old_eval = fri_compute_evaluation(subgroup_x, within, ab, evals, fri_betas[l]);
      if (!H.merkle_verify(R + L.q_step_off[l], 2 * arity, coset_index, R + L.q_step_off[l] + 2 * arity, (u32)L.q_step_sib[l],
                           proof + L.fri_caps + l * L.capw))
        return 6;
      for (u32 i = 0; i < ab; i++) subgroup_x = gl_sqr(subgroup_x);
      xi = coset_index;
    }
    gl2 fv = base2(0);
    for (size_t j = L.final_len; j-- > 0;) fv = gl2_add(gl2_mul(fv, base2(subgroup_x)), rd2(proof + L.final_poly + 2 * j));
    if (!gl2_eq(fv, old_eval)) return 7;
  }
  return 0;
}
}  // namespace

extern "C" int lcp2_verify(const lcp2_circuit *c, const uint64_t *proof, size_t proof_words, const uint64_t *public_inputs,
                           size_t num_public_inputs, int *failed_check) {
  if (!c || !proof) return LCP2_E_INVALID;
  VerifierView v = verifier_view(c);
  if (v.npi && !public_inputs) return LCP2_E_INVALID;
  // an untrusted proof is only ever read through the layout of THIS circuit: refuse any other length up front
  if (proof_words != ProofLayout(*v.p).total || num_public_inputs != v.npi) return LCP2_E_INVALID;
  int rc = verify_impl(v, (const u64 *)proof, (const u64 *)public_inputs);
  if (failed_check) *failed_check = rc;
  return rc == 0 ? LCP2_OK : LCP2_E_VERIFY;
}

// ------------------------------------------------------------------ proof <-> bytes (plonky2 util/serialization.rs, [RECALL])
namespace {
// walks the flat proof in field order; `word` sees every field element slot, `count` every MerkleProof sibling-count byte
template <class Word, class Count>
bool walk_proof(const lcp2_params &p, Word &&word, Count &&count) {
  const ProofLayout L(p);
  const size_t CH = p.num_challenges, NR = p.num_routed_wires, NC = p.num_constants, W = p.num_wires, Q = p.quotient_degree_factor;
  const size_t npp = (NR + Q - 1) / Q - 1;
  auto run = [&](size_t off, size_t n) { for (size_t i = 0; i < n; i++) if (!word(off + i)) return false; return true; };
  if (!run(L.wires_cap, 3 * L.capw)) return false;
  // OpeningSet: constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys (the flat layout's order)
  if (!run(L.op_constants, 2 * (NC + NR + W)) || !run(L.op_zs, 2 * CH) || !run(L.op_zs_next, 2 * CH) || !run(L.op_pp, 2 * CH * npp) || !run(L.op_quot, 2 * CH * Q)) return false;
  if (!run(L.fri_caps, p.num_fri_layers * L.capw)) return false;
  for (u32 q = 0; q < p.num_query_rounds; q++) {
    const size_t R = L.queries + (size_t)q * L.query_words;
    for (int o = 0; o < 4; o++) {
      if (!run(R + L.q_init_off[o], L.q_init_cols[o])) return false;
      if (!count(L.q_init_sib)) return false;
      if (!run(R + L.q_init_off[o] + L.q_init_cols[o], 4 * L.q_init_sib)) return false;
    }
    for (u32 l = 0; l < p.num_fri_layers; l++) {
      const size_t ev = (size_t)2 << p.fri_arity_bits[l];
      if (!run(R + L.q_step_off[l], ev)) return false;
      if (!count(L.q_step_sib[l])) return false;
      if (!run(R + L.q_step_off[l] + ev, 4 * L.q_step_sib[l])) return false;
    }
  }
  return run(L.final_poly, 2 * L.final_len) && run(L.pow_witness, 1);
}
// the full shape check of build(): ProofLayout subtracts the FRI arities from the LDE height in unsigned arithmetic, so a schedule that
// does not fit (a single arity of 31, a sum above degree_bits) must be refused before any layout is computed
bool params_ok(const lcp2_params *p) {
  bool unsupported;
  return p && !params_problem(*p, &unsupported) && p->degree_bits >= 1 && p->degree_bits + p->rate_bits <= 30 && p->cap_height <= p->degree_bits + p->rate_bits &&
         p->num_fri_layers <= LCP2_MAX_FRI_LAYERS && p->num_query_rounds <= 64 && p->quotient_degree_factor >= 1 && p->num_routed_wires >= 1 &&
         p->num_routed_wires <= p->num_wires && p->num_wires <= 65535 && p->num_constants <= 65535 && p->num_challenges >= 1 && p->num_challenges <= 4;
}
inline void put64(uint8_t *o, u64 v) { for (int i = 0; i < 8; i++) o[i] = (uint8_t)(v >> (8 * i)); }
inline u64 get64(const uint8_t *o) { u64 v = 0; for (int i = 0; i < 8; i++) v |= (u64)o[i] << (8 * i); return v; }
}  // namespace

extern "C" int lcp2_proof_layout_of(const lcp2_params *p, lcp2_proof_layout *o) {
  if (!params_ok(p) || !o) return LCP2_E_INVALID;
  const ProofLayout L(*p);
  memset(o, 0, sizeof *o);
  o->cap_words = L.capw; o->wires_cap = L.wires_cap; o->zs_cap = L.zs_cap; o->quot_cap = L.quot_cap;
  o->op_constants = L.op_constants; o->op_sigmas = L.op_sigmas; o->op_wires = L.op_wires; o->op_zs = L.op_zs; o->op_zs_next = L.op_zs_next;
  o->op_partial_products = L.op_pp; o->op_quotient = L.op_quot;
  o->fri_caps = L.fri_caps; o->queries = L.queries; o->query_words = L.query_words; o->q_init_sib = L.q_init_sib;
  for (int i = 0; i < 4; i++) { o->q_init_off[i] = L.q_init_off[i]; o->q_init_cols[i] = L.q_init_cols[i]; }
  for (int l = 0; l < LCP2_MAX_FRI_LAYERS; l++) { o->q_step_off[l] = L.q_step_off[l]; o->q_step_sib[l] = L.q_step_sib[l]; }
  o->final_poly = L.final_poly; o->final_len = L.final_len; o->pow_witness = L.pow_witness; o->total = L.total;
  return LCP2_OK;
}

extern "C" size_t lcp2_proof_bytes(const lcp2_params *p, size_t npi, uint32_t flags) {
  if (!params_ok(p)) return 0;
  size_t n = 0;
  walk_proof(*p, [&](size_t) { n += 8; return true; }, [&](size_t) { n += 1; return true; });
  return n + 8 * npi + ((flags & LCP2_SER_PUBLIC_INPUT_COUNT) ? 8 : 0);
}
extern "C" int lcp2_proof_to_bytes(const lcp2_params *p, const uint64_t *proof, size_t proof_words, const uint64_t *pis, size_t npi, uint32_t flags,
                                   uint8_t *out, size_t out_len) {
  if (!params_ok(p) || !proof || !out || (npi && !pis)) return LCP2_E_INVALID;
  if (proof_words != ProofLayout(*p).total || out_len != lcp2_proof_bytes(p, npi, flags)) return LCP2_E_INVALID;
  size_t pos = 0;
  bool ok = walk_proof(*p, [&](size_t w) { if (proof[w] >= GL_P) return false; put64(out + pos, proof[w]); pos += 8; return true; },
                       [&](size_t nsib) { if (nsib > 255) return false; out[pos++] = (uint8_t)nsib; return true; });
  if (!ok) return LCP2_E_INVALID;
  if (flags & LCP2_SER_PUBLIC_INPUT_COUNT) { put64(out + pos, npi); pos += 8; }
  for (size_t i = 0; i < npi; i++) { put64(out + pos, gl_canon(pis[i])); pos += 8; }
  return pos == out_len ? LCP2_OK : LCP2_E_INVALID;
}
extern "C" int lcp2_proof_from_bytes(const lcp2_params *p, const uint8_t *bytes, size_t len, uint32_t flags, uint64_t *proof, size_t proof_words,
                                     uint64_t *pis, size_t npi) {
  if (!params_ok(p) || !bytes || !proof || (npi && !pis)) return LCP2_E_INVALID;
  if (proof_words != ProofLayout(*p).total || len != lcp2_proof_bytes(p, npi, flags)) return LCP2_E_INVALID;  // nothing is read past `len`
  size_t pos = 0;
  bool ok = walk_proof(*p, [&](size_t w) { const u64 v = get64(bytes + pos); pos += 8; if (v >= GL_P) return false; proof[w] = v; return true; },
                       [&](size_t nsib) { return bytes[pos++] == (uint8_t)nsib && nsib <= 255; });
  if (!ok) return LCP2_E_INVALID;
  if (flags & LCP2_SER_PUBLIC_INPUT_COUNT) { if (get64(bytes + pos) != npi) return LCP2_E_INVALID; pos += 8; }
  for (size_t i = 0; i < npi; i++) { const u64 v = get64(bytes + pos); pos += 8; if (v >= GL_P) return LCP2_E_INVALID; pis[i] = v; }
  return LCP2_OK;
}
extern "C" int lcp2_verifier_data_to_bytes(const lcp2_circuit *c, uint8_t *out, size_t out_len) {
  if (!c || !out) return LCP2_E_INVALID;
  const VerifierView v = verifier_view(c);
  const size_t capw = (size_t)4 << v.p->cap_height;
  if (out_len != (capw + 4) * 8) return LCP2_E_INVALID;
  for (size_t i = 0; i < capw; i++) put64(out + 8 * i, v.cs_cap[i]);
  for (size_t i = 0; i < 4; i++) put64(out + 8 * (capw + i), v.digest[i]);
  return LCP2_OK;
}
