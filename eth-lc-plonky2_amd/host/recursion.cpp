// Recursive verifier gadget (recursion.hpp).  Every function restates, over targets, the step of csrc/verifier.hip::verify_impl
// named beside it; the two must stay in lockstep (tests/cpp/test_gadgets.cpp proves an inner circuit, verifies the proof natively
// and in-circuit, and checks that a tampered proof makes the outer witness generation fail).
#include "recursion.hpp"
#include "host_internal.hpp"

namespace lc {

CommonCircuitData CommonCircuitData::of(const CircuitDescription &d) {
  CommonCircuitData c;
  c.params = d.params; c.k_is = d.k_is; c.num_selectors = d.num_selectors; c.num_public_inputs = d.num_public_inputs;
  c.gates = d.gates; c.code = d.code; c.imm = d.imm;
  return c;
}

namespace {
// ---- arithmetic over "a constant known at build time, or a target", so that constants fold instead of costing gates
struct SV { bool k; F v; Target t; };
inline SV K(F v) { return SV{true, v % GOLDILOCKS_P, Target{}}; }
inline SV T(Target t) { return SV{false, 0, t}; }
struct EV { SV a, b; };  // a + b X
inline F f_neg(F a) { return a ? GOLDILOCKS_P - a : 0; }
inline F f_inv(F x) { return f_pow(x, GOLDILOCKS_P - 2); }
inline uint32_t bitrev(uint32_t x, uint32_t bits) { uint32_t r = 0; for (uint32_t i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i); return r; }

struct Ar {
  CircuitBuilder &B;
  Target one, zero;
  explicit Ar(CircuitBuilder &b) : B(b), one(b.one()), zero(b.zero()) {}
  // gate constants (const_0, const_1 of an ArithmeticGate row) are shared by the 20 operations of a row: only coefficients
  // from a small set go there, anything else becomes a constant target
  static bool small(F c) { return c < 65536 || c > GOLDILOCKS_P - 65536; }
  Target tg(SV x) { return x.k ? B.constant(x.v) : x.t; }

  // c0 * x * y + c1 * z with every constant folded
  SV lin(F c0, SV x, SV y, F c1, SV z) {
    F pc = c0 % GOLDILOCKS_P;
    Target px = one, py = one;
    int nt = 0;
    if (x.k) pc = f_mul(pc, x.v); else { px = x.t; nt++; }
    if (y.k) pc = f_mul(pc, y.v); else { (nt ? py : px) = y.t; nt++; }
    const bool pconst = nt == 0 || pc == 0;
    F ac = c1 % GOLDILOCKS_P;
    bool aconst = true;
    Target az = zero;
    if (z.k) ac = f_mul(ac, z.v);
    else if (ac != 0) { aconst = false; az = z.t; }
    if (pconst && aconst) return K(f_add(nt == 0 ? pc : 0, ac));
    struct { F c; Target x, y; } pt;  // product term  c * x * y
    struct { F c; Target z; } at;     // added term    c * z
    auto const_term = [&](F c) -> decltype(at) {
      if (c == 0) return {0, zero};
      if (small(c)) return {c, one};
      return {1, B.constant(c)};
    };
    if (pconst) { pt = {ac, az, one}; at = const_term(nt == 0 ? pc : 0); }
    else { pt = {pc, px, py}; if (aconst) at = const_term(ac); else at = {ac, az}; }
    if (at.c == 0 && pt.c == 1 && pt.y.id == one.id) return T(pt.x);
    if (!small(pt.c)) {
      if (pt.y.id != one.id) pt.x = B.arithmetic(1, pt.x, pt.y, 0, zero);
      pt.y = B.constant(pt.c);
      pt.c = 1;
    }
    if (!small(at.c)) { at.z = B.arithmetic(1, at.z, B.constant(at.c), 0, zero); at.c = 1; }
    return T(B.arithmetic(pt.c, pt.x, pt.y, at.c, at.z));
  }
  SV add(SV x, SV y) { return lin(1, x, K(1), 1, y); }
  SV sub(SV x, SV y) { return lin(1, x, K(1), GOLDILOCKS_P - 1, y); }
  SV mul(SV x, SV y) { return lin(1, x, y, 0, K(0)); }
  SV mul_add(SV x, SV y, SV z) { return lin(1, x, y, 1, z); }
  SV select(BoolTarget b, SV x, SV y) { return mul_add(T(b.target), sub(x, y), y); }  // b ? x : y
  void assert_eq(SV x, SV y) { B.connect(tg(x), tg(y)); }

  EV ext(Target c0, Target c1) { return EV{T(c0), T(c1)}; }
  EV base(SV x) { return EV{x, K(0)}; }
  EV eadd(EV x, EV y) { return EV{add(x.a, y.a), add(x.b, y.b)}; }
  EV esub(EV x, EV y) { return EV{sub(x.a, y.a), sub(x.b, y.b)}; }
  EV emul(EV x, EV y) {  // (a + bX)(c + dX) = ac + 7bd + (ad + bc) X
    return EV{lin(1, x.a, y.a, 1, lin(7, x.b, y.b, 0, K(0))), lin(1, x.a, y.b, 1, lin(1, x.b, y.a, 0, K(0)))};
  }
  EV emul_add(EV x, EV y, EV z) { return EV{lin(7, x.b, y.b, 1, lin(1, x.a, y.a, 1, z.a)), lin(1, x.b, y.a, 1, lin(1, x.a, y.b, 1, z.b))}; }
  EV escale(EV x, SV s) { return EV{mul(x.a, s), mul(x.b, s)}; }
  EV escale_add(EV x, SV s, EV z) { return EV{mul_add(x.a, s, z.a), mul_add(x.b, s, z.b)}; }  // x * s + z, s in the base field
  EV eselect(BoolTarget b, EV x, EV y) { return EV{select(b, x.a, y.a), select(b, x.b, y.b)}; }
  void eassert_eq(EV x, EV y) { assert_eq(x.a, y.a); assert_eq(x.b, y.b); }
  EV einv(EV x) {  // the generator supplies the inverse, x * inv = 1 pins it (and x = 0 cannot be proved)
    std::array<Target, 2> h = B.hint_ext_inverse(tg(x.a), tg(x.b));
    EV inv = ext(h[0], h[1]);
    eassert_eq(emul(x, inv), EV{K(1), K(0)});
    return inv;
  }
  EV epow2k(EV x, uint32_t k) { for (uint32_t i = 0; i < k; i++) x = emul(x, x); return x; }

  // entries[index], index given by little-endian bits: a tree of selections
  template <class V, class Sel>
  V random_access(const std::vector<BoolTarget> &bits, size_t first, size_t nbits, std::vector<V> entries, Sel sel) {
    if (entries.size() != ((size_t)1 << nbits)) throw std::runtime_error("random_access: entry count");
    for (size_t l = 0; l < nbits; l++) {
      std::vector<V> next(entries.size() / 2);
      for (size_t i = 0; i < next.size(); i++) next[i] = sel(bits[first + l], entries[2 * i + 1], entries[2 * i]);
      entries.swap(next);
    }
    return entries[0];
  }
};

}  // namespace

// ------------------------------------------------------------------ hashing gadgets
std::array<Target, 4> hash_n_to_hash_no_pad(CircuitBuilder &B, const std::vector<Target> &inputs) {
  std::array<Target, 12> s;
  s.fill(B.zero());
  for (size_t off = 0; off < inputs.size(); off += 8) {  // overwrite-mode sponge, rate 8 (HostPoseidon::hash_no_pad)
    for (size_t i = 0; i < 8 && off + i < inputs.size(); i++) s[i] = inputs[off + i];
    s = B.poseidon(s);
  }
  return {s[0], s[1], s[2], s[3]};
}

// HostPoseidon::merkle_verify: leaf_index_bits are the little-endian bits of the leaf index; the first siblings.size() / 4 of them
// choose the side at each level (the PoseidonGate's swap input), the rest select the cap entry
void verify_merkle_proof_to_cap(CircuitBuilder &B, const std::vector<Target> &leaf, const std::vector<BoolTarget> &bits,
                                const std::vector<Target> &siblings, const std::vector<Target> &cap) {
  const size_t nsib = siblings.size() / 4;
  size_t cap_bits = 0;
  while (((size_t)4 << cap_bits) < cap.size()) cap_bits++;
  if (siblings.size() % 4 || ((size_t)4 << cap_bits) != cap.size() || bits.size() != nsib + cap_bits) throw std::runtime_error("verify_merkle_proof_to_cap: shape");
  std::array<Target, 4> cur;
  if (leaf.size() <= 4) {  // hash_or_noop
    for (size_t i = 0; i < 4; i++) cur[i] = i < leaf.size() ? leaf[i] : B.zero();
  } else {
    cur = hash_n_to_hash_no_pad(B, leaf);
  }
  for (size_t k = 0; k < nsib; k++) {
    std::array<Target, 12> s;
    s.fill(B.zero());
    for (size_t i = 0; i < 4; i++) { s[i] = cur[i]; s[4 + i] = siblings[4 * k + i]; }
    std::array<Target, 12> out = B.poseidon(s, bits[k]);  // index bit set: two_to_one(sibling, cur)
    for (size_t i = 0; i < 4; i++) cur[i] = out[i];
  }
  Ar A(B);
  for (size_t i = 0; i < 4; i++) {
    std::vector<SV> column(cap.size() / 4);
    for (size_t e = 0; e < column.size(); e++) column[e] = T(cap[4 * e + i]);
    SV want = A.random_access<SV>(bits, nsib, cap_bits, column, [&](BoolTarget b, SV x, SV y) { return A.select(b, x, y); });
    A.assert_eq(T(cur[i]), want);
  }
}

// ------------------------------------------------------------------ Challenger (csrc/host_protocol.hpp HostChallenger)
RecursiveChallenger::RecursiveChallenger(CircuitBuilder &b) : b_(b) { sponge_.fill(b.zero()); }
void RecursiveChallenger::observe_element(Target t) {
  output_.clear();
  input_.push_back(t);
  if (input_.size() == 8) duplex();
}
Target RecursiveChallenger::get_challenge() {
  if (!input_.empty() || output_.empty()) duplex();
  Target t = output_.back();
  output_.pop_back();
  return t;
}
void RecursiveChallenger::duplex() {
  for (size_t i = 0; i < input_.size(); i++) sponge_[i] = input_[i];
  input_.clear();
  sponge_ = b_.poseidon(sponge_);
  output_.assign(sponge_.begin(), sponge_.begin() + 8);
}

// ------------------------------------------------------------------ targets
ProofWithPublicInputsTarget add_virtual_proof_with_pis(CircuitBuilder &B, const CommonCircuitData &c) {
  ProofWithPublicInputsTarget t;
  t.proof.resize(lcp2_proof_words(&c.params));
  if (t.proof.empty()) throw std::runtime_error("add_virtual_proof_with_pis: bad parameters");
  for (Target &w : t.proof) w = B.add_virtual_target();
  t.public_inputs.resize(c.num_public_inputs);
  for (Target &w : t.public_inputs) w = B.add_virtual_target();
  return t;
}
VerifierCircuitTarget add_virtual_verifier_data(CircuitBuilder &B, uint32_t cap_height) {
  VerifierCircuitTarget v;
  v.constants_sigmas_cap.resize((size_t)4 << cap_height);
  for (Target &w : v.constants_sigmas_cap) w = B.add_virtual_target();
  for (Target &w : v.circuit_digest) w = B.add_virtual_target();
  return v;
}
VerifierCircuitTarget constant_verifier_data(CircuitBuilder &B, const uint64_t digest[4], const std::vector<uint64_t> &cap) {
  VerifierCircuitTarget v;
  for (uint64_t w : cap) v.constants_sigmas_cap.push_back(B.constant(w));
  for (int i = 0; i < 4; i++) v.circuit_digest[i] = B.constant(digest[i]);
  return v;
}
void set_proof_with_pis_target(PartialWitness &pw, const ProofWithPublicInputsTarget &t, const ProofWithPublicInputs &p) {
  if (p.proof.size() != t.proof.size() || p.public_inputs.size() != t.public_inputs.size()) throw std::runtime_error("set_proof_with_pis_target: shape mismatch");
  for (size_t i = 0; i < t.proof.size(); i++) {
    if (p.proof[i] >= GOLDILOCKS_P) throw UnsatisfiedError("proof element is not a canonical field element");  // verify_impl check 1
    pw.set_target(t.proof[i], p.proof[i]);
  }
  for (size_t i = 0; i < t.public_inputs.size(); i++) pw.set_target(t.public_inputs[i], p.public_inputs[i]);
}
void set_verifier_data_target(PartialWitness &pw, const VerifierCircuitTarget &t, const uint64_t digest[4], const std::vector<uint64_t> &cap) {
  if (cap.size() != t.constants_sigmas_cap.size()) throw std::runtime_error("set_verifier_data_target: cap size");
  for (size_t i = 0; i < cap.size(); i++) pw.set_target(t.constants_sigmas_cap[i], cap[i]);
  for (int i = 0; i < 4; i++) pw.set_target(t.circuit_digest[i], digest[i]);
}

// ------------------------------------------------------------------ verify_proof
namespace {
// csrc/verifier.hip eval_gates_ext: the inner circuit's gate programs over extension targets at zeta
void eval_gates_ext(Ar &A, const CommonCircuitData &c, const std::vector<EV> &wires, const std::vector<EV> &consts, const SV pis[4], const std::vector<SV> &alphas,
                    std::vector<EV> &out) {
  static const uint64_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  const uint32_t CH = c.params.num_challenges;
  const EV ezero{K(0), K(0)};
  out.assign(CH, ezero);
  std::vector<EV> regs(64, ezero);
  for (const lcp2_gate &G : c.gates) {
    const bool fwd = (G.flags & LCP2_GATE_EMIT_FORWARD) != 0;
    std::vector<EV> acc(CH, ezero);
    std::vector<SV> apow(CH, K(1));
    for (uint32_t pc = G.code_offset; pc < G.code_offset + G.code_len; pc++) {
      const uint32_t w0 = c.code[2 * pc], w1 = c.code[2 * pc + 1];
      const uint32_t op = w0 & 0xF, dst = (w0 >> 8) & 0xFF, ka = (w0 >> 16) & 0xF, kb = (w0 >> 20) & 0xF, ia = w1 & 0xFFFF, ib = w1 >> 16;
      auto fetch = [&](uint32_t k, uint32_t i) -> EV {
        switch (k) {
          case 0: return regs[i];
          case 1: return wires[i];
          case 2: return consts[c.num_selectors + i];
          case 3: return EV{K(c.imm[i]), K(0)};
          default: return EV{pis[i], K(0)};
        }
      };
      if (op == LCP2_OP_PMDS) {
        EV in[12];
        for (int i = 0; i < 12; i++) in[i] = regs[ia + i];
        for (int r = 0; r < 12; r++) {
          EV t{K(c.imm[ib + r]), K(0)};
          if (r == 0) t = A.escale_add(in[0], K(8), t);
          for (int i = 0; i < 12; i++) t = A.escale_add(in[(i + r) % 12], K(CIRC[i]), t);
          regs[dst + r] = t;
        }
        continue;
      }
      EV a = fetch(ka, ia);
      if (op == LCP2_OP_EMIT || op == LCP2_OP_EMITBOOL) {
        if (op == LCP2_OP_EMITBOOL) a = A.esub(A.emul(a, a), a);
        for (uint32_t k = 0; k < CH; k++) {
          if (fwd) { acc[k] = A.escale_add(a, apow[k], acc[k]); apow[k] = A.mul(apow[k], alphas[k]); }
          else acc[k] = A.escale_add(acc[k], alphas[k], a);
        }
        continue;
      }
      if (op == LCP2_OP_SBOX) { EV x2 = A.emul(a, a), x4 = A.emul(x2, x2), x3 = A.emul(x2, a); regs[dst] = A.emul(x3, x4); continue; }
      EV b = fetch(kb, ib);
      switch (op) {
        case LCP2_OP_ADD: regs[dst] = A.eadd(a, b); break;
        case LCP2_OP_SUB: regs[dst] = A.esub(a, b); break;
        case LCP2_OP_MUL: regs[dst] = A.emul(a, b); break;
        case LCP2_OP_XOR: { EV ab = A.emul(a, b); regs[dst] = A.esub(A.esub(A.eadd(a, b), ab), ab); break; }
        case LCP2_OP_DBLADD: regs[dst] = A.eadd(A.eadd(a, a), b); break;
        default: regs[dst] = A.emul_add(a, b, regs[dst]); break;  // LCP2_OP_MULADD
      }
    }
    EV s = consts[G.selector_index], f{K(1), K(0)};
    for (uint32_t j = G.group_start; j < G.group_end; j++)
      if (j != G.selector_value) f = A.emul(f, A.esub(EV{K(j), K(0)}, s));
    if (c.num_selectors > 1) f = A.emul(f, A.esub(EV{K(0xFFFFFFFFull), K(0)}, s));
    for (uint32_t k = 0; k < CH; k++) out[k] = A.emul_add(f, acc[k], out[k]);
  }
}

// fri/verifier.rs compute_evaluation: the value at beta of the interpolant through the coset of x.  The points are
// p_i = s g^i (s = the coset's first point), so with Z(X) = X^arity - s^arity the barycentric form is
//   P(beta) = Z(beta) / (arity s^arity) * sum_i ev_i p_i / (beta - p_i)
// (csrc/verifier.hip::fri_compute_evaluation evaluates the same interpolant by the plain Lagrange formula)
EV fri_compute_evaluation(Ar &A, SV x, const std::vector<BoolTarget> &xi_bits, uint32_t ab, const std::vector<EV> &evals_bitrev, EV beta, EV beta_pow_arity) {
  const uint32_t arity = 1u << ab;
  const F g = f_root_of_unity(ab);
  // coset_start = x * g^(-bitrev(within)): bit j of `within` carries weight 2^(ab - 1 - j) in the reversed index
  SV s = x;
  for (uint32_t j = 0; j < ab; j++) {
    const F w = f_inv(f_pow(g, 1ull << (ab - 1 - j)));
    s = A.mul(s, A.lin(1, T(xi_bits[j].target), K(f_add(w, GOLDILOCKS_P - 1)), 1, K(1)));  // bit ? w : 1
  }
  SV s_pow = s;
  for (uint32_t i = 0; i < ab; i++) s_pow = A.mul(s_pow, s_pow);
  EV sum{K(0), K(0)};
  for (uint32_t i = 0; i < arity; i++) {
    SV p = A.mul(s, K(f_pow(g, i)));
    EV inv = A.einv(EV{A.sub(beta.a, p), beta.b});
    sum = A.emul_add(A.escale(evals_bitrev[bitrev(i, ab)], p), inv, sum);
  }
  EV z = EV{A.sub(beta_pow_arity.a, s_pow), beta_pow_arity.b};
  Target norm = A.B.inverse(A.tg(A.mul(s_pow, K(arity))));
  return A.escale(A.emul(z, sum), T(norm));
}
}  // namespace

void verify_proof(CircuitBuilder &B, const ProofWithPublicInputsTarget &pt, const VerifierCircuitTarget &vd, const CommonCircuitData &c) {
  const lcp2_params &p = c.params;
  lcp2_proof_layout L;
  if (lcp2_proof_layout_of(&p, &L) != LCP2_OK || pt.proof.size() != L.total || pt.public_inputs.size() != c.num_public_inputs ||
      vd.constants_sigmas_cap.size() != L.cap_words)
    throw std::runtime_error("verify_proof: the targets do not have the shape of this circuit's proofs");
  const uint64_t n = 1ull << p.degree_bits;
  const uint32_t W = p.num_wires, NR = p.num_routed_wires, NC = p.num_constants, CH = p.num_challenges, Q = p.quotient_degree_factor;
  const uint32_t nchunks = (NR + Q - 1) / Q, npp = nchunks - 1, lgN = p.degree_bits + p.rate_bits;
  Ar A(B);
  const std::vector<Target> &pr = pt.proof;
  auto rd2 = [&](size_t off) { return A.ext(pr[off], pr[off + 1]); };
  auto span = [&](size_t off, size_t cnt) { return std::vector<Target>(pr.begin() + off, pr.begin() + off + cnt); };

  // ---- get_challenges
  const std::array<Target, 4> pi_hash = hash_n_to_hash_no_pad(B, pt.public_inputs);
  RecursiveChallenger ch(B);
  ch.observe_elements({vd.circuit_digest.begin(), vd.circuit_digest.end()});
  ch.observe_elements({pi_hash.begin(), pi_hash.end()});
  ch.observe_elements(span(L.wires_cap, L.cap_words));
  std::vector<SV> betas, gammas, alphas;
  for (uint32_t k = 0; k < CH; k++) betas.push_back(T(ch.get_challenge()));
  for (uint32_t k = 0; k < CH; k++) gammas.push_back(T(ch.get_challenge()));
  ch.observe_elements(span(L.zs_cap, L.cap_words));
  for (uint32_t k = 0; k < CH; k++) alphas.push_back(T(ch.get_challenge()));
  ch.observe_elements(span(L.quot_cap, L.cap_words));
  const ExtensionTarget zeta_t = ch.get_extension_challenge();
  const EV zeta = A.ext(zeta_t.c0, zeta_t.c1);
  ch.observe_elements(span(L.op_constants, 2 * (NC + NR + W)));
  ch.observe_elements(span(L.op_zs, 2 * CH));
  ch.observe_elements(span(L.op_partial_products, 2 * CH * npp));
  ch.observe_elements(span(L.op_quotient, 2 * CH * Q));
  ch.observe_elements(span(L.op_zs_next, 2 * CH));
  const ExtensionTarget fa = ch.get_extension_challenge();
  const EV fri_alpha = A.ext(fa.c0, fa.c1);
  std::vector<EV> fri_betas;
  for (uint32_t l = 0; l < p.num_fri_layers; l++) {
    ch.observe_elements(span(L.fri_caps + l * L.cap_words, L.cap_words));
    const ExtensionTarget b = ch.get_extension_challenge();
    fri_betas.push_back(A.ext(b.c0, b.c1));
  }
  ch.observe_elements(span(L.final_poly, 2 * L.final_len));
  ch.observe_element(pr[L.pow_witness]);
  // fri_proof_of_work: the leading proof_of_work_bits of the response are zero
  if (p.proof_of_work_bits < 1 || p.proof_of_work_bits > 31) throw std::runtime_error("verify_proof: proof_of_work_bits out of range");
  B.split_canonical(ch.get_challenge(), 32 - p.proof_of_work_bits);

  // ---- vanishing(zeta) = Z_H(zeta) * t(zeta)
  std::vector<EV> ow(W), oc(NC + NR);
  for (uint32_t j = 0; j < W; j++) ow[j] = rd2(L.op_wires + 2 * j);
  for (uint32_t j = 0; j < NC + NR; j++) oc[j] = rd2(L.op_constants + 2 * j);
  const EV one{K(1), K(0)};
  const EV zeta_n = A.epow2k(zeta, p.degree_bits);
  const EV zh = A.esub(zeta_n, one);
  {
    const EV l0 = A.emul(zh, A.einv(A.escale(A.esub(zeta, one), K(n % GOLDILOCKS_P))));
    std::vector<EV> terms;
    for (uint32_t k = 0; k < CH; k++) terms.push_back(A.emul(l0, A.esub(rd2(L.op_zs + 2 * k), one)));
    for (uint32_t k = 0; k < CH; k++) {
      EV prev = rd2(L.op_zs + 2 * k);
      for (uint32_t cch = 0; cch < nchunks; cch++) {
        EV pn = one, pd = one;
        for (uint32_t j = cch * Q; j < NR && j < (cch + 1) * Q; j++) {
          EV num = A.escale_add(A.escale(zeta, K(c.k_is[j])), betas[k], ow[j]);
          num.a = A.add(num.a, gammas[k]);
          EV den = A.escale_add(oc[NC + j], betas[k], ow[j]);
          den.a = A.add(den.a, gammas[k]);
          pn = A.emul(pn, num);
          pd = A.emul(pd, den);
        }
        EV next = cch < npp ? rd2(L.op_partial_products + 2 * (k * npp + cch)) : rd2(L.op_zs_next + 2 * k);
        terms.push_back(A.esub(A.emul(prev, pn), A.emul(next, pd)));
        prev = next;
      }
    }
    const SV pis[4] = {T(pi_hash[0]), T(pi_hash[1]), T(pi_hash[2]), T(pi_hash[3])};
    std::vector<EV> gates;
    eval_gates_ext(A, c, ow, oc, pis, alphas, gates);
    for (uint32_t k = 0; k < CH; k++) {
      EV acc = gates[k];
      for (size_t t = terms.size(); t-- > 0;) acc = A.escale_add(acc, alphas[k], terms[t]);
      EV tq{K(0), K(0)};
      for (uint32_t j = Q; j-- > 0;) tq = A.emul_add(tq, zeta_n, rd2(L.op_quotient + 2 * (k * Q + j)));
      A.eassert_eq(acc, A.emul(zh, tq));
    }
  }

  // ---- FRI
  EV red0{K(0), K(0)}, red1{K(0), K(0)};
  {
    std::vector<EV> vals;
    for (uint32_t j = 0; j < NC + NR + W; j++) vals.push_back(rd2(L.op_constants + 2 * j));
    for (uint32_t j = 0; j < CH; j++) vals.push_back(rd2(L.op_zs + 2 * j));
    for (uint32_t j = 0; j < CH * npp; j++) vals.push_back(rd2(L.op_partial_products + 2 * j));
    for (uint32_t j = 0; j < CH * Q; j++) vals.push_back(rd2(L.op_quotient + 2 * j));
    for (size_t j = vals.size(); j-- > 0;) red0 = A.emul_add(red0, fri_alpha, vals[j]);
    for (uint32_t j = CH; j-- > 0;) red1 = A.emul_add(red1, fri_alpha, rd2(L.op_zs_next + 2 * j));
  }
  const EV g_zeta = A.escale(zeta, K(f_root_of_unity(p.degree_bits)));
  EV alpha_ch = one;
  for (uint32_t k = 0; k < CH; k++) alpha_ch = A.emul(alpha_ch, fri_alpha);
  std::vector<EV> beta_pow;  // beta_l^(arity_l), shared by the query rounds
  for (uint32_t l = 0; l < p.num_fri_layers; l++) beta_pow.push_back(A.epow2k(fri_betas[l], p.fri_arity_bits[l]));
  const std::vector<Target> caps[4] = {vd.constants_sigmas_cap, span(L.wires_cap, L.cap_words), span(L.zs_cap, L.cap_words), span(L.quot_cap, L.cap_words)};
  const F wN = f_root_of_unity(lgN);
  for (uint32_t q = 0; q < p.num_query_rounds; q++) {
    std::vector<BoolTarget> xbits = B.split_canonical(ch.get_challenge()).bits;
    xbits.resize(lgN);  // x_index = challenge mod 2^lgN
    const size_t R = L.queries + (size_t)q * L.query_words;
    for (int o = 0; o < 4; o++)
      verify_merkle_proof_to_cap(B, span(R + L.q_init_off[o], L.q_init_cols[o]), xbits, span(R + L.q_init_off[o] + L.q_init_cols[o], 4 * L.q_init_sib), caps[o]);
    // subgroup_x = g * w_N^bitrev(x_index): bit i of the index carries weight 2^(lgN - 1 - i)
    SV x = K(7);
    for (uint32_t i = 0; i < lgN; i++) {
      const F w = f_pow(wN, 1ull << (lgN - 1 - i));
      x = A.mul(x, A.lin(1, T(xbits[i].target), K(f_add(w, GOLDILOCKS_P - 1)), 1, K(1)));
    }
    EV r0{K(0), K(0)};
    for (int o = 3; o >= 0; o--)
      for (size_t j = L.q_init_cols[o]; j-- > 0;) r0 = A.emul_add(r0, fri_alpha, EV{T(pr[R + L.q_init_off[o] + j]), K(0)});
    EV sum = A.emul(A.esub(r0, red0), A.einv(EV{A.sub(x, zeta.a), A.sub(K(0), zeta.b)}));
    EV r1{K(0), K(0)};
    for (uint32_t j = CH; j-- > 0;) r1 = A.emul_add(r1, fri_alpha, EV{T(pr[R + L.q_init_off[2] + j]), K(0)});
    sum = A.emul_add(sum, alpha_ch, A.emul(A.esub(r1, red1), A.einv(EV{A.sub(x, g_zeta.a), A.sub(K(0), g_zeta.b)})));
    EV old_eval = sum;
    size_t bit0 = 0;  // xi = x_index >> bit0
    for (uint32_t l = 0; l < p.num_fri_layers; l++) {
      const uint32_t ab = p.fri_arity_bits[l], arity = 1u << ab;
      std::vector<EV> evals(arity);
      for (uint32_t j = 0; j < arity; j++) evals[j] = rd2(R + L.q_step_off[l] + 2 * j);
      std::vector<BoolTarget> within(xbits.begin() + bit0, xbits.begin() + bit0 + ab), coset(xbits.begin() + bit0 + ab, xbits.end());
      EV mine = A.random_access<EV>(within, 0, ab, evals, [&](BoolTarget b, EV u, EV v) { return A.eselect(b, u, v); });
      A.eassert_eq(mine, old_eval);
      old_eval = fri_compute_evaluation(A, x, within, ab, evals, fri_betas[l], beta_pow[l]);
      verify_merkle_proof_to_cap(B, span(R + L.q_step_off[l], 2 * arity), coset, span(R + L.q_step_off[l] + 2 * arity, 4 * L.q_step_sib[l]),
                                 span(L.fri_caps + l * L.cap_words, L.cap_words));
      for (uint32_t i = 0; i < ab; i++) x = A.mul(x, x);
      bit0 += ab;
    }
    EV fv{K(0), K(0)};
    for (size_t j = L.final_len; j-- > 0;) fv = A.escale_add(fv, x, rd2(L.final_poly + 2 * j));
    A.eassert_eq(fv, old_eval);
  }
}

}  // namespace lc
