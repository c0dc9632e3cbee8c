/* ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of plonky2 0.1.4 `prove()` / `verify()` for one circuit
 * (un-vendored, pin /root/reference/Cargo.lock:2347-2350; the reference reaches it
 * only through build/prove/verify, eth-lc-plonky2/src/main.rs:226-233 and
 * src/unit_tests.rs:29-35).  Follows, function by function:
 *   iop/challenger.rs            Challenger (duplex sponge, overwrite mode)
 *   fri/oracle.rs                PolynomialBatch::from_values/from_coeffs, prove_openings
 *   plonk/prover.rs              wires_permutation_partial_products_and_zs, compute_quotient_polys
 *   plonk/vanishing_poly.rs      eval_vanishing_poly(_base_batch), check_partial_products
 *   plonk/proof.rs               OpeningSet::new / to_fri_openings
 *   fri/prover.rs                fri_committed_trees, fri_proof_of_work, fri_prover_query_rounds
 *   plonk/verifier.rs, fri/verifier.rs   verify_with_challenges, verify_fri_proof
 * PARITY UNPINNED: the reference holds no fixture for any value computed here (no
 * proof bytes, caps or challenges are recorded) and cannot be built (no Rust);
 * the restatement is from the published algorithm [RECALL] and is self-checked by
 * prove -> verify plus tamper tests.  Deliberate deviations, both documented in
 * (1) the proof-of-work witness is the MINIMUM valid one (plonky2 uses a
 * nondeterministic rayon find_any); (2) gates are described by a small constraint
 * bytecode ("gate program") because plonky2's gate objects cannot cross a C ABI; the
 * programs of plonky2's own gates (tests/ and eth-lc-plonky2_amd/circuit.py) restate their eval_unfiltered.
 * Public inputs are bound as in plonky2: PublicInputGate compares four wires with public_inputs_hash.
 */
#include "plonk.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ small helpers */
static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); } return p; }
static void *xcalloc(size_t n, size_t s) { void *p = calloc(n ? n : 1, s); if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); } return p; }

/* ------------------------------------------------------------------ Challenger */
typedef struct {
  uint64_t state[12];
  uint64_t in[8]; int nin;
  uint64_t out[8]; int nout;
} challenger;
static void ch_init(challenger *c) { memset(c, 0, sizeof *c); }
static void ch_duplex(challenger *c) {
  for (int i = 0; i < c->nin; i++) c->state[i] = c->in[i];
  c->nin = 0;
  orc_poseidon_permute(c->state);
  memcpy(c->out, c->state, 8 * sizeof(uint64_t));
  c->nout = 8;
}
static void ch_observe(challenger *c, uint64_t x) {
  c->nout = 0;
  c->in[c->nin++] = gl_canon(x);
  if (c->nin == 8) ch_duplex(c);
}
static void ch_observe_n(challenger *c, const uint64_t *x, size_t n) { for (size_t i = 0; i < n; i++) ch_observe(c, x[i]); }
static uint64_t ch_get(challenger *c) {
  if (c->nin || !c->nout) ch_duplex(c);
  return c->out[--c->nout];
}
static gl2_t ch_get_ext(challenger *c) { uint64_t a = ch_get(c), b = ch_get(c); return gl2_make(a, b); }

/* ------------------------------------------------------------------ PolynomialBatch */
struct orc_batch {
  size_t ncols, n, N;
  unsigned lgn, rate_bits;
  uint64_t *coeffs; /* [ncols][n] */
  uint64_t *leaves; /* [N][ncols], leaf order: leaf i = LDE row bitrev(i) */
  orc_merkle *tree;
};
static void batch_free(orc_batch *b) { if (b) { free(b->coeffs); free(b->leaves); orc_merkle_free(b->tree); free(b); } }

static orc_batch *batch_from_coeffs_owned(uint64_t *coeffs, size_t ncols, unsigned lgn, unsigned rate_bits, unsigned cap_height) {
  orc_batch *b = (orc_batch *)xcalloc(1, sizeof *b);
  b->ncols = ncols; b->lgn = lgn; b->n = (size_t)1 << lgn; b->rate_bits = rate_bits; b->N = b->n << rate_bits;
  b->coeffs = coeffs;
  uint64_t *lde = (uint64_t *)xmalloc(ncols * b->N * sizeof(uint64_t));
  orc_lde_batch(coeffs, ncols, b->n, rate_bits, GL_GENERATOR, lde);
  b->leaves = (uint64_t *)xmalloc(ncols * b->N * sizeof(uint64_t));
  unsigned lgN = lgn + rate_bits;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < b->N; i++) {
    size_t src = gl_bitrev(i, lgN);
    for (size_t c = 0; c < ncols; c++) b->leaves[i * ncols + c] = lde[c * b->N + src];
  }
  free(lde);
  b->tree = orc_merkle_build(b->leaves, b->N, ncols, cap_height);
  return b;
}
static orc_batch *batch_from_values(const uint64_t *values, size_t ncols, unsigned lgn, unsigned rate_bits, unsigned cap_height) {
  size_t n = (size_t)1 << lgn;
  uint64_t *coeffs = (uint64_t *)xmalloc(ncols * n * sizeof(uint64_t));
  memcpy(coeffs, values, ncols * n * sizeof(uint64_t));
  orc_ifft_batch(coeffs, ncols, n);
  return batch_from_coeffs_owned(coeffs, ncols, lgn, rate_bits, cap_height);
}
/* get_lde_values(i, step): row reverse_bits(i * step) of the leaf matrix */
static const uint64_t *batch_lde_row(const orc_batch *b, size_t i_natural) {
  return b->leaves + gl_bitrev(i_natural, b->lgn + b->rate_bits) * b->ncols;
}

/* ------------------------------------------------------------------ circuit */
struct orc_circuit {
  orc_params p;
  size_t n;
  uint32_t num_selectors, num_gates, npi;
  orc_gate *gates;
  uint32_t *code; size_t code_words;
  uint64_t *imm; size_t num_imm;
  uint64_t *k_is;
  uint64_t *cs_values;       /* [NC + NR][n] */
  orc_batch *cs;             /* constants_sigmas commitment (NULL in a verifier-only circuit) */
  uint64_t *cs_cap;          /* its Merkle cap: all the verifier needs of the preprocessed polynomials */
  uint64_t digest[4];
  uint32_t max_gate_constraints;
  orc_challenges last;
};
#define NPP(p) (((p).num_routed_wires + (p).quotient_degree_factor - 1) / (p).quotient_degree_factor - 1)

static orc_circuit *circuit_new(const orc_params *p, const uint64_t *constants_sigmas, const uint64_t *k_is, uint32_t num_selectors,
                                const orc_gate *gates, uint32_t num_gates, const uint32_t *code, size_t code_words,
                                const uint64_t *imm, size_t num_imm, uint32_t npi, int commit) {
  if (p->num_challenges > 4 || p->num_query_rounds > 64 || p->num_fri_layers > ORC_MAX_FRI_LAYERS) return NULL;
  if (p->quotient_degree_factor != (1u << p->rate_bits)) return NULL; /* the LDE doubles as the quotient domain */
  orc_circuit *c = (orc_circuit *)xcalloc(1, sizeof *c);
  c->p = *p; c->n = (size_t)1 << p->degree_bits; c->num_selectors = num_selectors; c->num_gates = num_gates; c->npi = npi;
  size_t ncs = p->num_constants + p->num_routed_wires;
  c->gates = (orc_gate *)xmalloc(num_gates * sizeof(orc_gate)); memcpy(c->gates, gates, num_gates * sizeof(orc_gate));
  c->code = (uint32_t *)xmalloc(code_words * 4); memcpy(c->code, code, code_words * 4); c->code_words = code_words;
  c->imm = (uint64_t *)xmalloc(num_imm * 8); for (size_t i = 0; i < num_imm; i++) c->imm[i] = gl_canon(imm[i]); c->num_imm = num_imm;
  c->k_is = (uint64_t *)xmalloc(p->num_routed_wires * 8); for (uint32_t i = 0; i < p->num_routed_wires; i++) c->k_is[i] = gl_canon(k_is[i]);
  c->cs_values = (uint64_t *)xmalloc(ncs * c->n * 8);
  for (size_t i = 0; i < ncs * c->n; i++) c->cs_values[i] = gl_canon(constants_sigmas[i]);
  for (uint32_t g = 0; g < num_gates; g++) if (gates[g].num_constraints > c->max_gate_constraints) c->max_gate_constraints = gates[g].num_constraints;
  if (!commit) return c; /* witness checking only: no preprocessed commitment, no digest */
  c->cs = batch_from_values(c->cs_values, ncs, p->degree_bits, p->rate_bits, p->cap_height);
  /* circuit_builder.rs::build: circuit_digest = hash_no_pad(constants_sigmas_cap || domain_separator_digest || degree_bits),
   * domain_separator_digest = hash_pad(domain_separator), the separator empty by default: pad10*1 to one rate block
   * [1, 0, 0, 0, 0, 0, 0, 1] */
  size_t capw = (size_t)4 << p->cap_height;
  uint64_t *buf = (uint64_t *)xmalloc((capw + 5) * 8);
  c->cs_cap = (uint64_t *)xmalloc(capw * 8);
  memcpy(c->cs_cap, c->cs->tree->cap, capw * 8);
  memcpy(buf, c->cs_cap, capw * 8);
  const uint64_t empty_padded[8] = {1, 0, 0, 0, 0, 0, 0, 1};
  orc_hash_no_pad(empty_padded, 8, buf + capw);
  buf[capw + 4] = p->degree_bits;
  orc_hash_no_pad(buf, capw + 5, c->digest);
  free(buf);
  return c;
}
orc_circuit *orc_circuit_new(const orc_params *p, const uint64_t *constants_sigmas, const uint64_t *k_is, uint32_t num_selectors,
                             const orc_gate *gates, uint32_t num_gates, const uint32_t *code, size_t code_words,
                             const uint64_t *imm, size_t num_imm, uint32_t npi) {
  return circuit_new(p, constants_sigmas, k_is, num_selectors, gates, num_gates, code, code_words, imm, num_imm, npi, 1);
}
/* circuit for orc_check_witness only (skips the constants/sigmas commitment, which dominates at 2^19 rows) */
orc_circuit *orc_circuit_new_unbuilt(const orc_params *p, const uint64_t *constants_sigmas, const uint64_t *k_is, uint32_t num_selectors,
                                     const orc_gate *gates, uint32_t num_gates, const uint32_t *code, size_t code_words,
                                     const uint64_t *imm, size_t num_imm, uint32_t npi) {
  return circuit_new(p, constants_sigmas, k_is, num_selectors, gates, num_gates, code, code_words, imm, num_imm, npi, 0);
}
/* Verifier-only circuit (= plonky2's VerifierCircuitData: VerifierOnlyCircuitData {constants_sigmas_cap, circuit_digest} +
 * CommonCircuitData {config, gates, selectors, k_is, num_public_inputs}): no preprocessed values, no build().  orc_verify works
 * on it at any size; orc_prove / orc_check_witness refuse it. */
orc_circuit *orc_verifier_new(const orc_params *p, const uint64_t *k_is, uint32_t num_selectors, const orc_gate *gates,
                              uint32_t num_gates, const uint32_t *code, size_t code_words, const uint64_t *imm, size_t num_imm,
                              uint32_t npi, const uint64_t digest[4], const uint64_t *cap) {
  if (p->num_challenges > 4 || p->num_query_rounds > 64 || p->num_fri_layers > ORC_MAX_FRI_LAYERS) return NULL;
  if (p->quotient_degree_factor != (1u << p->rate_bits)) return NULL;
  orc_circuit *c = (orc_circuit *)xcalloc(1, sizeof *c);
  c->p = *p; c->n = (size_t)1 << p->degree_bits; c->num_selectors = num_selectors; c->num_gates = num_gates; c->npi = npi;
  c->gates = (orc_gate *)xmalloc(num_gates * sizeof(orc_gate)); memcpy(c->gates, gates, num_gates * sizeof(orc_gate));
  c->code = (uint32_t *)xmalloc(code_words * 4); memcpy(c->code, code, code_words * 4); c->code_words = code_words;
  c->imm = (uint64_t *)xmalloc(num_imm * 8); for (size_t i = 0; i < num_imm; i++) c->imm[i] = gl_canon(imm[i]); c->num_imm = num_imm;
  c->k_is = (uint64_t *)xmalloc(p->num_routed_wires * 8); for (uint32_t i = 0; i < p->num_routed_wires; i++) c->k_is[i] = gl_canon(k_is[i]);
  for (uint32_t g = 0; g < num_gates; g++) if (gates[g].num_constraints > c->max_gate_constraints) c->max_gate_constraints = gates[g].num_constraints;
  size_t capw = (size_t)4 << p->cap_height;
  c->cs_cap = (uint64_t *)xmalloc(capw * 8);
  memcpy(c->cs_cap, cap, capw * 8);
  memcpy(c->digest, digest, 32);
  return c;
}
void orc_circuit_free(orc_circuit *c) {
  if (!c) return;
  free(c->gates); free(c->code); free(c->imm); free(c->k_is); free(c->cs_values); free(c->cs_cap); batch_free(c->cs); free(c);
}
void orc_circuit_digest(const orc_circuit *c, uint64_t digest[4], uint64_t *cap) {
  memcpy(digest, c->digest, 32);
  if (cap) memcpy(cap, c->cs_cap, ((size_t)4 << c->p.cap_height) * 8);
}
void orc_last_challenges(const orc_circuit *c, orc_challenges *out) { *out = c->last; }

/* ------------------------------------------------------------------ proof layout */
typedef struct {
  size_t wires_cap, zs_cap, quot_cap;
  size_t op_constants, op_sigmas, op_wires, op_zs, op_zs_next, op_pp, op_quot, op_end;
  size_t fri_caps;          /* L caps */
  size_t queries;           /* Q rounds, each `query_words` */
  size_t query_words;
  size_t q_init_off[4], q_init_cols[4], q_init_sib;   /* inside a query round */
  size_t q_step_off[ORC_MAX_FRI_LAYERS], q_step_sib[ORC_MAX_FRI_LAYERS];
  size_t final_poly, final_len;
  size_t pow_witness;
  size_t total;
  size_t capw, lgN;
} layout_t;

static void make_layout(const orc_params *p, layout_t *L) {
  memset(L, 0, sizeof *L);
  size_t capw = (size_t)4 << p->cap_height, o = 0;
  size_t CH = p->num_challenges, NR = p->num_routed_wires, NC = p->num_constants, W = p->num_wires, Q = p->quotient_degree_factor;
  size_t npp = NPP(*p);
  L->capw = capw; L->lgN = p->degree_bits + p->rate_bits;
  L->wires_cap = o; o += capw; L->zs_cap = o; o += capw; L->quot_cap = o; o += capw;
  L->op_constants = o; o += 2 * NC; L->op_sigmas = o; o += 2 * NR; L->op_wires = o; o += 2 * W;
  L->op_zs = o; o += 2 * CH; L->op_zs_next = o; o += 2 * CH; L->op_pp = o; o += 2 * CH * npp; L->op_quot = o; o += 2 * CH * Q;
  L->op_end = o;
  L->fri_caps = o; o += p->num_fri_layers * capw;
  size_t cols[4] = {NC + NR, W, CH * (1 + npp), CH * Q};
  size_t q = 0;
  L->q_init_sib = L->lgN - p->cap_height;
  for (int i = 0; i < 4; i++) { L->q_init_off[i] = q; L->q_init_cols[i] = cols[i]; q += cols[i] + 4 * L->q_init_sib; }
  size_t lg = L->lgN;
  for (uint32_t l = 0; l < p->num_fri_layers; l++) {
    lg -= p->fri_arity_bits[l];
    L->q_step_off[l] = q;
    L->q_step_sib[l] = lg - p->cap_height;
    q += ((size_t)2 << p->fri_arity_bits[l]) + 4 * L->q_step_sib[l];
  }
  L->query_words = q;
  L->queries = o; o += q * p->num_query_rounds;
  size_t fl = p->degree_bits;
  for (uint32_t l = 0; l < p->num_fri_layers; l++) fl -= p->fri_arity_bits[l];
  L->final_len = (size_t)1 << fl;
  L->final_poly = o; o += 2 * L->final_len;
  L->pow_witness = o; o += 1;
  L->total = o;
}
size_t orc_proof_words(const orc_params *p) { layout_t L; make_layout(p, &L); return L.total; }

/* ------------------------------------------------------------------ gate programs
 * Evaluated over the base field by the prover (on the LDE coset) and over the extension field by the
 * verifier (at zeta).  acc[c] <- acc[c] * alpha[c] + v on EMIT: the program lists a gate's constraints from
 * the LAST to the FIRST, so acc[c] = sum_i alpha[c]^i * constraint_i.  */
#define DECODE(code, pc) uint32_t w0 = (code)[2 * (pc)], w1 = (code)[2 * (pc) + 1]; \
  uint32_t op = w0 & 0xF, dst = (w0 >> 8) & 0xFF, ka = (w0 >> 16) & 0xF, kb = (w0 >> 20) & 0xF, ia = w1 & 0xFFFF, ib = w1 >> 16

static inline uint64_t operand_b(const orc_circuit *c, uint32_t k, uint32_t i, const uint64_t *regs, const uint64_t *wires,
                                 const uint64_t *consts, const uint64_t *pis) {
  switch (k) {
    case ORC_K_REG: return regs[i];
    case ORC_K_WIRE: return wires[i];
    case ORC_K_CONST: return consts[c->num_selectors + i];
    case ORC_K_IMM: return c->imm[i];
    default: return pis[i];
  }
}
static uint64_t filter_b(const orc_circuit *c, const orc_gate *g, const uint64_t *consts) {
  uint64_t s = consts[g->selector_index], f = 1;
  for (uint32_t j = g->group_start; j < g->group_end; j++)
    if (j != g->selector_value) f = gl_mul(f, gl_sub(j, s));
  if (c->num_selectors > 1) f = gl_mul(f, gl_sub(ORC_UNUSED_SELECTOR, s));
  return f;
}
static const uint64_t PMDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const uint64_t PMDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static inline uint64_t sbox7_b(uint64_t x) { uint64_t x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x3 = gl_mul(x2, x); return gl_mul(x3, x4); }

/* out[ch] = sum_gates filter * sum_i alpha_ch^i constraint_i ; raw (nullable): unfiltered constraints of gate `raw_gate`.
 * The sum over a gate's constraints is accumulated with running powers of alpha for ORC_GATE_EMIT_FORWARD gates and as a
 * Horner chain for the others (their programs list the constraints last to first): two formulations of the same sum. */
static void eval_gates_base(const orc_circuit *c, const uint64_t *wires, const uint64_t *consts, const uint64_t *pis,
                            const uint64_t *alphas, uint64_t *out, int raw_gate, uint64_t *raw) {
  uint32_t CH = c->p.num_challenges;
  for (uint32_t k = 0; k < CH; k++) out[k] = 0;
  uint64_t regs[ORC_MAX_REGS];
  for (uint32_t g = 0; g < c->num_gates; g++) {
    const orc_gate *G = &c->gates[g];
    const int fwd = (G->flags & ORC_GATE_EMIT_FORWARD) != 0;
    uint64_t acc[4] = {0, 0, 0, 0}, apow[4] = {1, 1, 1, 1};
    uint32_t emitted = 0;
    for (uint32_t pc = G->code_offset; pc < G->code_offset + G->code_len; pc++) {
      DECODE(c->code, pc);
      if (op == ORC_OP_PMDS) {
        uint64_t in[12];
        for (int i = 0; i < 12; i++) in[i] = regs[ia + i];
        for (int r = 0; r < 12; r++) {
          uint64_t v = gl_add(c->imm[ib + r], gl_mul(in[r], PMDS_DIAG[r]));
          for (int i = 0; i < 12; i++) v = gl_add(v, gl_mul(in[(i + r) % 12], PMDS_CIRC[i]));
          regs[dst + r] = v;
        }
        continue;
      }
      uint64_t a = operand_b(c, ka, ia, regs, wires, consts, pis);
      if (op == ORC_OP_EMIT || op == ORC_OP_EMITBOOL) {
        if (op == ORC_OP_EMITBOOL) a = gl_sub(gl_mul(a, a), a);
        for (uint32_t k = 0; k < CH; k++) {
          if (fwd) { acc[k] = gl_add(acc[k], gl_mul(apow[k], a)); apow[k] = gl_mul(apow[k], alphas[k]); }
          else acc[k] = gl_add(gl_mul(acc[k], alphas[k]), a);
        }
        if (raw && (int)g == raw_gate) raw[fwd ? emitted : G->num_constraints - 1 - emitted] = a;
        emitted++;
        continue;
      }
      if (op == ORC_OP_SBOX) { regs[dst] = sbox7_b(a); continue; }
      uint64_t b = operand_b(c, kb, ib, regs, wires, consts, pis);
      switch (op) {
        case ORC_OP_ADD: regs[dst] = gl_add(a, b); break;
        case ORC_OP_SUB: regs[dst] = gl_sub(a, b); break;
        case ORC_OP_MUL: regs[dst] = gl_mul(a, b); break;
        case ORC_OP_XOR: { uint64_t ab = gl_mul(a, b); regs[dst] = gl_sub(gl_sub(gl_add(a, b), ab), ab); break; }
        case ORC_OP_DBLADD: regs[dst] = gl_add(gl_add(a, a), b); break;
        default: regs[dst] = gl_add(regs[dst], gl_mul(a, b)); break; /* ORC_OP_MULADD */
      }
    }
    uint64_t f = filter_b(c, G, consts);
    for (uint32_t k = 0; k < CH; k++) out[k] = gl_add(out[k], gl_mul(f, acc[k]));
  }
}

static inline gl2_t operand_e(const orc_circuit *c, uint32_t k, uint32_t i, const gl2_t *regs, const gl2_t *wires,
                              const gl2_t *consts, const uint64_t *pis) {
  switch (k) {
    case ORC_K_REG: return regs[i];
    case ORC_K_WIRE: return wires[i];
    case ORC_K_CONST: return consts[c->num_selectors + i];
    case ORC_K_IMM: return gl2_from_base(c->imm[i]);
    default: return gl2_from_base(pis[i]);
  }
}
static inline gl2_t sbox7_e(gl2_t x) { gl2_t x2 = gl2_mul(x, x), x4 = gl2_mul(x2, x2), x3 = gl2_mul(x2, x); return gl2_mul(x3, x4); }
static void eval_gates_ext(const orc_circuit *c, const gl2_t *wires, const gl2_t *consts, const uint64_t *pis,
                           const uint64_t *alphas, gl2_t *out) {
  uint32_t CH = c->p.num_challenges;
  for (uint32_t k = 0; k < CH; k++) out[k] = gl2_from_base(0);
  gl2_t regs[ORC_MAX_REGS];
  for (uint32_t g = 0; g < c->num_gates; g++) {
    const orc_gate *G = &c->gates[g];
    const int fwd = (G->flags & ORC_GATE_EMIT_FORWARD) != 0;
    gl2_t acc[4];
    uint64_t apow[4] = {1, 1, 1, 1};
    for (int k = 0; k < 4; k++) acc[k] = gl2_from_base(0);
    for (uint32_t pc = G->code_offset; pc < G->code_offset + G->code_len; pc++) {
      DECODE(c->code, pc);
      if (op == ORC_OP_PMDS) {
        gl2_t in[12];
        for (int i = 0; i < 12; i++) in[i] = regs[ia + i];
        for (int r = 0; r < 12; r++) {
          gl2_t v = gl2_add(gl2_from_base(c->imm[ib + r]), gl2_scale(in[r], PMDS_DIAG[r]));
          for (int i = 0; i < 12; i++) v = gl2_add(v, gl2_scale(in[(i + r) % 12], PMDS_CIRC[i]));
          regs[dst + r] = v;
        }
        continue;
      }
      gl2_t a = operand_e(c, ka, ia, regs, wires, consts, pis);
      if (op == ORC_OP_EMIT || op == ORC_OP_EMITBOOL) {
        if (op == ORC_OP_EMITBOOL) a = gl2_sub(gl2_mul(a, a), a);
        for (uint32_t k = 0; k < CH; k++) {
          if (fwd) { acc[k] = gl2_add(acc[k], gl2_scale(a, apow[k])); apow[k] = gl_mul(apow[k], alphas[k]); }
          else acc[k] = gl2_add(gl2_scale(acc[k], alphas[k]), a);
        }
        continue;
      }
      if (op == ORC_OP_SBOX) { regs[dst] = sbox7_e(a); continue; }
      gl2_t b = operand_e(c, kb, ib, regs, wires, consts, pis);
      switch (op) {
        case ORC_OP_ADD: regs[dst] = gl2_add(a, b); break;
        case ORC_OP_SUB: regs[dst] = gl2_sub(a, b); break;
        case ORC_OP_MUL: regs[dst] = gl2_mul(a, b); break;
        case ORC_OP_XOR: { gl2_t ab = gl2_mul(a, b); regs[dst] = gl2_sub(gl2_sub(gl2_add(a, b), ab), ab); break; }
        case ORC_OP_DBLADD: regs[dst] = gl2_add(gl2_add(a, a), b); break;
        default: regs[dst] = gl2_add(regs[dst], gl2_mul(a, b)); break; /* ORC_OP_MULADD */
      }
    }
    gl2_t s = consts[G->selector_index], f = gl2_from_base(1);
    for (uint32_t j = G->group_start; j < G->group_end; j++)
      if (j != G->selector_value) f = gl2_mul(f, gl2_sub(gl2_from_base(j), s));
    if (c->num_selectors > 1) f = gl2_mul(f, gl2_sub(gl2_from_base(ORC_UNUSED_SELECTOR), s));
    for (uint32_t k = 0; k < CH; k++) out[k] = gl2_add(out[k], gl2_mul(f, acc[k]));
  }
}

size_t orc_check_witness(const orc_circuit *c, const uint64_t *wires, const uint64_t *pis_in, uint64_t first_bad[2]) {
  if (!c->cs_values) return (size_t)-1; /* verifier-only circuit: no preprocessed values */
  size_t n = c->n, bad = 0;
  uint32_t W = c->p.num_wires, NC = c->p.num_constants;
  uint64_t *pis = (uint64_t *)xmalloc((c->npi + 1) * 8);
  for (uint32_t i = 0; i < c->npi; i++) pis[i] = gl_canon(pis_in[i]);
  uint64_t pi_hash[4];
  orc_hash_no_pad(pis, c->npi, pi_hash);
  uint64_t *w = (uint64_t *)xmalloc(W * 8), *k = (uint64_t *)xmalloc(NC * 8), *raw = (uint64_t *)xmalloc((c->max_gate_constraints + 1) * 8);
  uint64_t alphas[4] = {1, 1, 1, 1}, out[4];
  for (size_t r = 0; r < n; r++) {
    for (uint32_t j = 0; j < W; j++) w[j] = gl_canon(wires[j * n + r]);
    for (uint32_t j = 0; j < NC; j++) k[j] = c->cs_values[j * n + r];
    for (uint32_t g = 0; g < c->num_gates; g++) {
      const orc_gate *G = &c->gates[g];
      if (k[G->selector_index] != G->selector_value) continue;
      eval_gates_base(c, w, k, pi_hash, alphas, out, (int)g, raw);
      for (uint32_t i = 0; i < G->num_constraints; i++)
        if (raw[i]) { if (!bad && first_bad) { first_bad[0] = r; first_bad[1] = i; } bad++; }
    }
  }
  free(pis); free(w); free(k); free(raw);
  return bad;
}

/* ------------------------------------------------------------------ prover pieces */
static uint64_t *subgroup_table(unsigned lg) {
  size_t n = (size_t)1 << lg;
  uint64_t *t = (uint64_t *)xmalloc(n * 8), w = gl_root_of_unity(lg);
  t[0] = 1;
  for (size_t i = 1; i < n; i++) t[i] = gl_mul(t[i - 1], w);
  return t;
}

/* wires_permutation_partial_products_and_zs for every challenge; output polys (values on H), column-major:
 * [Z_0 .. Z_{CH-1}, pp_{0,0..npp-1}, pp_{1,0..npp-1}, ...] */
static uint64_t *partial_products_and_zs(const orc_circuit *c, const uint64_t *wires, const uint64_t *betas, const uint64_t *gammas) {
  const orc_params *p = &c->p;
  size_t n = c->n, NR = p->num_routed_wires, NC = p->num_constants, CH = p->num_challenges, Q = p->quotient_degree_factor;
  size_t nchunks = (NR + Q - 1) / Q, npp = nchunks - 1;
  uint64_t *out = (uint64_t *)xmalloc(CH * (1 + npp) * n * 8);
  uint64_t *sub = subgroup_table(p->degree_bits);
  uint64_t *qc = (uint64_t *)xmalloc(n * nchunks * 8); /* quotient chunk products per row */
  for (size_t ch = 0; ch < CH; ch++) {
    uint64_t beta = betas[ch], gamma = gammas[ch];
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
      uint64_t x = sub[i];
      for (size_t k = 0; k < nchunks; k++) {
        uint64_t prod = 1;
        for (size_t j = k * Q; j < NR && j < (k + 1) * Q; j++) {
          uint64_t wv = gl_canon(wires[j * n + i]);
          uint64_t num = gl_add(gl_add(wv, gl_mul(beta, gl_mul(c->k_is[j], x))), gamma);
          uint64_t den = gl_add(gl_add(wv, gl_mul(beta, c->cs_values[(NC + j) * n + i])), gamma);
          prod = gl_mul(prod, gl_mul(num, gl_inv(den)));
        }
        qc[i * nchunks + k] = prod;
      }
    }
    uint64_t z = 1;
    uint64_t *Z = out + ch * n, *PP = out + (CH + ch * npp) * n;
    for (size_t i = 0; i < n; i++) {
      Z[i] = z;
      uint64_t acc = z;
      for (size_t k = 0; k < nchunks; k++) {
        acc = gl_mul(acc, qc[i * nchunks + k]);
        if (k < npp) PP[k * n + i] = acc;
      }
      z = acc; /* Z(g x) */
    }
  }
  free(sub); free(qc);
  return out;
}

/* compute_quotient_polys: returns the CH*Q coefficient chunks, column-major [CH*Q][n] */
static uint64_t *quotient_chunks(const orc_circuit *c, const orc_batch *wires, const orc_batch *zs, const uint64_t *pis,
                                 const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas) {
  const orc_params *p = &c->p;
  size_t n = c->n, N = n << p->rate_bits, NR = p->num_routed_wires, NC = p->num_constants, CH = p->num_challenges, Q = p->quotient_degree_factor;
  size_t nchunks = (NR + Q - 1) / Q, npp = nchunks - 1;
  unsigned lgN = p->degree_bits + p->rate_bits;
  uint64_t *pts = subgroup_table(lgN);
  uint64_t *vals = (uint64_t *)xmalloc(CH * N * 8);
  /* Z_H(7 w^i) takes 2^rate_bits values */
  uint64_t zh_inv[256];
  uint64_t shift_n = gl_pow(GL_GENERATOR, n);
  for (size_t r = 0; r < ((size_t)1 << p->rate_bits); r++)
    zh_inv[r] = gl_inv(gl_sub(gl_mul(shift_n, gl_pow(gl_root_of_unity(p->rate_bits), r)), 1));
  uint64_t ninv = gl_inv((uint64_t)n % GL_P);
  size_t next_step = (size_t)1 << p->rate_bits; /* quotient domain = LDE domain, so step = 1 and next row = +2^rate_bits */
#pragma omp parallel for schedule(dynamic, 64)
  for (size_t i = 0; i < N; i++) {
    uint64_t x = gl_mul(GL_GENERATOR, pts[i]);
    const uint64_t *cs = batch_lde_row(c->cs, i), *w = batch_lde_row(wires, i), *zp = batch_lde_row(zs, i);
    const uint64_t *zp_next = batch_lde_row(zs, (i + next_step) % N);
    uint64_t zh = gl_sub(gl_mul(shift_n, gl_pow(gl_root_of_unity(p->rate_bits), i % next_step)), 1);
    uint64_t l0 = gl_mul(gl_mul(zh, ninv), gl_inv(gl_sub(x, 1))); /* eval_l_0 ; x != 1 on the coset */
    uint64_t terms[4 + 4 * 16];
    size_t nt = 0;
    for (size_t ch = 0; ch < CH; ch++) terms[nt++] = gl_mul(l0, gl_sub(zp[ch], 1));
    for (size_t ch = 0; ch < CH; ch++) {
      uint64_t prev = zp[ch];
      for (size_t k = 0; k < nchunks; k++) {
        uint64_t pn = 1, pd = 1;
        for (size_t j = k * Q; j < NR && j < (k + 1) * Q; j++) {
          pn = gl_mul(pn, gl_add(gl_add(w[j], gl_mul(betas[ch], gl_mul(c->k_is[j], x))), gammas[ch]));
          pd = gl_mul(pd, gl_add(gl_add(w[j], gl_mul(betas[ch], cs[NC + j])), gammas[ch]));
        }
        uint64_t next = k < npp ? zp[CH + ch * npp + k] : zp_next[ch];
        terms[nt++] = gl_sub(gl_mul(prev, pn), gl_mul(next, pd));
        prev = next;
      }
    }
    uint64_t gates[4];
    eval_gates_base(c, w, cs, pis, alphas, gates, -1, NULL);
    for (size_t ch = 0; ch < CH; ch++) {
      uint64_t a = alphas[ch], acc = gates[ch];
      for (size_t t = nt; t-- > 0;) acc = gl_add(gl_mul(acc, a), terms[t]); /* reduce_with_powers over [terms, gate constraints] */
      vals[ch * N + i] = gl_mul(acc, zh_inv[i % next_step]);
    }
  }
  free(pts);
  uint64_t *chunks = (uint64_t *)xmalloc(CH * Q * n * 8);
  for (size_t ch = 0; ch < CH; ch++) {
    orc_coset_ifft(vals + ch * N, N, GL_GENERATOR);
    memcpy(chunks + ch * Q * n, vals + ch * N, N * 8); /* N = Q * n: chunk k = coefficients [k n, (k+1) n) */
  }
  free(vals);
  return chunks;
}

static gl2_t eval_poly_ext(const uint64_t *coeffs, size_t n, gl2_t z) {
  gl2_t acc = gl2_from_base(0);
  for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, z), gl2_from_base(coeffs[i]));
  return acc;
}

/* ------------------------------------------------------------------ FRI prover */
typedef struct { size_t nleaves, leaf_len; uint64_t *leaves; orc_merkle *tree; } fri_layer;

static int leading_zeros64(uint64_t x) { int n = 0; if (!x) return 64; while (!(x >> 63)) { x <<= 1; n++; } return n; }

static uint64_t fri_proof_of_work(challenger *ch, unsigned pow_bits) {
  uint64_t base[12];
  memcpy(base, ch->state, sizeof base);
  for (int i = 0; i < ch->nin; i++) base[i] = ch->in[i];
  int pos = ch->nin;
  uint64_t found = 0;
  int have = 0;
  /* minimum valid witness: scan candidates in blocks, keep the smallest hit */
  for (uint64_t start = 0; !have; start += 1 << 14) {
    uint64_t best = ~0ull;
#pragma omp parallel for schedule(static) reduction(min : best)
    for (uint64_t w = start; w < start + (1 << 14); w++) {
      uint64_t s[12];
      memcpy(s, base, sizeof s);
      s[pos] = w;
      orc_poseidon_permute(s);
      if (leading_zeros64(s[7]) >= (int)pow_bits && w < best) best = w;
    }
    if (best != ~0ull) { found = best; have = 1; }
  }
  ch_observe(ch, found);
  uint64_t resp = ch_get(ch);
  if (leading_zeros64(resp) < (int)pow_bits) { fprintf(stderr, "oracle: PoW self-check failed\n"); abort(); }
  return found;
}

/* ------------------------------------------------------------------ prove */
int orc_prove(const orc_circuit *cc, const uint64_t *wires_in, const uint64_t *pis_in, uint64_t *proof) {
  orc_circuit *c = (orc_circuit *)cc; /* only `last` is written */
  if (!c->cs) return -1; /* verifier-only or unbuilt circuit */
  const orc_params *p = &c->p;
  size_t n = c->n, N = n << p->rate_bits, W = p->num_wires, NR = p->num_routed_wires, NC = p->num_constants, CH = p->num_challenges,
         Q = p->quotient_degree_factor;
  size_t npp = NPP(*p);
  unsigned lgN = p->degree_bits + p->rate_bits;
  layout_t L;
  make_layout(p, &L);
  memset(proof, 0, L.total * 8);
  uint64_t *pis = (uint64_t *)xmalloc((c->npi + 1) * 8);
  for (uint32_t i = 0; i < c->npi; i++) pis[i] = gl_canon(pis_in[i]);
  uint64_t pi_hash[4];
  orc_hash_no_pad(pis, c->npi, pi_hash);

  uint64_t *wires = (uint64_t *)xmalloc(W * n * 8);
  for (size_t i = 0; i < W * n; i++) wires[i] = gl_canon(wires_in[i]);
  orc_batch *wb = batch_from_values(wires, W, p->degree_bits, p->rate_bits, p->cap_height);
  memcpy(proof + L.wires_cap, wb->tree->cap, L.capw * 8);

  challenger ch;
  ch_init(&ch);
  ch_observe_n(&ch, c->digest, 4);
  ch_observe_n(&ch, pi_hash, 4);
  ch_observe_n(&ch, wb->tree->cap, L.capw);
  uint64_t betas[4], gammas[4], alphas[4];
  for (size_t k = 0; k < CH; k++) betas[k] = ch_get(&ch);
  for (size_t k = 0; k < CH; k++) gammas[k] = ch_get(&ch);

  uint64_t *zs_vals = partial_products_and_zs(c, wires, betas, gammas);
  orc_batch *zb = batch_from_values(zs_vals, CH * (1 + npp), p->degree_bits, p->rate_bits, p->cap_height);
  free(zs_vals);
  memcpy(proof + L.zs_cap, zb->tree->cap, L.capw * 8);
  ch_observe_n(&ch, zb->tree->cap, L.capw);
  for (size_t k = 0; k < CH; k++) alphas[k] = ch_get(&ch);

  uint64_t *qchunks = quotient_chunks(c, wb, zb, pi_hash, betas, gammas, alphas);
  orc_batch *qb = batch_from_coeffs_owned(qchunks, CH * Q, p->degree_bits, p->rate_bits, p->cap_height);
  memcpy(proof + L.quot_cap, qb->tree->cap, L.capw * 8);
  ch_observe_n(&ch, qb->tree->cap, L.capw);
  gl2_t zeta = ch_get_ext(&ch);
  gl2_t g_zeta = gl2_scale(zeta, gl_root_of_unity(p->degree_bits));

  /* OpeningSet::new */
  const orc_batch *oracles[4] = {c->cs, wb, zb, qb};
  size_t op_off[4] = {L.op_constants, L.op_wires, L.op_zs, L.op_quot};
  for (int o = 0; o < 4; o++) {
    const orc_batch *b = oracles[o];
#pragma omp parallel for schedule(dynamic)
    for (size_t j = 0; j < b->ncols; j++) {
      gl2_t v = eval_poly_ext(b->coeffs + j * n, n, zeta);
      size_t dst;
      if (o == 2) dst = j < CH ? L.op_zs + 2 * j : L.op_pp + 2 * (j - CH);
      else dst = op_off[o] + 2 * j; /* oracle 0: constants then sigmas are contiguous */
      proof[dst] = v.c[0]; proof[dst + 1] = v.c[1];
    }
  }
  for (size_t j = 0; j < CH; j++) {
    gl2_t v = eval_poly_ext(zb->coeffs + j * n, n, g_zeta);
    proof[L.op_zs_next + 2 * j] = v.c[0]; proof[L.op_zs_next + 2 * j + 1] = v.c[1];
  }
  /* observe_openings(to_fri_openings): zeta batch = constants, sigmas, wires, zs, partial products, quotient; then zs_next */
  ch_observe_n(&ch, proof + L.op_constants, 2 * (NC + NR + W));
  ch_observe_n(&ch, proof + L.op_zs, 2 * CH);
  ch_observe_n(&ch, proof + L.op_pp, 2 * CH * npp);
  ch_observe_n(&ch, proof + L.op_quot, 2 * CH * Q);
  ch_observe_n(&ch, proof + L.op_zs_next, 2 * CH);

  /* prove_openings */
  gl2_t alpha = ch_get_ext(&ch);
  gl2_t *final_poly = (gl2_t *)xcalloc(N, sizeof(gl2_t)); /* lde(rate_bits): zero padded to N */
  {
    /* batch 0: every polynomial of every oracle at zeta */
    gl2_t *comp = (gl2_t *)xcalloc(n, sizeof(gl2_t));
    gl2_t apow = gl2_from_base(1);
    size_t count0 = 0;
    for (int o = 0; o < 4; o++)
      for (size_t j = 0; j < oracles[o]->ncols; j++) {
        const uint64_t *cf = oracles[o]->coeffs + j * n;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) comp[i] = gl2_add(comp[i], gl2_scale(apow, cf[i]));
        apow = gl2_mul(apow, alpha);
        count0++;
      }
    /* divide_by_linear(zeta), padded back with a zero */
    gl2_t *quot0 = (gl2_t *)xcalloc(n, sizeof(gl2_t));
    gl2_t acc = gl2_from_base(0);
    for (size_t i = n; i-- > 0;) { acc = gl2_add(gl2_mul(acc, zeta), comp[i]); if (i > 0) quot0[i - 1] = acc; }
    /* batch 1: the Z polynomials at g*zeta */
    memset(comp, 0, n * sizeof(gl2_t));
    apow = gl2_from_base(1);
    for (size_t j = 0; j < CH; j++) {
      const uint64_t *cf = zb->coeffs + j * n;
      for (size_t i = 0; i < n; i++) comp[i] = gl2_add(comp[i], gl2_scale(apow, cf[i]));
      apow = gl2_mul(apow, alpha);
    }
    gl2_t *quot1 = (gl2_t *)xcalloc(n, sizeof(gl2_t));
    acc = gl2_from_base(0);
    for (size_t i = n; i-- > 0;) { acc = gl2_add(gl2_mul(acc, g_zeta), comp[i]); if (i > 0) quot1[i - 1] = acc; }
    /* final = quot0 * alpha^(count of batch 1) + quot1   (ReducingFactor::shift_poly) */
    gl2_t sh = gl2_pow(alpha, CH);
    for (size_t i = 0; i < n; i++) final_poly[i] = gl2_add(gl2_mul(quot0[i], sh), quot1[i]);
    free(comp); free(quot0); free(quot1);
    (void)count0;
  }
  /* fri_committed_trees */
  size_t cur_len = N; /* coefficient vector length (zero padded), values live on shift * <w_cur_len> */
  uint64_t *c0 = (uint64_t *)xmalloc(N * 8), *c1 = (uint64_t *)xmalloc(N * 8);
  for (size_t i = 0; i < N; i++) { c0[i] = final_poly[i].c[0]; c1[i] = final_poly[i].c[1]; }
  uint64_t shift = GL_GENERATOR;
  fri_layer layers[ORC_MAX_FRI_LAYERS];
  gl2_t fri_betas[ORC_MAX_FRI_LAYERS];
  for (uint32_t l = 0; l < p->num_fri_layers; l++) {
    unsigned ab = p->fri_arity_bits[l];
    size_t arity = (size_t)1 << ab;
    unsigned lg = gl_log2(cur_len);
    /* values = coeffs.coset_fft(shift), then reverse_index_bits, chunk by arity, flatten */
    uint64_t *v0 = (uint64_t *)xmalloc(cur_len * 8), *v1 = (uint64_t *)xmalloc(cur_len * 8);
    memcpy(v0, c0, cur_len * 8); memcpy(v1, c1, cur_len * 8);
    orc_coset_fft(v0, cur_len, shift); orc_coset_fft(v1, cur_len, shift);
    fri_layer *F = &layers[l];
    F->nleaves = cur_len >> ab; F->leaf_len = 2 * arity;
    F->leaves = (uint64_t *)xmalloc(cur_len * 2 * 8);
    for (size_t i = 0; i < cur_len; i++) { size_t s = gl_bitrev(i, lg); F->leaves[2 * i] = v0[s]; F->leaves[2 * i + 1] = v1[s]; }
    free(v0); free(v1);
    F->tree = orc_merkle_build(F->leaves, F->nleaves, F->leaf_len, p->cap_height);
    memcpy(proof + L.fri_caps + l * L.capw, F->tree->cap, L.capw * 8);
    ch_observe_n(&ch, F->tree->cap, L.capw);
    gl2_t beta = ch_get_ext(&ch);
    fri_betas[l] = beta;
    /* coeffs = chunks_exact(arity).map(|chunk| reduce_with_powers(chunk, beta)) */
    size_t nl = cur_len >> ab;
    for (size_t k = 0; k < nl; k++) {
      gl2_t acc = gl2_from_base(0);
      for (size_t j = arity; j-- > 0;) acc = gl2_add(gl2_mul(acc, beta), gl2_make(c0[k * arity + j], c1[k * arity + j]));
      c0[k] = acc.c[0]; c1[k] = acc.c[1];
    }
    cur_len = nl;
    shift = gl_pow(shift, arity);
  }
  /* final polynomial: truncate the (zero) padding */
  for (size_t i = 0; i < L.final_len; i++) { proof[L.final_poly + 2 * i] = c0[i]; proof[L.final_poly + 2 * i + 1] = c1[i]; }
  ch_observe_n(&ch, proof + L.final_poly, 2 * L.final_len);
  uint64_t pow_witness = fri_proof_of_work(&ch, p->proof_of_work_bits);
  proof[L.pow_witness] = pow_witness;

  /* query rounds */
  for (uint32_t q = 0; q < p->num_query_rounds; q++) {
    size_t x_index = (size_t)(ch_get(&ch) % N);
    c->last.query_indices[q] = x_index;
    uint64_t *R = proof + L.queries + q * L.query_words;
    for (int o = 0; o < 4; o++) {
      memcpy(R + L.q_init_off[o], oracles[o]->leaves + x_index * oracles[o]->ncols, oracles[o]->ncols * 8);
      orc_merkle_prove(oracles[o]->tree, x_index, R + L.q_init_off[o] + oracles[o]->ncols);
    }
    size_t xi = x_index;
    for (uint32_t l = 0; l < p->num_fri_layers; l++) {
      unsigned ab = p->fri_arity_bits[l];
      size_t leaf = xi >> ab;
      memcpy(R + L.q_step_off[l], layers[l].leaves + leaf * layers[l].leaf_len, layers[l].leaf_len * 8);
      orc_merkle_prove(layers[l].tree, leaf, R + L.q_step_off[l] + layers[l].leaf_len);
      xi = leaf;
    }
  }
  /* record challenges for stage-wise parity tests */
  memcpy(c->last.betas, betas, sizeof betas); memcpy(c->last.gammas, gammas, sizeof gammas); memcpy(c->last.alphas, alphas, sizeof alphas);
  c->last.zeta[0] = zeta.c[0]; c->last.zeta[1] = zeta.c[1];
  c->last.fri_alpha[0] = alpha.c[0]; c->last.fri_alpha[1] = alpha.c[1];
  for (uint32_t l = 0; l < p->num_fri_layers; l++) { c->last.fri_betas[l][0] = fri_betas[l].c[0]; c->last.fri_betas[l][1] = fri_betas[l].c[1]; }
  c->last.pow_witness = pow_witness;

  for (uint32_t l = 0; l < p->num_fri_layers; l++) { free(layers[l].leaves); orc_merkle_free(layers[l].tree); }
  free(c0); free(c1); free(final_poly); free(pis); free(wires);
  batch_free(wb); batch_free(zb); batch_free(qb);
  (void)lgN;
  return 0;
}

/* ------------------------------------------------------------------ verify */
static gl2_t rd2(const uint64_t *p) { return gl2_make(p[0], p[1]); }

/* compute_evaluation (fri/verifier.rs): P(beta) for the degree < arity interpolant through the coset of x */
static gl2_t fri_compute_evaluation(uint64_t x, size_t x_index_within_coset, unsigned arity_bits, const gl2_t *evals_bitrev, gl2_t beta) {
  size_t arity = (size_t)1 << arity_bits;
  uint64_t g = gl_root_of_unity(arity_bits);
  gl2_t ev[64];
  for (size_t i = 0; i < arity; i++) ev[gl_bitrev(i, arity_bits)] = evals_bitrev[i]; /* reverse_index_bits_in_place */
  size_t rev = gl_bitrev(x_index_within_coset, arity_bits);
  uint64_t coset_start = gl_mul(x, gl_pow(g, arity - rev));
  uint64_t pts[64];
  uint64_t y = 1;
  for (size_t i = 0; i < arity; i++) { pts[i] = gl_mul(coset_start, y); y = gl_mul(y, g); }
  /* Lagrange interpolation at beta */
  gl2_t acc = gl2_from_base(0);
  for (size_t i = 0; i < arity; i++) {
    gl2_t num = gl2_from_base(1);
    uint64_t den = 1;
    for (size_t j = 0; j < arity; j++) {
      if (j == i) continue;
      num = gl2_mul(num, gl2_sub(beta, gl2_from_base(pts[j])));
      den = gl_mul(den, gl_sub(pts[i], pts[j]));
    }
    acc = gl2_add(acc, gl2_mul(ev[i], gl2_scale(num, gl_inv(den))));
  }
  return acc;
}

/* codes: 1 encoding, 2 proof of work, 3 vanishing identity, 4 initial Merkle proof, 5 FRI consistency,
 * 6 FRI layer Merkle proof, 7 final polynomial */
int orc_verify(const orc_circuit *c, const uint64_t *proof, const uint64_t *pis_in) {
  const orc_params *p = &c->p;
  size_t n = c->n, N = n << p->rate_bits, W = p->num_wires, NR = p->num_routed_wires, NC = p->num_constants, CH = p->num_challenges,
         Q = p->quotient_degree_factor;
  size_t nchunks = (NR + Q - 1) / Q, npp = nchunks - 1;
  unsigned lgN = p->degree_bits + p->rate_bits;
  layout_t L;
  make_layout(p, &L);
  for (size_t i = 0; i < L.total; i++) if (proof[i] >= GL_P) return 1;
  uint64_t *pis = (uint64_t *)xmalloc((c->npi + 1) * 8);
  for (uint32_t i = 0; i < c->npi; i++) pis[i] = gl_canon(pis_in[i]);
  uint64_t pi_hash[4];
  orc_hash_no_pad(pis, c->npi, pi_hash);
  int rc = 0;
  gl2_t *ow = NULL, *oc = NULL;

  challenger ch;
  ch_init(&ch);
  ch_observe_n(&ch, c->digest, 4);
  ch_observe_n(&ch, pi_hash, 4);
  ch_observe_n(&ch, proof + L.wires_cap, L.capw);
  uint64_t betas[4], gammas[4], alphas[4];
  for (size_t k = 0; k < CH; k++) betas[k] = ch_get(&ch);
  for (size_t k = 0; k < CH; k++) gammas[k] = ch_get(&ch);
  ch_observe_n(&ch, proof + L.zs_cap, L.capw);
  for (size_t k = 0; k < CH; k++) alphas[k] = ch_get(&ch);
  ch_observe_n(&ch, proof + L.quot_cap, L.capw);
  gl2_t zeta = ch_get_ext(&ch);
  ch_observe_n(&ch, proof + L.op_constants, 2 * (NC + NR + W));
  ch_observe_n(&ch, proof + L.op_zs, 2 * CH);
  ch_observe_n(&ch, proof + L.op_pp, 2 * CH * npp);
  ch_observe_n(&ch, proof + L.op_quot, 2 * CH * Q);
  ch_observe_n(&ch, proof + L.op_zs_next, 2 * CH);
  gl2_t fri_alpha = ch_get_ext(&ch);
  gl2_t fri_betas[ORC_MAX_FRI_LAYERS];
  for (uint32_t l = 0; l < p->num_fri_layers; l++) { ch_observe_n(&ch, proof + L.fri_caps + l * L.capw, L.capw); fri_betas[l] = ch_get_ext(&ch); }
  ch_observe_n(&ch, proof + L.final_poly, 2 * L.final_len);
  ch_observe(&ch, proof[L.pow_witness]);
  if (leading_zeros64(ch_get(&ch)) < (int)p->proof_of_work_bits) { rc = 2; goto done; }

  /* ---- vanishing(zeta) == Z_H(zeta) * t(zeta)   (verify_with_challenges + eval_vanishing_poly) */
  ow = (gl2_t *)xmalloc(W * sizeof(gl2_t)); oc = (gl2_t *)xmalloc((NC + NR) * sizeof(gl2_t));
  for (size_t j = 0; j < W; j++) ow[j] = rd2(proof + L.op_wires + 2 * j);
  for (size_t j = 0; j < NC + NR; j++) oc[j] = rd2(proof + L.op_constants + 2 * j);
  {
    gl2_t zeta_n = zeta;
    for (unsigned i = 0; i < p->degree_bits; i++) zeta_n = gl2_mul(zeta_n, zeta_n);
    gl2_t zh = gl2_sub(zeta_n, gl2_from_base(1));
    gl2_t one = gl2_from_base(1);
    gl2_t l0 = gl2_eq(zeta, one) ? one
                                 : gl2_mul(zh, gl2_inv(gl2_scale(gl2_sub(zeta, one), (uint64_t)n % GL_P)));
    gl2_t terms[4 + 4 * 16];
    size_t nt = 0;
    for (size_t k = 0; k < CH; k++) terms[nt++] = gl2_mul(l0, gl2_sub(rd2(proof + L.op_zs + 2 * k), one));
    for (size_t k = 0; k < CH; k++) {
      gl2_t prev = rd2(proof + L.op_zs + 2 * k);
      for (size_t cidx = 0; cidx < nchunks; cidx++) {
        gl2_t pn = one, pd = one;
        for (size_t j = cidx * Q; j < NR && j < (cidx + 1) * Q; j++) {
          gl2_t sid = gl2_scale(zeta, c->k_is[j]);
          pn = gl2_mul(pn, gl2_add(gl2_add(ow[j], gl2_scale(sid, betas[k])), gl2_from_base(gammas[k])));
          pd = gl2_mul(pd, gl2_add(gl2_add(ow[j], gl2_scale(oc[NC + j], betas[k])), gl2_from_base(gammas[k])));
        }
        gl2_t next = cidx < npp ? rd2(proof + L.op_pp + 2 * (k * npp + cidx)) : rd2(proof + L.op_zs_next + 2 * k);
        terms[nt++] = gl2_sub(gl2_mul(prev, pn), gl2_mul(next, pd));
        prev = next;
      }
    }
    gl2_t gates[4];
    eval_gates_ext(c, ow, oc, pi_hash, alphas, gates);
    for (size_t k = 0; k < CH; k++) {
      gl2_t acc = gates[k];
      for (size_t t = nt; t-- > 0;) acc = gl2_add(gl2_scale(acc, alphas[k]), terms[t]);
      gl2_t tq = gl2_from_base(0); /* reduce_with_powers(quotient chunk openings, zeta^n) */
      for (size_t j = Q; j-- > 0;) tq = gl2_add(gl2_mul(tq, zeta_n), rd2(proof + L.op_quot + 2 * (k * Q + j)));
      if (!gl2_eq(acc, gl2_mul(zh, tq))) { rc = 3; goto done; }
    }
  }
  /* ---- FRI (verify_fri_proof) */
  {
    /* PrecomputedReducedOpenings */
    gl2_t red0 = gl2_from_base(0), red1 = gl2_from_base(0);
    {
      /* batch 0 values in order: constants, sigmas, wires, zs, pp, quotient  -> sum alpha^j v_j */
      size_t tot0 = NC + NR + W + CH + CH * npp + CH * Q;
      gl2_t *v = (gl2_t *)xmalloc(tot0 * sizeof(gl2_t));
      size_t k = 0;
      for (size_t j = 0; j < NC + NR + W; j++) v[k++] = rd2(proof + L.op_constants + 2 * j);
      for (size_t j = 0; j < CH; j++) v[k++] = rd2(proof + L.op_zs + 2 * j);
      for (size_t j = 0; j < CH * npp; j++) v[k++] = rd2(proof + L.op_pp + 2 * j);
      for (size_t j = 0; j < CH * Q; j++) v[k++] = rd2(proof + L.op_quot + 2 * j);
      for (size_t j = tot0; j-- > 0;) red0 = gl2_add(gl2_mul(red0, fri_alpha), v[j]);
      for (size_t j = CH; j-- > 0;) red1 = gl2_add(gl2_mul(red1, fri_alpha), rd2(proof + L.op_zs_next + 2 * j));
      free(v);
    }
    gl2_t g_zeta = gl2_scale(zeta, gl_root_of_unity(p->degree_bits));
    const uint64_t *caps[4] = {c->cs_cap, proof + L.wires_cap, proof + L.zs_cap, proof + L.quot_cap};
    gl2_t alpha_ch = gl2_pow(fri_alpha, CH);
    for (uint32_t q = 0; q < p->num_query_rounds && !rc; q++) {
      size_t x_index = (size_t)(ch_get(&ch) % N);
      const uint64_t *R = proof + L.queries + q * L.query_words;
      for (int o = 0; o < 4; o++)
        if (!orc_merkle_verify(R + L.q_init_off[o], L.q_init_cols[o], x_index, R + L.q_init_off[o] + L.q_init_cols[o], (unsigned)L.q_init_sib, caps[o])) { rc = 4; break; }
      if (rc) break;
      uint64_t subgroup_x = gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(lgN), gl_bitrev(x_index, lgN)));
      /* fri_combine_initial */
      gl2_t sum;
      {
        gl2_t r0 = gl2_from_base(0);
        for (int o = 3; o >= 0; o--)
          for (size_t j = L.q_init_cols[o]; j-- > 0;) r0 = gl2_add(gl2_mul(r0, fri_alpha), gl2_from_base(R[L.q_init_off[o] + j]));
        gl2_t xs = gl2_from_base(subgroup_x);
        sum = gl2_mul(gl2_sub(r0, red0), gl2_inv(gl2_sub(xs, zeta)));
        gl2_t r1 = gl2_from_base(0);
        for (size_t j = CH; j-- > 0;) r1 = gl2_add(gl2_mul(r1, fri_alpha), gl2_from_base(R[L.q_init_off[2] + j]));
        sum = gl2_mul(sum, alpha_ch);
        sum = gl2_add(sum, gl2_mul(gl2_sub(r1, red1), gl2_inv(gl2_sub(xs, g_zeta))));
      }
      gl2_t old_eval = sum;
      size_t xi = x_index;
      for (uint32_t l = 0; l < p->num_fri_layers; l++) {
        unsigned ab = p->fri_arity_bits[l];
        size_t arity = (size_t)1 << ab;
        gl2_t evals[64];
        for (size_t j = 0; j < arity; j++) evals[j] = rd2(R + L.q_step_off[l] + 2 * j);
        size_t coset_index = xi >> ab, within = xi & (arity - 1);
        if (!gl2_eq(evals[within], old_eval)) { rc = 5; break; }
        old_eval = fri_compute_evaluation(subgroup_x, within, ab, evals, fri_betas[l]);
        if (!orc_merkle_verify(R + L.q_step_off[l], 2 * arity, coset_index, R + L.q_step_off[l] + 2 * arity, (unsigned)L.q_step_sib[l],
                               proof + L.fri_caps + l * L.capw)) { rc = 6; break; }
        for (unsigned i = 0; i < ab; i++) subgroup_x = gl_sqr(subgroup_x);
        xi = coset_index;
      }
      if (rc) break;
      gl2_t fv = gl2_from_base(0), xs = gl2_from_base(subgroup_x);
      for (size_t j = L.final_len; j-- > 0;) fv = gl2_add(gl2_mul(fv, xs), rd2(proof + L.final_poly + 2 * j));
      if (!gl2_eq(fv, old_eval)) { rc = 7; break; }
    }
  }
done:
  free(pis); free(ow); free(oc);
  return rc;
}
