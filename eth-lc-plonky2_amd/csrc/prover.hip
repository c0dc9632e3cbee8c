// Host orchestration of the prover: lcp2_circuit_create (= build()), lcp2_prove (= data.prove(pw)).
//
// Follows plonky2 0.1.4 plonk/prover.rs::prove step by step; every heavy step is a
// kernel from kernels_*.hip on the context's stream.  The Fiat-Shamir challenger (row a14: a few hundred
// permutations) runs on the host between the commitments; it needs only the 512-byte caps, the opening
// values and the final polynomial, which are the only device-to-host copies before the query phase.
#include <cstring>
#include "host_protocol.hpp"
#include "internal.hpp"
#include "ntt_host.hpp"
#include "prover_kernels.hpp"

using namespace lcp2;

// a batched opening in flight: lcp2_fri_open runs its three phases back to back, a coset-sharded proof exchanges the
// openings after the first and the cap of FRI layer 0 after the second
struct FriOpenState {
  gl2 zeta{}, alpha{};
  HostChallenger ch;
  gl2 fri_betas[LCP2_MAX_FRI_LAYERS] = {};
  u64 pow_witness = 0;
  std::vector<u64> idx;
  int phase = 0;  // 0: none, 1: openings evaluated, 2: final polynomial composed and FRI layer 0 committed
};

struct lcp2_circuit {
  lcp2_ctx *ctx = nullptr;
  lcp2_params p{};
  uint32_t npi = 0, num_selectors = 0, num_regs = 1, dev_regs = 1;
  std::vector<GateDev> dev_gates;  // the gate table as uploaded: offsets into the staged code
  std::vector<lcp2_gate> gates;
  std::vector<uint32_t> code;
  std::vector<u64> imm, k_is;
  u64 digest[4] = {0, 0, 0, 0};
  std::vector<u64> cs_cap;
  u64 last_challenges[97] = {0};
  // device: description
  DevBuf d_gates, d_code, d_stage, d_imm, d_kis, d_l0, d_zh_inv, cs_values;
  lcp2_oracle cs;  // constants_sigmas commitment
  // device: per-proof workspace (allocated once)
  lcp2_oracle wires, zs, quot;
  DevBuf wires_vals, zs_vals, chunk_q, row_tot, scan_tmp, qvals, planes, small, partial, tables, alpha_limbs, open_out;
  DevBuf fri_c[2];                       // ping-pong coefficient planes [2][m]
  std::vector<DevBuf> fri_vals, fri_dig; // per layer: value planes [2][8 m_l], digests
  std::vector<std::vector<u64>> fri_level_off;
  std::vector<DevBuf> fri_d_level_off;
  DevBuf q_idx, q_buf;
  // staged proving (lcp2_commit_wires -> lcp2_perm_zs -> lcp2_quotient -> lcp2_fri_open)
  const u64 *d_wires_cur = nullptr;
  enum Stage { ST_NONE = 0, ST_WIRES, ST_ZS, ST_QVALS, ST_QUOT };  // what the handle holds of the proof in flight
  Stage stage = ST_NONE;
  FriOpenState fo;
  // coset-sharded circuit: this handle holds the leaf blocks [bf, bf + bc) of every LDE and Merkle tree;
  // bc = 0: all of them.  cap_final: the full constants_sigmas cap (hence the digest) is known.
  uint32_t bf = 0, bc = 0;
  bool cap_final = true;
  bool sharded() const { return bc != 0; }
  uint32_t nblocks() const { return bc ? bc : (1u << p.rate_bits); }
  // row exchange form of a sharded proof (lcp2_commit_wires_rows): the handle holds the witness VALUES of the rows
  // [row0(), row0() + rows()) only - rank r of `world` holds the r-th block of n / world rows - and runs the permutation
  // argument and the gate check on them
  bool rows_mode = false, cs_rows_ready = false;
  int perm_phase = 0;  // row exchange form: 1 after lcp2_perm_zs_rows_begin, 2 after _finish (the order is enforced: _commit reads what they wrote)
  DevBuf cs_rows;   // the constants on this rank's rows, [num_constants][rows()]
  DevBuf zs_rows;   // exchange buffer of Z / partial products, [world][num_challenges * (1 + npp)][rows()]
  // a sharded circuit with at most 8 blocks interpolates the quotient coset by coset (each rank its own blocks, before the
  // exchange of the planes) and combines the interpolants into the chunks afterwards: no rank transforms 2^rate_bits n points
  DevBuf q_combine;  // the combining matrix [R][R] (k_quotient_combine)
  bool local_quotient() const { return sharded() && p.rate_bits <= 3; }
  u64 perm_wrap[2 * QUOTIENT_MAX_CH] = {0};  // per challenge: Z before the block's last row, the last row's quotient (host)
  u64 noncanon_host = 0;   // stage_wires: a witness value was >= p (arrives with the wires cap)
  DevBuf leaf_state;       // chunked commitment of the wires (lcp2_commit_wires_chunk): the sponge state of every local leaf, [12][leaves]
  int chunk_next = -1;     // the column the next chunk must start at; -1: no chunked commitment in progress
  DevBuf wit_slot[2];      // staged host witnesses (lcp2_witness_stage), [num_wires][n] each
  hipEvent_t wit_ready[2] = {nullptr, nullptr};  // the slot's upload has finished (recorded on the context's copy stream)
  bool wit_staged[2] = {false, false};
  ~lcp2_circuit() { for (hipEvent_t e : wit_ready) if (e) (void)hipEventDestroy(e); }
  bool check_pending = false;  // the gate-check verdict of stage_quotient_values has not been read yet (it arrives with the quotient cap)
  uint32_t world() const { return bc ? (1u << p.rate_bits) / bc : 1; }
  uint32_t rank() const { return bc ? bf / bc : 0; }
  u64 rows() const { return rows_mode ? (1ull << p.degree_bits) / world() : (1ull << p.degree_bits); }
  u64 row0() const { return rows_mode ? rows() * rank() : 0; }
};

namespace lcp2 {
// shape checks shared by build() and the verifier-only constructor: everything the prover's workspaces and the verifier's
// fixed-size arrays rely on.  Returns nullptr or the reason; *unsupported says which status it is.
const char *params_problem(const lcp2_params &p, bool *unsupported) {
  *unsupported = false;
  if (p.degree_bits < 1 || p.rate_bits < 1 || p.rate_bits > 8 || p.degree_bits + p.rate_bits > 30) return "degree_bits / rate_bits out of range";
  if (p.num_wires == 0 || p.num_wires > 65535 || p.num_constants > 65535) return "bad column counts";
  if (p.num_routed_wires > p.num_wires || p.num_routed_wires == 0) return "bad routed wire count";
  if (p.cap_height > p.degree_bits + p.rate_bits) return "cap_height exceeds the LDE tree";
  *unsupported = true;
  if (p.quotient_degree_factor != (1u << p.rate_bits)) return "quotient_degree_factor must equal 2^rate_bits";
  if (p.num_challenges < 1 || p.num_challenges > QUOTIENT_MAX_CH) return "num_challenges must be 1 or 2";
  if ((p.num_routed_wires + p.quotient_degree_factor - 1) / p.quotient_degree_factor > PERM_MAX_CHUNKS) return "too many routed wires";
  if (p.num_query_rounds > 64 || p.num_fri_layers > LCP2_MAX_FRI_LAYERS) return "too many queries / layers";
  if (p.proof_of_work_bits < 1 || p.proof_of_work_bits > 40) return "proof_of_work_bits out of range";
  *unsupported = false;
  u32 lg = p.degree_bits + p.rate_bits, d = p.degree_bits;
  for (u32 l = 0; l < p.num_fri_layers; l++) {
    u32 ab = p.fri_arity_bits[l];
    if (ab < 1 || ab > 5 || ab > d || lg - ab < p.cap_height) return "bad FRI arity schedule";
    lg -= ab; d -= ab;
  }
  return nullptr;
}
}  // namespace lcp2

namespace {
#define LCP2_TRY(expr) do { int rc_ = (expr); if (rc_ != LCP2_OK) return rc_; } while (0)

inline u32 npp_of(const lcp2_params &p) { return (p.num_routed_wires + p.quotient_degree_factor - 1) / p.quotient_degree_factor - 1; }

int check_params(lcp2_ctx *ctx, const lcp2_params &p) {
  bool unsupported;
  if (const char *why = params_problem(p, &unsupported)) return ctx->fail(unsupported ? LCP2_E_UNSUPPORTED : LCP2_E_INVALID, why);
  return LCP2_OK;
}

// alpha_c^e, e < QUOTIENT_TERM_POWS, as three 22-bit limbs (QuotientArgs::alpha_limbs)
std::vector<uint32_t> alpha_limb_table(const u64 *alphas, u32 CH) {
  std::vector<uint32_t> t((size_t)QUOTIENT_MAX_CH * QUOTIENT_TERM_POWS * 4, 0);
  for (u32 c = 0; c < CH; c++) {
    u64 pw = 1;
    const u64 al = gl_canon(alphas[c]);
    for (u32 e = 0; e < QUOTIENT_TERM_POWS; e++) {
      uint32_t *w = &t[((size_t)c * QUOTIENT_TERM_POWS + e) * 4];
      w[0] = (uint32_t)(pw & 0x3FFFFF); w[1] = (uint32_t)((pw >> 22) & 0x3FFFFF); w[2] = (uint32_t)(pw >> 44);
      pw = gl_mul(pw, al);
    }
  }
  return t;
}

int upload(lcp2_ctx *ctx, DevBuf &b, const void *src, size_t bytes) {
  LCP2_HIP(ctx, b.ensure(bytes));
  if (bytes) LCP2_HIP(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return LCP2_OK;
}
// Device-to-host copies of a proof go through the context's pinned staging buffer: the pieces of one transcript step (a cap and a
// flag, all the openings, ...) are queued back to back and arrive with ONE synchronisation of the stream.
struct Download {
  lcp2_ctx *ctx;
  struct Piece { void *dst; size_t off, bytes; };
  std::vector<Piece> pieces;
  size_t used = 0;
  bool direct = false;  // a piece did not fit the staging buffer (or there is none): it went straight to its destination
  explicit Download(lcp2_ctx *c) : ctx(c) {}
  int add(void *dst, const void *src, size_t bytes) {
    if (!bytes) return LCP2_OK;
    if (!ctx->pin || used + bytes > lcp2_ctx::PIN_BYTES) {
      LCP2_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
      direct = true;
      return LCP2_OK;
    }
    LCP2_HIP(ctx, hipMemcpyAsync((char *)ctx->pin + used, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    pieces.push_back({dst, used, bytes});
    used += (bytes + 7) & ~(size_t)7;
    return LCP2_OK;
  }
  int wait() {  // the one synchronisation; the staged pieces land in their destinations
    LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (const Piece &q : pieces) memcpy(q.dst, (const char *)ctx->pin + q.off, q.bytes);
    pieces.clear();
    used = 0;
    return LCP2_OK;
  }
};
int download(lcp2_ctx *ctx, void *dst, const void *src, size_t bytes) {
  Download d(ctx);
  LCP2_TRY(d.add(dst, src, bytes));
  return d.wait();
}

// two-level extension power tables: lo[j] = z^j (j < 2^h), hi[j] = z^(j << h) (j <= count >> h), interleaved [c0, c1]
void ext_pow_tables(gl2 z, u32 h, u64 hi_count, std::vector<u64> &out, size_t &lo_off, size_t &hi_off) {
  lo_off = out.size();
  gl2 cur = gl2_make(1, 0);
  for (u64 j = 0; j < (1ull << h); j++) { out.push_back(cur.c0); out.push_back(cur.c1); cur = gl2_mul(cur, z); }
  hi_off = out.size();
  gl2 step = cur;  // z^(2^h)
  cur = gl2_make(1, 0);
  for (u64 j = 0; j < hi_count; j++) { out.push_back(cur.c0); out.push_back(cur.c1); cur = gl2_mul(cur, step); }
}
}  // namespace

// ------------------------------------------------------------------ build()
// validate the programs once so that neither the kernels nor the host verifier ever index out of range
static const char *validate_programs(const lcp2_circuit_desc *d) {
  const lcp2_params &p = d->params;
  if (d->num_regs > 64 || d->num_selectors > p.num_constants) return "bad gate set";
  const size_t nregs = std::max(d->num_regs, 1u);
  for (u32 g = 0; g < d->num_gates; g++) {
    const lcp2_gate &G = d->gates[g];
    if (G.selector_index >= d->num_selectors || ((size_t)G.code_offset + (size_t)G.code_len) * 2 > d->code_words || G.group_end < G.group_start ||
        (G.flags & ~(LCP2_GATE_EMIT_FORWARD | LCP2_GATE_NATIVE_MASK)))
      return "gate descriptor out of range";
    switch (G.flags & LCP2_GATE_NATIVE_MASK) {
      case 0: break;
      case LCP2_GATE_NATIVE_POSEIDON:
        if (!(G.flags & LCP2_GATE_EMIT_FORWARD) || G.num_constraints != 123 || p.num_wires < 135) return "LCP2_GATE_NATIVE_POSEIDON needs 135 wires, 123 forward-emitted constraints";
        break;
      case LCP2_GATE_NATIVE_ARITHMETIC:
        if ((G.flags & LCP2_GATE_EMIT_FORWARD) || G.num_constraints == 0 || 4 * (size_t)G.num_constraints > p.num_wires || p.num_constants - d->num_selectors < 2)
          return "LCP2_GATE_NATIVE_ARITHMETIC needs 4 wires per operation and 2 gate constants";
        break;
      case LCP2_GATE_NATIVE_BASE_SUM2:
        if ((G.flags & LCP2_GATE_EMIT_FORWARD) || G.num_constraints < 2 || G.num_constraints > p.num_wires) return "LCP2_GATE_NATIVE_BASE_SUM2 needs num_limbs + 1 wires";
        break;
      default:
        // a generated evaluator weights constraint j with alpha^j whichever way the program lists them (the claim check decides)
        if (!(G.flags & 0x8000u) || ((G.flags >> 8) & 0x7Fu) >= QUOTIENT_GENERATED_GATES || G.num_constraints > QUOTIENT_TERM_POWS) return "unknown native gate id";
        break;
    }
    size_t emits_seen = 0;
    for (size_t pc = G.code_offset; pc < (size_t)G.code_offset + G.code_len; pc++) {
      u32 w0 = d->code[2 * pc], w1 = d->code[2 * pc + 1];
      u32 op = w0 & 0xF, dst = (w0 >> 8) & 0xFF, kk[2] = {(w0 >> 16) & 0xF, (w0 >> 20) & 0xF}, ii[2] = {w1 & 0xFFFF, w1 >> 16};
      const bool emits = op == LCP2_OP_EMIT || op == LCP2_OP_EMITBOOL;
      if (op > LCP2_OP_PMDS) return "bad instruction";
      if (op == LCP2_OP_PMDS) {  // register windows of 12 and a block of 12 immediates
        if (kk[0] != 0 || kk[1] != 3 || (size_t)dst + 12 > nregs || (size_t)ii[0] + 12 > nregs || (size_t)ii[1] + 12 > d->num_imm) return "PMDS window out of range";
        continue;
      }
      if (!emits && dst >= nregs) return "bad instruction";
      emits_seen += emits;
      const int nsrc = (emits || op == LCP2_OP_SBOX) ? 1 : 2;
      for (int k = 0; k < nsrc; k++) {
        size_t lim = kk[k] == 0 ? nregs : kk[k] == 1 ? p.num_wires : kk[k] == 2 ? p.num_constants - d->num_selectors
                     : kk[k] == 3 ? d->num_imm : kk[k] == 4 ? 4 : 0;
        if (ii[k] >= lim) return "operand out of range";
      }
    }
    if (emits_seen != G.num_constraints) return "num_constraints does not match the program";
  }
  return nullptr;
}

// cap of oracle `o` into a full-size cap buffer: a sharded circuit writes its own entries at their global position and
// zeros elsewhere (its share: the caps of all ranks OR-ed together are the cap)
static int queue_cap(Download &d, lcp2_circuit *c, const lcp2_oracle &o, u64 *dst) {
  const size_t capw = (size_t)4 << c->p.cap_height;
  if (!c->sharded()) return d.add(dst, o.cap_dev(), capw * 8);
  const size_t per_block = (size_t)4 << (c->p.cap_height - c->p.rate_bits);
  memset(dst, 0, capw * 8);
  return d.add(dst + c->bf * per_block, o.cap_dev(), per_block * c->bc * 8);
}
static int download_cap(lcp2_circuit *c, const lcp2_oracle &o, u64 *dst) {
  Download d(c->ctx);
  LCP2_TRY(queue_cap(d, c, o, dst));
  return d.wait();
}

// circuit_builder.rs::build: circuit_digest = hash_no_pad(constants_sigmas_cap || domain_separator_digest || degree_bits) with
// domain_separator_digest = hash_pad(domain separator), the separator empty unless the builder sets one: pad10*1 = [1, 0 x 6, 1]
static void circuit_digest(const std::vector<u64> &cs_cap, u32 degree_bits, u64 digest[4]) {
  std::vector<u64> buf(cs_cap);
  const u64 empty_padded[8] = {1, 0, 0, 0, 0, 0, 0, 1};
  u64 ds[4];
  HostPoseidon::get().hash_no_pad(empty_padded, 8, ds);
  buf.insert(buf.end(), ds, ds + 4);
  buf.push_back(degree_bits);
  HostPoseidon::get().hash_no_pad(buf.data(), buf.size(), digest);
}

// The LCP2_GATE_NATIVE_* claims of the description, checked on the device: program and native evaluator on 256 random points
// (a polynomial identity in 135 + NC variables of degree <= 9: a wrong claim survives with probability ~2^-60).
static int check_native_gates(lcp2_circuit *c) {
  lcp2_ctx *ctx = c->ctx;
  bool any = false;
  for (const lcp2_gate &G : c->gates) any = any || (G.flags & LCP2_GATE_NATIVE_MASK);
  if (!any) return LCP2_OK;
  const lcp2_params &p = c->p;
  const u64 cnt = 256;
  u64 seed = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { seed += 0x9E3779B97F4A7C15ull; u64 z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return gl_canon(z ^ (z >> 31)); };
  std::vector<u64> hw((size_t)p.num_wires * cnt), hc((size_t)p.num_constants * cnt), hs(SMALL_GATE_SCALE + (size_t)QUOTIENT_MAX_CH * c->gates.size(), 0);
  for (auto &v : hw) v = rnd();
  for (auto &v : hc) v = rnd();
  for (u32 i = 0; i < 4; i++) hs[SMALL_PI_HASH + i] = rnd();
  DevBuf dw, dc;
  LCP2_TRY(upload(ctx, dw, hw.data(), hw.size() * 8));
  LCP2_TRY(upload(ctx, dc, hc.data(), hc.size() * 8));
  u64 bad = ~0ull;
  // two settings of the challenges: random ones, and alpha = 0 (there the combination is the FIRST constraint alone, the corner in
  // which a forward and a last-to-first evaluator differ if one of them folds in the wrong direction)
  for (int zero_alpha = 0; zero_alpha < 2 && bad == ~0ull; zero_alpha++) {
    for (u32 k = 0; k < p.num_challenges; k++) {
      const u64 al = zero_alpha ? 0 : gl_canon(rnd() | 1);
      hs[SMALL_ALPHAS + k] = al;
      hs[SMALL_ALPHA_INV + k] = al ? gl_inv(al) : 0;
      for (size_t g = 0; g < c->gates.size(); g++)
        hs[SMALL_GATE_SCALE + g * QUOTIENT_MAX_CH + k] = c->gates[g].num_constraints ? gl_pow(al, c->gates[g].num_constraints - 1) : 1;
    }
    hs[SMALL_CHECK] = ~0ull;
    LCP2_HIP(ctx, hipMemcpyAsync(c->small.p, hs.data(), hs.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    const std::vector<uint32_t> limbs = alpha_limb_table(&hs[SMALL_ALPHAS], p.num_challenges);
    LCP2_TRY(upload(ctx, c->alpha_limbs, limbs.data(), limbs.size() * 4));
    QuotientArgs a{};
    u64 *d_small = c->small.u();
    a.wires = dw.u(); a.consts = dc.u(); a.stride = cnt; a.count = cnt;
    a.alphas = d_small + SMALL_ALPHAS; a.alpha_inv = d_small + SMALL_ALPHA_INV; a.pis = d_small + SMALL_PI_HASH; a.gate_scale = d_small + SMALL_GATE_SCALE;
    a.alpha_limbs = (const u32 *)c->alpha_limbs.p;
    a.imm = c->d_imm.u(); a.code = (const u32 *)c->d_code.p; a.gates = (const GateDev *)c->d_gates.p; a.stage_list = (const u32 *)c->d_stage.p;
    a.num_wires = p.num_wires; a.num_gates = (u32)c->gates.size(); a.num_selectors = c->num_selectors; a.num_constants = p.num_constants;
    a.num_challenges = p.num_challenges; a.num_regs = c->num_regs; a.rc = ctx->d_rc;
    launch_native_check(ctx->stream, a, c->dev_gates, (unsigned long long *)(d_small + SMALL_CHECK));
    LCP2_HIP(ctx, hipGetLastError());
    LCP2_TRY(download(ctx, &bad, d_small + SMALL_CHECK, 8));
  }
  if (bad != ~0ull) return ctx->fail(LCP2_E_INVALID, "a gate flagged LCP2_GATE_NATIVE_* does not compute what its program computes");
  return LCP2_OK;
}

static int circuit_create(lcp2_ctx *ctx, const lcp2_circuit_desc *d, uint32_t bf, uint32_t bc, lcp2_circuit **out) {
  if (!ctx || !d || !out) return LCP2_E_INVALID;
  *out = nullptr;
  if (!d->constants_sigmas || !d->k_is || !d->gates || !d->code || (d->num_imm && !d->imm)) return ctx->fail(LCP2_E_INVALID, "null description field");
  LCP2_TRY(check_params(ctx, d->params));
  const lcp2_params &p = d->params;
  if (d->num_selectors > p.num_constants || d->num_regs > 64 || d->num_gates == 0) return ctx->fail(LCP2_E_INVALID, "bad gate set");
  if (d->num_public_inputs > (1u << 20)) return ctx->fail(LCP2_E_UNSUPPORTED, "too many public inputs");
  if (const char *why = validate_programs(d)) return ctx->fail(LCP2_E_INVALID, why);
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<lcp2_circuit> c(new lcp2_circuit());
  c->ctx = ctx; c->p = p; c->npi = d->num_public_inputs; c->num_selectors = d->num_selectors; c->num_regs = std::max(d->num_regs, 1u);
  if (bc) {
    if (p.cap_height < p.rate_bits) return ctx->fail(LCP2_E_INVALID, "sharded circuit: needs cap_height >= rate_bits");
    if ((bc & (bc - 1)) || bf % bc || bf + bc > (1u << p.rate_bits)) return ctx->fail(LCP2_E_INVALID, "sharded circuit: block range must be an aligned power of two");
    c->bf = bf; c->bc = bc; c->cap_final = false;
    for (lcp2_oracle *o : {&c->cs, &c->wires, &c->zs, &c->quot}) { o->block_first = bf; o->block_count = bc; }
  }
  c->gates.assign(d->gates, d->gates + d->num_gates);
  c->code.assign(d->code, d->code + d->code_words);
  c->imm.resize(std::max<size_t>(d->num_imm, 1), 0);
  for (size_t i = 0; i < d->num_imm; i++) c->imm[i] = gl_canon(d->imm[i]);
  c->k_is.resize(p.num_routed_wires);
  for (u32 i = 0; i < p.num_routed_wires; i++) c->k_is[i] = gl_canon(d->k_is[i]);
  const u64 n = 1ull << p.degree_bits, N = n << p.rate_bits;
  const u32 ncs = p.num_constants + p.num_routed_wires, CH = p.num_challenges, npp = npp_of(p), nchunks = npp + 1;
  {  // the device runs the staged form of the programs (prover_kernels.hpp); the verifier keeps the caller's form
    static_assert(sizeof(GateDev) == sizeof(lcp2_gate), "GateDev mirrors lcp2_gate");
    std::vector<GateDev> &dev_gates = c->dev_gates;
    dev_gates.resize(c->gates.size());
    memcpy(dev_gates.data(), c->gates.data(), c->gates.size() * sizeof(lcp2_gate));
    std::vector<uint32_t> staged, lists;
    stage_gate_programs(c->code, dev_gates, p.num_wires, c->num_selectors, staged, lists);
    // LDS registers the DEVICE needs: programs that run natively never touch them (their count is only a verifier matter)
    c->dev_regs = 1;
    for (const lcp2_gate &G : c->gates) {
      if (G.flags & LCP2_GATE_NATIVE_MASK) continue;
      for (size_t pc = G.code_offset; pc < (size_t)G.code_offset + G.code_len; pc++) {
        const u32 w0 = c->code[2 * pc], w1 = c->code[2 * pc + 1], op = w0 & 0xF, dst = (w0 >> 8) & 0xFF;
        const bool emits = op == LCP2_OP_EMIT || op == LCP2_OP_EMITBOOL;
        u32 top = emits ? 0 : dst + (op == LCP2_OP_PMDS ? 12 : 1);
        if (((w0 >> 16) & 0xF) == 0 && op != LCP2_OP_PMDS) top = std::max(top, (w1 & 0xFFFF) + 1);
        if (op == LCP2_OP_PMDS) top = std::max(top, (w1 & 0xFFFF) + 12);
        if (!emits && op != LCP2_OP_SBOX && op != LCP2_OP_PMDS && ((w0 >> 20) & 0xF) == 0) top = std::max(top, (w1 >> 16) + 1);
        c->dev_regs = std::max(c->dev_regs, top);
      }
    }
    staged.resize(staged.size() + 4, 0);  // padded by two instructions: K6 fetches one instruction ahead of the one it executes
    LCP2_TRY(upload(ctx, c->d_gates, dev_gates.data(), dev_gates.size() * sizeof(GateDev)));
    LCP2_TRY(upload(ctx, c->d_code, staged.data(), staged.size() * 4));
    LCP2_TRY(upload(ctx, c->d_stage, lists.data(), lists.size() * 4));
    LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the staging vectors go out of scope
  }
  LCP2_TRY(upload(ctx, c->d_imm, c->imm.data(), c->imm.size() * 8));
  LCP2_TRY(upload(ctx, c->d_kis, c->k_is.data(), c->k_is.size() * 8));
  // constants_sigmas values stay resident (K5 reads the sigma columns on H)
  LCP2_HIP(ctx, c->cs_values.alloc((size_t)ncs * n * 8));
  LCP2_HIP(ctx, hipMemcpyAsync(c->cs_values.p, d->constants_sigmas, (size_t)ncs * n * 8,
                               d->constants_sigmas_mem == LCP2_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  LCP2_TRY(commit_values_dev(ctx, c->cs_values.u(), ncs, p.degree_bits, p.rate_bits, p.cap_height, &c->cs));
  const size_t capw = (size_t)4 << p.cap_height;
  c->cs_cap.resize(capw);
  LCP2_TRY(download_cap(c.get(), c->cs, c->cs_cap.data()));
  if (!c->sharded()) circuit_digest(c->cs_cap, p.degree_bits, c->digest);  // sharded: lcp2_circuit_set_constants_cap
  // L_0 on the LDE points (leaf order): LDE of the polynomial with all coefficients 1/n
  {
    DevBuf ones;
    LCP2_HIP(ctx, ones.alloc(n * 8));
    launch_fill(ctx->stream, ones.u(), n, gl_inv(n % GL_P));
    LCP2_HIP(ctx, c->d_l0.alloc(N * 8));
    DeviceNttBackend be{ctx};
    NttHost<DeviceNttBackend> ntt(be);
    ntt.forward(ones.u(), n, c->d_l0.u(), N, p.degree_bits, 1, GL_GENERATOR, p.rate_bits);
    if (be.status) return be.status;
    LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  {  // 1 / Z_H(7 w_N^j) depends on j mod 2^rate_bits = bitrev of the top rate_bits of the leaf index
    std::vector<u64> t(1u << p.rate_bits);
    u64 shift_n = gl_pow(GL_GENERATOR, n), wr = gl_root_of_unity(p.rate_bits);
    for (u32 top = 0; top < (1u << p.rate_bits); top++) {
      u32 r = bitrev32(top, p.rate_bits);
      t[top] = gl_inv(gl_sub(gl_mul(shift_n, gl_pow(wr, r)), 1));
    }
    LCP2_TRY(upload(ctx, c->d_zh_inv, t.data(), t.size() * 8));
  }
  // per-proof workspace
  LCP2_HIP(ctx, c->zs_vals.alloc((size_t)CH * (1 + npp) * n * 8));
  LCP2_HIP(ctx, c->chunk_q.alloc((size_t)CH * nchunks * n * 8));
  LCP2_HIP(ctx, c->row_tot.alloc((size_t)CH * n * 8));
  LCP2_HIP(ctx, c->scan_tmp.alloc(std::max(scan_scratch_words(n, 4), (u64)16) * 8));
  LCP2_HIP(ctx, c->qvals.alloc((size_t)CH * N * 8));
  LCP2_HIP(ctx, c->planes.alloc((size_t)4 * n * 8));
  LCP2_HIP(ctx, c->small.alloc((SMALL_GATE_SCALE + (size_t)QUOTIENT_MAX_CH * d->num_gates + 8) * 8));
  {
    u32 maxcols = std::max(std::max(ncs, p.num_wires), std::max(CH * (1 + npp), CH * p.quotient_degree_factor));
    u64 nchk = (n + EVAL_CHUNK - 1) / EVAL_CHUNK;
    LCP2_HIP(ctx, c->partial.alloc((size_t)maxcols * nchk * 16 + (size_t)maxcols * 16));
  }
  LCP2_HIP(ctx, c->tables.alloc(((size_t)8 * ((1ull << ((p.degree_bits + 1) / 2)) + (n >> ((p.degree_bits + 1) / 2)) + 2) + 4 * 1024 + 2 * (ncs + p.num_wires + 64) + 64) * 16));
  LCP2_HIP(ctx, c->fri_c[0].alloc((size_t)2 * n * 8));
  LCP2_HIP(ctx, c->fri_c[1].alloc((size_t)2 * n * 8));
  {
    u64 m = n;
    c->fri_vals.resize(p.num_fri_layers); c->fri_dig.resize(p.num_fri_layers);
    c->fri_level_off.resize(p.num_fri_layers); c->fri_d_level_off.resize(p.num_fri_layers);
    for (u32 l = 0; l < p.num_fri_layers; l++) {
      u64 nvals = m << p.rate_bits, nleaves = nvals >> p.fri_arity_bits[l];
      u32 h = 0;
      while ((1ull << h) < nleaves) h++;
      u32 nlev = h - p.cap_height + 1;
      if (l == 0 && c->sharded()) {  // its own leaf blocks only, down to its own cap entries: the same number of levels
        nvals = (u64)c->bc * m;
        nleaves = nvals >> p.fri_arity_bits[0];
      }
      LCP2_HIP(ctx, c->fri_vals[l].alloc((size_t)2 * nvals * 8));
      c->fri_level_off[l].resize(nlev);
      u64 tot = 0;
      for (u32 k = 0; k < nlev; k++) { c->fri_level_off[l][k] = tot; tot += nleaves >> k; }
      LCP2_HIP(ctx, c->fri_dig[l].alloc(tot * 32));
      LCP2_TRY(upload(ctx, c->fri_d_level_off[l], c->fri_level_off[l].data(), nlev * 8));
      m >>= p.fri_arity_bits[l];
    }
  }
  LCP2_HIP(ctx, c->q_idx.alloc(64 * 8 * (3 + LCP2_MAX_FRI_LAYERS)));
  LCP2_HIP(ctx, c->q_buf.alloc((size_t)64 * (ncs + p.num_wires + CH * (1 + npp) + CH * p.quotient_degree_factor + 4 * 4 * 32 + LCP2_MAX_FRI_LAYERS * (64 + 4 * 32)) * 8));
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LCP2_TRY(check_native_gates(c.get()));
  *out = c.release();
  return LCP2_OK;
}

extern "C" int lcp2_circuit_create(lcp2_ctx *ctx, const lcp2_circuit_desc *d, lcp2_circuit **out) { return circuit_create(ctx, d, 0, 0, out); }
extern "C" int lcp2_circuit_create_sharded(lcp2_ctx *ctx, const lcp2_circuit_desc *d, uint32_t block_first, uint32_t block_count,
                                           lcp2_circuit **out) {
  if (block_count == 0) return LCP2_E_INVALID;
  return circuit_create(ctx, d, block_first, block_count, out);
}
extern "C" int lcp2_circuit_set_constants_cap(lcp2_circuit *c, const uint64_t *cap) {
  if (!c || !cap) return LCP2_E_INVALID;
  if (!c->sharded()) return LCP2_E_INVALID;
  const size_t capw = (size_t)4 << c->p.cap_height, per_block = (size_t)4 << (c->p.cap_height - c->p.rate_bits);
  // the entries this handle computed itself must be in the cap it is given
  if (memcmp(cap + c->bf * per_block, c->cs_cap.data() + c->bf * per_block, per_block * c->bc * 8) != 0)
    return c->ctx->fail(LCP2_E_INVALID, "constants cap does not contain this shard's entries");
  c->cs_cap.assign((const u64 *)cap, (const u64 *)cap + capw);
  circuit_digest(c->cs_cap, c->p.degree_bits, c->digest);
  c->cap_final = true;
  return LCP2_OK;
}

extern "C" int lcp2_verifier_create(const lcp2_circuit_desc *d, const uint64_t digest[4], const uint64_t *cap, lcp2_circuit **out) {
  if (!d || !digest || !cap || !out || !d->k_is || !d->gates || !d->code) return LCP2_E_INVALID;
  *out = nullptr;
  const lcp2_params &p = d->params;
  bool unsupported;
  if (params_problem(p, &unsupported)) return unsupported ? LCP2_E_UNSUPPORTED : LCP2_E_INVALID;  // the verifier indexes fixed-size arrays by these
  if (d->num_selectors > p.num_constants || d->num_gates == 0 || (d->num_imm && !d->imm)) return LCP2_E_INVALID;
  if (validate_programs(d)) return LCP2_E_INVALID;
  lcp2_circuit *c = new lcp2_circuit();
  c->p = p; c->npi = d->num_public_inputs; c->num_selectors = d->num_selectors; c->num_regs = std::max(d->num_regs, 1u);
  c->gates.assign(d->gates, d->gates + d->num_gates);
  c->code.assign(d->code, d->code + d->code_words);
  c->imm.resize(std::max<size_t>(d->num_imm, 1), 0);
  for (size_t i = 0; i < d->num_imm; i++) c->imm[i] = gl_canon(d->imm[i]);
  c->k_is.resize(p.num_routed_wires);
  for (u32 i = 0; i < p.num_routed_wires; i++) c->k_is[i] = gl_canon(d->k_is[i]);
  memcpy(c->digest, digest, 32);
  c->cs_cap.assign((const u64 *)cap, (const u64 *)cap + ((size_t)4 << p.cap_height));
  *out = c;
  return LCP2_OK;
}

extern "C" void lcp2_circuit_destroy(lcp2_circuit *c) {
  if (!c) return;
  if (c->ctx) { (void)hipSetDevice(c->ctx->device); (void)hipStreamSynchronize(c->ctx->stream); }
  delete c;
}
extern "C" int lcp2_circuit_digest(const lcp2_circuit *c, uint64_t digest[4], uint64_t *cap) {
  if (!c || !digest) return LCP2_E_INVALID;
  memcpy(digest, c->digest, 32);
  if (cap) memcpy(cap, c->cs_cap.data(), c->cs_cap.size() * 8);
  return LCP2_OK;
}
extern "C" size_t lcp2_proof_words(const lcp2_params *p) {
  bool unsupported;
  if (!p || params_problem(*p, &unsupported)) return 0;  // the layout arithmetic relies on a sane FRI schedule
  return ProofLayout(*p).total;
}
extern "C" int lcp2_last_challenges(const lcp2_circuit *c, uint64_t out[97]) {
  if (!c || !out) return LCP2_E_INVALID;
  memcpy(out, c->last_challenges, sizeof c->last_challenges);
  return LCP2_OK;
}

// host-side accessors for the verifier (verifier.hip)
namespace lcp2 {
VerifierView verifier_view(const lcp2_circuit *c) {
  VerifierView v;
  v.p = &c->p; v.npi = c->npi; v.num_selectors = c->num_selectors;
  v.gates = c->gates.data(); v.num_gates = (u32)c->gates.size(); v.code = c->code.data(); v.imm = c->imm.data();
  v.k_is = c->k_is.data(); v.digest = c->digest; v.cs_cap = c->cs_cap.data();
  return v;
}
}  // namespace lcp2

// ------------------------------------------------------------------ prove()
namespace {
int open_oracle(lcp2_ctx *ctx, lcp2_oracle &o, const u64 *d_idx, u32 k, u64 *d_leaves, u64 *d_sib) {
  launch_gather_rows(ctx->stream, o.lde.u(), o.nleaves(), o.ncols, d_idx, k, d_leaves);
  launch_gather_digests(ctx->stream, o.digests.u(), (const u64 *)o.d_level_off.p, o.nlevels() - 1, d_idx, k, d_sib);
  return LCP2_OK;
}

// evaluate `ncols` coefficient columns at z; results (ext) land in d_out[2 * ncols].  d_tab: the power tables of z (eval_tables)
u32 eval_chunk_len(u64 n) { return (u32)std::min<u64>(n, EVAL_CHUNK); }
size_t eval_table_words(u64 n) { return 512 + 2 * (size_t)(n / eval_chunk_len(n)); }
void eval_tables(lcp2_circuit *c, gl2 z, u64 *d_tab) {  // z travels in the kernel arguments: no staging copy, no synchronisation
  const u64 n = 1ull << c->p.degree_bits;
  launch_eval_tables(c->ctx->stream, z.c0, z.c1, eval_chunk_len(n), (u32)(n / eval_chunk_len(n)), d_tab);
}
void eval_columns(lcp2_circuit *c, const u64 *coeffs, u32 ncols, gl2 z, u64 *d_out, const u64 *d_tab) {
  const u64 n = 1ull << c->p.degree_bits;
  EvalArgs a{};
  a.coeffs = coeffs; a.col_stride = n;
  a.chunk_len = eval_chunk_len(n);
  a.items = (a.chunk_len + 255) / 256;
  a.nchunks = (u32)(n / a.chunk_len);
  const gl2 zs = gl2_pow(z, 256);
  a.zstep[0] = zs.c0; a.zstep[1] = zs.c1;
  a.zpow_t = d_tab; a.zpow_chunk = d_tab + 512;
  a.partial = c->partial.u();
  launch_eval_polys(c->ctx->stream, a, ncols, d_out);
}
}  // namespace

// ---- the four seams of SURVEY section 8b: each is a function of its inputs and of the commitments made by the
// stages before it (held by the circuit handle); lcp2_prove is their composition under the Fiat-Shamir transcript.
namespace {
#define LCP2_STAGE_PROLOGUE \
    lcp2_ctx *ctx = c->ctx; \
    LCP2_HIP(ctx, hipSetDevice(ctx->device)); \
    const lcp2_params &p = c->p; \
    const u64 n = 1ull << p.degree_bits, N = n << p.rate_bits; \
    const u32 lgN = p.degree_bits + p.rate_bits, W = p.num_wires, NR = p.num_routed_wires, NC = p.num_constants, CH = p.num_challenges, \
              Q = p.quotient_degree_factor, npp = npp_of(p), nchunks = npp + 1, ncs = NC + NR; \
    const ProofLayout L(p); \
    hipStream_t s = ctx->stream; \
    DeviceNttBackend be{ctx}; \
    NttHost<DeviceNttBackend> ntt(be); \
    (void)N; (void)lgN; (void)W; (void)NR; (void)NC; (void)CH; (void)Q; (void)npp; (void)nchunks; (void)ncs; (void)L; (void)s; (void)ntt;

// PolynomialBatch::from_values on the witness (K1-K4).  d_coeffs (nullable, device): the coefficients of every wire column,
// already computed (a sharded proof runs the iNTT polynomial-parallel across the ranks and all-gathers the result).
// rows_only: `wires_in` is this rank's row block of the values, [num_wires][n / world] (device), see lcp2_commit_wires_rows.
int stage_wires(lcp2_circuit *c, const u64 *wires_in, lcp2_mem wires_mem, const u64 *d_coeffs, u64 *cap_out, bool rows_only = false) {
  LCP2_STAGE_PROLOGUE
  if (rows_only && (!c->sharded() || !d_coeffs || wires_mem != LCP2_MEM_DEVICE || n < c->world()))
    return ctx->fail(LCP2_E_INVALID, "lcp2_commit_wires_rows: needs a sharded circuit with at least one row per rank, device buffers");
  c->rows_mode = rows_only;
  c->perm_phase = 0;
  c->fo.phase = 0;  // a new proof: an opening stage left half-way belongs to the previous one
  const u64 *d_wires = wires_in;
  if (wires_mem == LCP2_MEM_HOST) {
    LCP2_HIP(ctx, c->wires_vals.ensure((size_t)W * n * 8));
    LCP2_HIP(ctx, hipMemcpyAsync(c->wires_vals.p, wires_in, (size_t)W * n * 8, hipMemcpyHostToDevice, s));
    d_wires = c->wires_vals.u();
  }
  c->stage = lcp2_circuit::ST_NONE;
  // The caller's buffer may hold non-canonical values (any u64): the transforms and K5 canonicalise what they load, the witness
  // check of the quotient stage does not.  The bit-reversal of the iNTT, which reads every value anyway, reports whether one is
  // >= p (no extra traffic); stage_perm_zs then takes a canonical copy before anything reads the values again.
  unsigned long long *d_flag = (unsigned long long *)(c->small.u() + SMALL_NONCANON);
  launch_set_words(s, c->small.u() + SMALL_NONCANON, SmallWords{}, 1);
  if (d_coeffs) {
    LCP2_TRY(commit_coeffs_dev(ctx, d_coeffs, W, p.degree_bits, p.rate_bits, p.cap_height, &c->wires, true));
    launch_canon_copy(s, d_wires, nullptr, (u64)W * c->rows(), d_flag);  // the values did not pass through an iNTT here: scan them
    if (rows_only && !c->cs_rows_ready) {  // the gate check reads the constants with the stride of the wires
      const u64 R = c->rows();
      LCP2_HIP(ctx, c->cs_rows.ensure((size_t)NC * R * 8));
      launch_copy_2d(s, c->cs_rows.u(), R, c->cs_values.u() + c->row0(), n, R, NC);
      c->cs_rows_ready = true;
    }
  } else {
    LCP2_TRY(commit_values_dev(ctx, d_wires, W, p.degree_bits, p.rate_bits, p.cap_height, &c->wires, d_flag));
  }
  {  // the cap and the non-canonical flag with one synchronisation
    Download d(ctx);
    LCP2_TRY(queue_cap(d, c, c->wires, cap_out));
    LCP2_TRY(d.add(&c->noncanon_host, d_flag, 8));
    LCP2_TRY(d.wait());
  }
  c->d_wires_cur = d_wires;
  c->stage = lcp2_circuit::ST_WIRES;
  return LCP2_OK;
}

// wires_permutation_partial_products_and_zs + commitment (K5, K1-K4), in three steps so that a sharded proof in the row
// exchange form can run K5 on its own rows: perm_begin (chunk products and their running product inside the block),
// perm_finish (Z and the partial products, times the product of the blocks before this one), perm_commit.
PermArgs perm_args(lcp2_circuit *c, NttHost<DeviceNttBackend> &ntt, u64 *zs_out) {
  const lcp2_params &p = c->p;
  const u64 n = 1ull << p.degree_bits, R = c->rows();
  u64 *d_small = c->small.u();
  PermArgs a{};
  a.wires = c->d_wires_cur; a.wires_stride = R;
  a.sigmas = c->cs_values.u() + (u64)p.num_constants * n + c->row0(); a.sigma_stride = n;
  a.k_is = c->d_kis.u();
  a.subgroup = ntt.root_table(p.degree_bits, false);
  a.betas = d_small + SMALL_BETAS; a.gammas = d_small + SMALL_GAMMAS; a.prefix = nullptr;
  a.chunk_q = c->chunk_q.u(); a.row_tot = c->row_tot.u(); a.zs_out = zs_out;
  a.n = R; a.row0 = c->row0();
  a.num_routed = p.num_routed_wires; a.chunk = p.quotient_degree_factor; a.nchunks = npp_of(p) + 1; a.num_challenges = p.num_challenges;
  return a;
}
// where K5 writes: the value buffer of the commitment, or this rank's slot of the exchange buffer
u64 *perm_out(lcp2_circuit *c) {
  const u64 ncz = (u64)c->p.num_challenges * (1 + npp_of(c->p));
  return c->rows_mode ? c->zs_rows.u() + (u64)c->rank() * ncz * c->rows() : c->zs_vals.u();
}

int perm_begin(lcp2_circuit *c, const u64 *betas, const u64 *gammas) {
  LCP2_STAGE_PROLOGUE
  if (c->stage < lcp2_circuit::ST_WIRES) return ctx->fail(LCP2_E_INVALID, "lcp2_perm_zs: the wires are not committed");
  const u64 R = c->rows();
  u64 *d_small = c->small.u();
  {  // the challenges travel in the kernel arguments (betas at SMALL_BETAS, gammas right behind them)
    static_assert(SMALL_GAMMAS == SMALL_BETAS + 4 && QUOTIENT_MAX_CH <= 4, "betas and gammas are set with one launch");
    SmallWords w{};
    for (u32 k = 0; k < CH; k++) { w.v[k] = gl_canon(betas[k]); w.v[4 + k] = gl_canon(gammas[k]); }
    launch_set_words(s, d_small + SMALL_BETAS, w, 8);
  }
  if (c->noncanon_host) {  // rare: a witness with values in [p, 2^64): continue from a canonical copy (stage_wires)
    LCP2_HIP(ctx, c->wires_vals.ensure((size_t)W * R * 8));
    launch_canon_copy(s, c->d_wires_cur, c->wires_vals.u(), (u64)W * R, nullptr);  // (a host witness is already the library's copy: in place)
    c->d_wires_cur = c->wires_vals.u();
  }
  if (c->rows_mode) LCP2_HIP(ctx, c->zs_rows.ensure((size_t)CH * (1 + npp) * n * 8));
  // ---- K5: the quotient chunks of every row and Z inside the block (exclusive prefix product of the row totals)
  u64 *zs_out = perm_out(c);
  PermArgs a = perm_args(c, ntt, zs_out);
  if (be.status) return be.status;
  {
    ProfScope ps(ctx, LCP2_K_PERM_Z, (double)R * 8.0 * (2.0 * NR + CH * (1.0 + npp)));
    launch_perm_chunks(s, a);
    launch_scan(s, true, c->row_tot.u(), zs_out, c->scan_tmp.u(), R, false, CH, R);
  }
  LCP2_HIP(ctx, hipGetLastError());
  return LCP2_OK;
}
// Z before the block's last row and that row's quotient, per challenge, into c->perm_wrap: queued behind whatever the caller
// downloads next (perm_finalize rescales zs_out in place only when a prefix is given, and then the caller has read these first)
int queue_perm_wrap(Download &d, lcp2_circuit *c) {
  const u64 R = c->rows();
  const u64 *zs_out = perm_out(c);
  for (u32 k = 0; k < c->p.num_challenges; k++) {
    LCP2_TRY(d.add(&c->perm_wrap[2 * k], zs_out + (u64)k * R + (R - 1), 8));
    LCP2_TRY(d.add(&c->perm_wrap[2 * k + 1], c->row_tot.u() + (u64)k * R + (R - 1), 8));
  }
  return LCP2_OK;
}

// prefix (nullable, host, [CH]): the product of the row blocks before this one
int perm_finish(lcp2_circuit *c, const u64 *prefix) {
  LCP2_STAGE_PROLOGUE
  PermArgs a = perm_args(c, ntt, perm_out(c));
  if (be.status) return be.status;
  if (prefix) {
    SmallWords w{};
    for (u32 k = 0; k < CH; k++) w.v[k] = prefix[k];
    launch_set_words(s, c->small.u() + SMALL_PERM_PREFIX, w, CH);
    a.prefix = c->small.u() + SMALL_PERM_PREFIX;
  }
  ProfScope ps(ctx, LCP2_K_PERM_Z, (double)c->rows() * 8.0 * CH * (1.0 + 2.0 * npp));
  launch_perm_finalize(s, a);
  LCP2_HIP(ctx, hipGetLastError());
  return LCP2_OK;
}

int perm_commit(lcp2_circuit *c, u64 *cap_out, bool with_wrap = false) {
  LCP2_STAGE_PROLOGUE
  const u32 ncz = CH * (1 + npp);
  if (c->rows_mode) {  // the exchange buffer holds every rank's rows, [rank][column][rows]: back to whole columns
    const u64 R = c->rows();
    for (u32 r = 0; r < c->world(); r++)
      launch_copy_2d(s, c->zs_vals.u() + (u64)r * R, n, c->zs_rows.u() + (u64)r * ncz * R, R, R, ncz);
  }
  LCP2_TRY(commit_values_dev(ctx, c->zs_vals.u(), ncz, p.degree_bits, p.rate_bits, p.cap_height, &c->zs));
  Download d(ctx);
  LCP2_TRY(queue_cap(d, c, c->zs, cap_out));
  if (with_wrap) LCP2_TRY(queue_perm_wrap(d, c));  // (the commitment reads zs_vals, it does not change it)
  LCP2_TRY(d.wait());
  c->stage = lcp2_circuit::ST_ZS;
  return LCP2_OK;
}

int stage_perm_zs(lcp2_circuit *c, const u64 *betas, const u64 *gammas, u64 *cap_out) {
  if (c->rows_mode) return c->ctx->fail(LCP2_E_INVALID, "row exchange form: lcp2_perm_zs_rows_begin / _finish / lcp2_perm_zs_commit");
  LCP2_TRY(perm_begin(c, betas, gammas));
  LCP2_TRY(perm_finish(c, nullptr));
  LCP2_TRY(perm_commit(c, cap_out, true));  // one synchronisation: the cap and perm_wrap
  // Copy constraints: Z must come back to 1 after the last row, Z(g^(n-1)) * (row n-1's quotient) = 1, which holds for
  // every beta, gamma exactly when the wire values are constant on the cycles of sigma (up to the soundness error of the
  // argument itself).  plonky2 reports a broken copy constraint as an Err of prove(); so does this (LCP2_E_UNSAT).
  for (u32 k = 0; k < c->p.num_challenges; k++)
    if (gl_mul(c->perm_wrap[2 * k], c->perm_wrap[2 * k + 1]) != 1) {
      c->stage = lcp2_circuit::ST_WIRES;
      return c->ctx->fail(LCP2_E_UNSAT, "the witness violates a copy constraint (the permutation product does not return to 1)");
    }
  return LCP2_OK;
}

// compute_quotient_polys + commitment (K6, K1-K4)
// defer_check: leave the gate-check verdict on the device; stage_quotient_commit reads it together with the quotient cap
int stage_quotient_values(lcp2_circuit *c, const u64 *alphas, const u64 *pi_hash, bool defer_check = false) {
  LCP2_STAGE_PROLOGUE
  if (c->stage < lcp2_circuit::ST_ZS) return ctx->fail(LCP2_E_INVALID, "lcp2_quotient: Z / partial products are not committed");
  u64 *d_small = c->small.u();
  u64 *d_betas = d_small + SMALL_BETAS, *d_gammas = d_small + SMALL_GAMMAS, *d_alphas = d_small + SMALL_ALPHAS;
  const u32 NG = (u32)c->gates.size();
  {  // alphas, their inverses and powers, the limb table, the public-input hash, the check flag, alpha^(m_g - 1) per gate: computed
     // on the device from the challenges in the kernel arguments (k_quotient_setup)
    QuotientSetupArgs qs{};
    for (u32 k = 0; k < CH; k++) qs.alphas[k] = gl_canon(alphas[k]);
    for (u32 i = 0; i < 4; i++) qs.pi_hash[i] = gl_canon(pi_hash[i]);
    qs.num_challenges = CH; qs.num_gates = NG; qs.gates = (const GateDev *)c->d_gates.p; qs.small = d_small;
    LCP2_HIP(ctx, c->alpha_limbs.ensure((size_t)QUOTIENT_MAX_CH * QUOTIENT_TERM_POWS * 16));
    qs.limbs = (u32 *)c->alpha_limbs.p;
    launch_quotient_setup(s, qs);
  }
  // ---- K6: quotient values on the coset, coset iNTT, chunking, commitment
  {
    QuotientArgs a{};
    a.wires = c->wires.lde.u(); a.consts = c->cs.lde.u(); a.zs = c->zs.lde.u(); a.l0 = c->d_l0.u(); a.zh_inv = c->d_zh_inv.u();
    u64 ls, hs;
    a.points = ntt.shift_table(gl_root_of_unity(lgN), lgN, 0, false, GL_GENERATOR, ls, hs);
    a.k_is = c->d_kis.u(); a.betas = d_betas; a.gammas = d_gammas; a.alphas = d_alphas; a.pis = d_small + SMALL_PI_HASH; a.imm = c->d_imm.u();
    a.kis_pow7 = 1;
    for (u32 j = 0; j < NR; j++) a.kis_pow7 &= c->k_is[j] == (j ? gl_mul(c->k_is[j - 1], 7) : 1);  // plonky2's coset shifts
    a.alpha_inv = d_small + SMALL_ALPHA_INV; a.gate_scale = d_small + SMALL_GATE_SCALE; a.alpha_pow = d_small + SMALL_ALPHA_POW;
    a.alpha_limbs = (const u32 *)c->alpha_limbs.p;
    a.code = (const u32 *)c->d_code.p; a.gates = (const GateDev *)c->d_gates.p; a.out = c->qvals.u();
    a.stage_list = (const u32 *)c->d_stage.p; a.num_wires = W; a.use_native = 1; a.rc = ctx->d_rc;
    a.N = N; a.lgN = lgN; a.rate_bits = p.rate_bits; a.num_gates = NG; a.num_selectors = c->num_selectors;
    a.num_constants = NC; a.num_routed = NR; a.chunk = Q; a.nchunks = nchunks; a.num_challenges = CH; a.num_regs = c->dev_regs;
    a.leaf0 = (u64)c->bf * n; a.count = (u64)c->nblocks() * n; a.stride = a.count;
    if (be.status) return be.status;
    // a sharded circuit fills its own leaf blocks and leaves zeros elsewhere: the ranks' buffers sum (or OR) to the values
    if (c->sharded()) LCP2_HIP(ctx, hipMemsetAsync(c->qvals.p, 0, (size_t)CH * N * 8, s));
    ProfScope ps(ctx, LCP2_K_QUOTIENT, (double)a.count * 8.0 * (W + ncs + CH * (1.0 + npp) + 2.0 + CH) + 8.0 * n * (W + NC));
    // the gate constraints on the n rows of H first (1/8 of the work below): a witness that violates one is the Err of prove()
    QuotientArgs h = a;
    h.wires = c->d_wires_cur; h.consts = c->rows_mode ? c->cs_rows.u() : c->cs_values.u(); h.leaf0 = 0; h.count = c->rows(); h.stride = c->rows();
    launch_gate_check(s, h, c->dev_gates, (unsigned long long *)(d_small + SMALL_CHECK));
    launch_quotient(s, a, c->dev_gates);
  }
  LCP2_HIP(ctx, hipGetLastError());
  if (c->local_quotient()) {  // block b is the coset of shift g w_N^bitrev(b), its values in bit-reversed order: interpolate in place
    ProfScope ps(ctx, LCP2_K_INTT, 16.0 * n * CH * c->nblocks());
    for (u32 b = c->bf; b < c->bf + c->nblocks(); b++) {
      const u64 shift = gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(lgN), bitrev32(b, p.rate_bits)));
      ntt.inverse_bitrev_in(c->qvals.u() + (u64)b * n, N, c->qvals.u() + (u64)b * n, N, p.degree_bits, CH, shift);
    }
    if (be.status) return be.status;
  }
  c->check_pending = defer_check;
  if (!defer_check) {
    u64 bad_row = ~0ull;
    LCP2_TRY(download(ctx, &bad_row, d_small + SMALL_CHECK, 8));  // synchronises the stream
    if (bad_row != ~0ull) return ctx->fail(LCP2_E_UNSAT, "the witness violates a gate constraint on row " + std::to_string(bad_row - 1 + c->row0()));
  }
  c->stage = lcp2_circuit::ST_QVALS;
  return LCP2_OK;
}

// coset iNTT of the (complete) quotient values, chunking, commitment
int stage_quotient_commit(lcp2_circuit *c, u64 *cap_out) {
  LCP2_STAGE_PROLOGUE
  if (c->stage != lcp2_circuit::ST_QVALS) return ctx->fail(LCP2_E_INVALID, "lcp2_quotient_commit: no quotient values");
  LCP2_HIP(ctx, c->quot.coeffs.ensure((size_t)CH * N * 8));
  if (c->local_quotient()) {
    const u32 R = 1u << p.rate_bits;
    if (!c->q_combine.p) {
      // interpolant_b = sum_k (s_b^n)^k Q_k with s_b^n = g^n w_R^bitrev(b)  =>  Q_k = g^(-n k) / R * sum_b w_R^(-bitrev(b) k) interpolant_b
      std::vector<u64> m((size_t)R * R);
      const u64 gninv = gl_inv(gl_pow(GL_GENERATOR, n)), wrinv = gl_inv(gl_root_of_unity(p.rate_bits)), rinv = gl_inv(R);
      for (u32 k = 0; k < R; k++)
        for (u32 b = 0; b < R; b++)
          m[(size_t)k * R + b] = gl_mul(gl_mul(gl_pow(gninv, k), rinv), gl_pow(wrinv, (u64)bitrev32(b, p.rate_bits) * k));
      LCP2_TRY(upload(ctx, c->q_combine, m.data(), m.size() * 8));
      LCP2_HIP(ctx, hipStreamSynchronize(s));  // `m` is a stack-lifetime staging buffer
    }
    ProfScope ps(ctx, LCP2_K_INTT, 16.0 * N * CH);
    launch_quotient_combine(s, c->qvals.u(), c->quot.coeffs.u(), c->q_combine.u(), n, R, N, CH);
    LCP2_HIP(ctx, hipGetLastError());
  } else {
    ProfScope ps(ctx, LCP2_K_INTT, 16.0 * N * CH);
    ntt.inverse_bitrev_in(c->qvals.u(), N, c->quot.coeffs.u(), N, lgN, CH, GL_GENERATOR);
  }
  if (be.status) return be.status;
  // N = Q n: the 8n coefficients of challenge c are exactly its Q chunks of n coefficients, already contiguous
  LCP2_TRY(commit_coeffs_dev(ctx, c->quot.coeffs.u(), CH * Q, p.degree_bits, p.rate_bits, p.cap_height, &c->quot, false));
  u64 bad_row = ~0ull;
  {
    Download d(ctx);
    LCP2_TRY(queue_cap(d, c, c->quot, cap_out));
    if (c->check_pending) LCP2_TRY(d.add(&bad_row, c->small.u() + SMALL_CHECK, 8));
    LCP2_TRY(d.wait());
  }
  if (c->check_pending && bad_row != ~0ull) {  // plonky2 would have produced an invalid proof here; this is the Err of prove()
    c->check_pending = false;
    c->stage = lcp2_circuit::ST_ZS;
    return ctx->fail(LCP2_E_UNSAT, "the witness violates a gate constraint on row " + std::to_string(bad_row - 1 + c->row0()));
  }
  c->check_pending = false;
  c->stage = lcp2_circuit::ST_QUOT;
  return LCP2_OK;
}

int stage_quotient(lcp2_circuit *c, const u64 *alphas, const u64 *pi_hash, u64 *cap_out) {
  if (c->sharded()) return c->ctx->fail(LCP2_E_INVALID, "sharded circuit: use lcp2_quotient_values, exchange the buffer, then lcp2_quotient_commit");
  LCP2_TRY(stage_quotient_values(c, alphas, pi_hash, true));  // the verdict of the gate check arrives with the cap: one synchronisation
  return stage_quotient_commit(c, cap_out);
}

// OpeningSet::new + PolynomialBatch::prove_openings (K7-K9, a13) in three phases (state in c->fo).  The challenger has observed
// everything up to the quotient cap and zeta was drawn from it; at the end it has absorbed the openings, the FRI caps, the final
// polynomial and the PoW witness and produced the query indices.  Together the phases write proof words [op_constants, total).
//
// A coset-sharded circuit (rank = bf / bc of world = 2^rate_bits / bc) does a share of the first two phases:
//   openings: the columns column_shard(rank) of every oracle (each rank holds all coefficients); zeros for the others
//   commit:   FRI layer 0 (the big one: LDE, leaf hashing, Merkle levels) for its own leaf blocks; its cap entries at their
//             global position.  Folding happens in coefficient form, so no values cross the ranks.
// and the caller sums the shares (lcp2_proof_section) before the next phase.

// first column and count of this rank's share of `nc` columns (the whole range for an unsharded circuit)
void column_share(const lcp2_circuit *c, u32 nc, u32 &first, u32 &count) {
  first = 0; count = nc;
  if (!c->sharded()) return;
  const u32 world = (1u << c->p.rate_bits) / c->bc, rank = c->bf / c->bc;
  const u32 base = nc / world, extra = nc % world;
  first = rank * base + std::min(rank, extra);
  count = base + (rank < extra ? 1 : 0);
}

int fri_open_openings(lcp2_circuit *c, u64 *proof) {
  LCP2_STAGE_PROLOGUE
  if (c->stage != lcp2_circuit::ST_QUOT) return ctx->fail(LCP2_E_INVALID, "lcp2_fri_open: the quotient is not committed");
  if (!c->cap_final) return ctx->fail(LCP2_E_INVALID, "sharded circuit: lcp2_circuit_set_constants_cap has not been called");
  FriOpenState &fo = c->fo;
  const gl2 zeta = fo.zeta, g_zeta = gl2_scale(zeta, gl_root_of_unity(p.degree_bits));
  lcp2_oracle *oracles[4] = {&c->cs, &c->wires, &c->zs, &c->quot};
  memset(proof + L.op_constants, 0, (L.total - L.op_constants) * 8);
  // the power tables of zeta and g zeta (device-made), every oracle's columns evaluated back to back, ONE copy back
  u64 *d_tab = c->tables.u(), *d_tab_g = d_tab + eval_table_words(n);
  const u32 all_cols = ncs + W + CH * (1 + npp) + CH * Q + CH;
  LCP2_HIP(ctx, c->open_out.ensure((size_t)2 * all_cols * 8));
  u64 *d_open = c->open_out.u();
  std::vector<u64> tmp((size_t)2 * all_cols);
  u32 share_cols = 0;
  for (int o = 0; o < 4; o++) { u32 f, k; column_share(c, oracles[o]->ncols, f, k); share_cols += k; }
  ProfScope ps(ctx, LCP2_K_OPENINGS, 8.0 * n * (share_cols + CH));
  eval_tables(c, zeta, d_tab);
  const bool with_next = !c->sharded() || c->bf == 0;
  if (with_next) eval_tables(c, g_zeta, d_tab_g);
  u32 at_col[5], first_col[4], num_cols[4], pos = 0;
  for (int o = 0; o < 4; o++) {
    column_share(c, oracles[o]->ncols, first_col[o], num_cols[o]);
    at_col[o] = pos;
    if (num_cols[o]) eval_columns(c, oracles[o]->coeffs.u() + (size_t)first_col[o] * n, num_cols[o], zeta, d_open + 2 * pos, d_tab);
    pos += num_cols[o];
  }
  at_col[4] = pos;
  if (with_next) { eval_columns(c, c->zs.coeffs.u(), CH, g_zeta, d_open + 2 * pos, d_tab_g); pos += CH; }
  LCP2_HIP(ctx, hipGetLastError());
  LCP2_TRY(download(ctx, tmp.data(), d_open, (size_t)2 * pos * 8));
  for (int o = 0; o < 4; o++)
    for (u32 j = 0; j < num_cols[o]; j++) {
      const u32 col = first_col[o] + j;
      size_t at = o == 0 ? L.op_constants + 2 * col   // constants then sigmas, contiguous
                : o == 1 ? L.op_wires + 2 * col
                : o == 2 ? (col < CH ? L.op_zs + 2 * col : L.op_pp + 2 * (col - CH))
                         : L.op_quot + 2 * col;
      proof[at] = tmp[2 * (at_col[o] + j)]; proof[at + 1] = tmp[2 * (at_col[o] + j) + 1];
    }
  if (with_next) memcpy(proof + L.op_zs_next, tmp.data() + 2 * at_col[4], 2 * CH * 8);
  fo.phase = 1;
  return LCP2_OK;
}

// LDE of the coefficients in fri_c[cur] (m of them, zero padding to 8m implicit), leaf hashing and Merkle levels of FRI layer l;
// the cap lands in the proof.  Layer 0 of a sharded circuit covers its own leaf blocks.
int fri_commit_layer(lcp2_circuit *c, u32 l, int cur, u64 m, u64 shift, u64 *proof) {
  LCP2_STAGE_PROLOGUE
  const u32 ab = p.fri_arity_bits[l], arity = 1u << ab;
  const bool part = l == 0 && c->sharded();
  u32 lgm = 0;
  while ((1ull << lgm) < m) lgm++;
  const u64 nvals = part ? (u64)c->bc * m : m << p.rate_bits, nleaves = nvals >> ab;
  u64 *vals = c->fri_vals[l].u();
  {
    ProfScope ps(ctx, LCP2_K_FRI, 16.0 * m + 16.0 * nvals + 32.0 * nleaves);
    // coset_fft of the zero-padded coefficients = 2^rate_bits coset transforms of the m coefficients; leaf order out
    if (part) ntt.forward(c->fri_c[cur].u(), m, vals, nvals, lgm, 2, shift, p.rate_bits, c->bf, c->bc);
    else ntt.forward(c->fri_c[cur].u(), m, vals, nvals, lgm, 2, shift, p.rate_bits);
    if (be.status) return be.status;
    launch_hash_ext_leaves(s, vals, vals + nvals, arity, nleaves, c->fri_dig[l].u(), ctx->d_rc);
    const auto &off = c->fri_level_off[l];
    for (size_t k = 1; k < off.size(); k++)
      launch_merkle_level(s, c->fri_dig[l].u() + 4 * off[k - 1], c->fri_dig[l].u() + 4 * off[k], nleaves >> k, ctx->d_rc);
  }
  LCP2_HIP(ctx, hipGetLastError());
  u64 *cap = proof + L.fri_caps + l * L.capw;
  const u64 *d_cap = c->fri_dig[l].u() + 4 * c->fri_level_off[l].back();
  if (!part) return download(ctx, cap, d_cap, L.capw * 8);
  const size_t per_block = (size_t)4 << (p.cap_height - p.rate_bits);
  memset(cap, 0, L.capw * 8);
  return download(ctx, cap + c->bf * per_block, d_cap, per_block * c->bc * 8);
}

int fri_open_commit(lcp2_circuit *c, u64 *proof) {
  LCP2_STAGE_PROLOGUE
  FriOpenState &fo = c->fo;
  if (fo.phase != 1) return ctx->fail(LCP2_E_INVALID, "lcp2_fri_open_commit: call lcp2_fri_open_begin first");
  HostChallenger &ch = fo.ch;
  const gl2 zeta = fo.zeta, g_zeta = gl2_scale(zeta, gl_root_of_unity(p.degree_bits));
  lcp2_oracle *oracles[4] = {&c->cs, &c->wires, &c->zs, &c->quot};
  ch.observe_n(proof + L.op_constants, 2 * (ncs + W));
  ch.observe_n(proof + L.op_zs, 2 * CH);
  ch.observe_n(proof + L.op_pp, 2 * CH * npp);
  ch.observe_n(proof + L.op_quot, 2 * CH * Q);
  ch.observe_n(proof + L.op_zs_next, 2 * CH);

  // ---- K7b: final polynomial of the batched opening
  const gl2 alpha = fo.alpha = ch.get_ext();
  {
    const u32 total_polys = ncs + W + CH * (1 + npp) + CH * Q;
    const u32 h = (p.degree_bits + 1) / 2;
    const u64 hi_count = (n >> h) + 1;
    // alpha^j and the two-level power tables of zeta, g zeta and their inverses: made on the device from the two challenges in
    // the kernel arguments (k_compose_tables; the host used to spend 0.3 ms here, then copy and synchronise)
    const size_t per = (size_t)2 * ((1ull << h) + hi_count);
    size_t off[4][2];
    for (int b = 0; b < 4; b++) { off[b][0] = (size_t)2 * total_polys + b * per; off[b][1] = off[b][0] + ((size_t)2 << h); }
    if (((size_t)2 * total_polys + 4 * per) * 8 > c->tables.bytes) return ctx->fail(LCP2_E_INVALID, "internal: table workspace too small");
    launch_compose_tables(s, alpha.c0, alpha.c1, zeta.c0, zeta.c1, gl_root_of_unity(p.degree_bits), total_polys, h, hi_count, c->tables.u());
    ComposeArgs a{};
    for (int o = 0; o < 4; o++) { a.coeffs[o] = oracles[o]->coeffs.u(); a.ncols[o] = oracles[o]->ncols; }
    a.num_challenges = CH; a.n = n;
    const u64 *T = c->tables.u();
    a.alpha_pows = T;
    a.z0_lo = T + off[0][0]; a.z0_hi = T + off[0][1]; a.z1_lo = T + off[1][0]; a.z1_hi = T + off[1][1];
    a.zi0_lo = T + off[2][0]; a.zi0_hi = T + off[2][1]; a.zi1_lo = T + off[3][0]; a.zi1_hi = T + off[3][1];
    a.zh = h; a.zmask = (1ull << h) - 1;
    gl2 ash = gl2_pow(alpha, CH);
    a.alpha_shift[0] = ash.c0; a.alpha_shift[1] = ash.c1;
    a.planes = c->planes.u();
    ProfScope ps(ctx, LCP2_K_OPENINGS, 8.0 * n * (total_polys + CH));
    launch_compose(s, a);
    launch_scan(s, false, c->planes.u(), c->planes.u(), c->scan_tmp.u(), n, true, 4, n);
    launch_divide_finalize(s, a, c->fri_c[0].u(), c->fri_c[0].u() + n);
  }
  LCP2_HIP(ctx, hipGetLastError());
  if (p.num_fri_layers) LCP2_TRY(fri_commit_layer(c, 0, 0, n, GL_GENERATOR, proof));
  fo.phase = 2;
  return LCP2_OK;
}

int fri_open_finish(lcp2_circuit *c, u64 *proof) {
  LCP2_STAGE_PROLOGUE
  FriOpenState &fo = c->fo;
  if (fo.phase != 2) return ctx->fail(LCP2_E_INVALID, "lcp2_fri_open_finish: call lcp2_fri_open_commit first");
  fo.phase = 0;
  HostChallenger &ch = fo.ch;
  gl2 *fri_betas = fo.fri_betas;
  std::vector<u64> &idx = fo.idx;
  lcp2_oracle *oracles[4] = {&c->cs, &c->wires, &c->zs, &c->quot};

  // ---- K8: FRI commit phase (layer 0 is committed already)
  u64 m = n;  // number of (possibly) non-zero coefficients; the zero padding to 8m is implicit
  u64 shift = GL_GENERATOR;
  int cur = 0;
  for (u32 l = 0; l < p.num_fri_layers; l++) {
    const u32 ab = p.fri_arity_bits[l], arity = 1u << ab;
    if (l) LCP2_TRY(fri_commit_layer(c, l, cur, m, shift, proof));
    ch.observe_n(proof + L.fri_caps + l * L.capw, L.capw);
    gl2 beta = ch.get_ext();
    fri_betas[l] = beta;
    {
      ProfScope ps(ctx, LCP2_K_FRI, 16.0 * m + 16.0 * (m >> ab));
      launch_fri_fold(s, c->fri_c[cur].u(), c->fri_c[cur].u() + m, c->fri_c[cur ^ 1].u(), c->fri_c[cur ^ 1].u() + (m >> ab), m >> ab, arity, beta.c0, beta.c1);
    }
    cur ^= 1;
    m >>= ab;
    shift = gl_pow(shift, arity);
  }
  if (m != L.final_len) return ctx->fail(LCP2_E_INVALID, "internal: final polynomial length mismatch");
  {
    std::vector<u64> f(2 * m);
    LCP2_TRY(download(ctx, f.data(), c->fri_c[cur].u(), 2 * m * 8));
    for (u64 i = 0; i < m; i++) { proof[L.final_poly + 2 * i] = f[i]; proof[L.final_poly + 2 * i + 1] = f[m + i]; }
  }
  ch.observe_n(proof + L.final_poly, 2 * L.final_len);

  // ---- K9: proof of work, minimum witness
  u64 pow_witness = 0;
  {
    PowArgs a{};
    ch.pow_state(a.state, a.pos);
    a.bits = p.proof_of_work_bits; a.rc = ctx->d_rc;
    u64 *d_res = c->small.u() + SMALL_POW;
    a.result = d_res;
    const u64 batch = 1ull << 20;
    u64 res = ~0ull;
    ProfScope ps(ctx, LCP2_K_POW, 0.0);
    for (u64 start = 0; res == ~0ull; start += batch) {
      if (start >= (1ull << 44)) return ctx->fail(LCP2_E_UNSUPPORTED, "proof of work not found");
      { SmallWords w{}; w.v[0] = ~0ull; launch_set_words(s, d_res, w, 1); }
      a.start = start;
      launch_pow_search(s, a, batch);
      LCP2_TRY(download(ctx, &res, d_res, 8));
    }
    pow_witness = res;
  }
  fo.pow_witness = proof[L.pow_witness] = pow_witness;
  ch.observe(pow_witness);
  {
    u64 resp = ch.get();
    if ((resp >> (64 - p.proof_of_work_bits)) != 0) return ctx->fail(LCP2_E_HIP, "internal: proof-of-work self check failed");
  }

  // ---- query phase: gather leaves and Merkle paths on the device, one copy back
  const u32 Qn = p.num_query_rounds;
  idx.assign(Qn * (1 + p.num_fri_layers), 0);
  for (u32 q = 0; q < Qn; q++) {
    u64 x = ch.get() % N;
    idx[q] = x;
    u64 xi = x;
    for (u32 l = 0; l < p.num_fri_layers; l++) { xi >>= p.fri_arity_bits[l]; idx[(1 + l) * Qn + q] = xi; }
  }
  // a sharded circuit answers the initial-tree and FRI-layer-0 parts of the queries whose leaf it holds and leaves zeros for
  // the others (its share of the proof); the smaller FRI layers are replicated on every rank
  const u64 leaf0 = (u64)c->bf * n, nlocal = (u64)c->nblocks() * n;
  std::vector<u64> up(idx);
  std::vector<char> mine(Qn, 1);
  for (u32 q = 0; q < Qn; q++) {
    mine[q] = idx[q] >= leaf0 && idx[q] < leaf0 + nlocal;
    up.push_back(mine[q] ? idx[q] - leaf0 : 0);
  }
  for (u32 q = 0; q < Qn; q++) up.push_back(p.num_fri_layers ? up[idx.size() + q] >> p.fri_arity_bits[0] : 0);  // layer-0 leaf, local
  if (ctx->pin && up.size() * 8 <= lcp2_ctx::PIN_BYTES) {  // through the pinned staging buffer (every earlier download has been waited for)
    memcpy(ctx->pin, up.data(), up.size() * 8);
    LCP2_HIP(ctx, hipMemcpyAsync(c->q_idx.p, ctx->pin, up.size() * 8, hipMemcpyHostToDevice, s));
  } else {
    LCP2_HIP(ctx, hipMemcpyAsync(c->q_idx.p, up.data(), up.size() * 8, hipMemcpyHostToDevice, s));
  }
  {
    u64 *d_idx = c->q_idx.u();
    const u64 *d_idx_local = d_idx + idx.size(), *d_idx_local0 = d_idx_local + Qn;
    u64 *buf = c->q_buf.u();
    size_t pos = 0;
    size_t o_leaf[4], o_sib[4], f_leaf[LCP2_MAX_FRI_LAYERS], f_sib[LCP2_MAX_FRI_LAYERS];
    for (int o = 0; o < 4; o++) {
      o_leaf[o] = pos; pos += (size_t)Qn * oracles[o]->ncols;
      o_sib[o] = pos; pos += (size_t)Qn * L.q_init_sib * 4;
      LCP2_TRY(open_oracle(ctx, *oracles[o], d_idx_local, Qn, buf + o_leaf[o], buf + o_sib[o]));
    }
    for (u32 l = 0; l < p.num_fri_layers; l++) {
      const u32 arity = 1u << p.fri_arity_bits[l];
      u64 nvals = N;  // values of layer l: N >> (arity bits of the layers before it)
      for (u32 k = 0; k < l; k++) nvals >>= p.fri_arity_bits[k];
      if (l == 0) nvals = nlocal;
      const u64 *d_leaf = l == 0 ? d_idx_local0 : d_idx + (1 + l) * Qn;
      f_leaf[l] = pos; pos += (size_t)Qn * 2 * arity;
      f_sib[l] = pos; pos += (size_t)Qn * L.q_step_sib[l] * 4;
      launch_gather_ext_leaves(s, c->fri_vals[l].u(), c->fri_vals[l].u() + nvals, arity, d_leaf, Qn, buf + f_leaf[l]);
      launch_gather_digests(s, c->fri_dig[l].u(), c->fri_d_level_off[l].u(), (u32)L.q_step_sib[l], d_leaf, Qn, buf + f_sib[l]);
    }
    LCP2_HIP(ctx, hipGetLastError());
    if (pos * 8 > c->q_buf.bytes) return ctx->fail(LCP2_E_INVALID, "internal: query workspace too small");
    std::vector<u64> h(pos);
    LCP2_TRY(download(ctx, h.data(), buf, pos * 8));
    for (u32 q = 0; q < Qn; q++) {
      u64 *R = proof + L.queries + (size_t)q * L.query_words;
      for (int o = 0; o < 4 && mine[q]; o++) {
        u32 nc = oracles[o]->ncols;
        memcpy(R + L.q_init_off[o], h.data() + o_leaf[o] + (size_t)q * nc, nc * 8);
        memcpy(R + L.q_init_off[o] + nc, h.data() + o_sib[o] + (size_t)q * L.q_init_sib * 4, L.q_init_sib * 32);
      }
      for (u32 l = 0; l < p.num_fri_layers; l++) {
        if (l == 0 && !mine[q]) continue;
        const u32 arity = 1u << p.fri_arity_bits[l];
        memcpy(R + L.q_step_off[l], h.data() + f_leaf[l] + (size_t)q * 2 * arity, 2 * arity * 8);
        memcpy(R + L.q_step_off[l] + 2 * arity, h.data() + f_sib[l] + (size_t)q * L.q_step_sib[l] * 4, L.q_step_sib[l] * 32);
      }
    }
  }
  // Shares must SUM to the proof (RCCL has no bitwise reductions): the words every rank holds identically (openings,
  // FRI caps, the smaller FRI layers, final polynomial, PoW witness) are contributed by the rank that holds leaf block 0 only.
  if (c->sharded() && c->bf != 0) {
    std::vector<u64> keep(proof + L.queries, proof + L.queries + (size_t)Qn * L.query_words);
    memset(proof + L.op_constants, 0, (L.total - L.op_constants) * 8);
    const size_t own_words = p.num_fri_layers >= 2 ? L.q_step_off[1] : L.query_words;  // initial trees and FRI layer 0 come first
    for (u32 q = 0; q < Qn; q++)
      if (mine[q]) memcpy(proof + L.queries + (size_t)q * L.query_words, keep.data() + (size_t)q * L.query_words, own_words * 8);
  }
  return LCP2_OK;
}

// the three phases back to back (a circuit that holds every leaf block)
int stage_fri_open(lcp2_circuit *c, gl2 zeta, HostChallenger &ch, u64 *proof, gl2 &alpha, gl2 *fri_betas, u64 &pow_witness, std::vector<u64> &idx) {
  if (c->sharded()) return c->ctx->fail(LCP2_E_INVALID, "sharded circuit: lcp2_fri_open_begin / _commit / _finish with their exchange steps");
  FriOpenState &fo = c->fo;
  fo.zeta = zeta; fo.ch = ch;
  LCP2_TRY(fri_open_openings(c, proof));
  LCP2_TRY(fri_open_commit(c, proof));
  LCP2_TRY(fri_open_finish(c, proof));
  ch = fo.ch; alpha = fo.alpha; pow_witness = fo.pow_witness; idx = fo.idx;
  for (u32 l = 0; l < c->p.num_fri_layers; l++) fri_betas[l] = fo.fri_betas[l];
  return LCP2_OK;
}
}  // namespace

extern "C" int lcp2_prove(lcp2_circuit *c, const uint64_t *wires_in_, lcp2_mem wires_mem, const uint64_t *public_inputs_, size_t num_public_inputs,
                          uint64_t *proof_, size_t proof_words) {
  if (!c || !wires_in_ || !proof_ || (c->npi && !public_inputs_)) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;  // verifier-only circuit
  if (num_public_inputs != c->npi) return c->ctx->fail(LCP2_E_INVALID, "lcp2_prove: public input count does not match the circuit");
  if (proof_words != ProofLayout(c->p).total) return c->ctx->fail(LCP2_E_INVALID, "lcp2_prove: proof buffer is not lcp2_proof_words() long");
  if (c->sharded()) return c->ctx->fail(LCP2_E_INVALID, "sharded circuit: drive the stages and their exchange steps (parallel.py ShardedProver)");
  const u64 *public_inputs = (const u64 *)public_inputs_;
  u64 *proof = (u64 *)proof_;
  const lcp2_params &p = c->p;
  const u32 CH = p.num_challenges;
  const ProofLayout L(p);
  memset(proof, 0, L.total * 8);
  HostPoseidon &H = HostPoseidon::get();
  std::vector<u64> pis(std::max<u32>(c->npi, 1), 0);
  for (u32 i = 0; i < c->npi; i++) pis[i] = gl_canon(public_inputs[i]);
  u64 pi_hash[4];
  H.hash_no_pad(pis.data(), c->npi, pi_hash);

  LCP2_TRY(stage_wires(c, (const u64 *)wires_in_, wires_mem, nullptr, proof + L.wires_cap));
  HostChallenger ch;
  ch.observe_n(c->digest, 4);
  ch.observe_n(pi_hash, 4);
  ch.observe_n(proof + L.wires_cap, L.capw);
  u64 betas[4] = {0}, gammas[4] = {0}, alphas[4] = {0};
  for (u32 k = 0; k < CH; k++) betas[k] = ch.get();
  for (u32 k = 0; k < CH; k++) gammas[k] = ch.get();
  LCP2_TRY(stage_perm_zs(c, betas, gammas, proof + L.zs_cap));
  ch.observe_n(proof + L.zs_cap, L.capw);
  for (u32 k = 0; k < CH; k++) alphas[k] = ch.get();
  LCP2_TRY(stage_quotient(c, alphas, pi_hash, proof + L.quot_cap));
  ch.observe_n(proof + L.quot_cap, L.capw);
  const gl2 zeta = ch.get_ext();
  gl2 alpha, fri_betas[LCP2_MAX_FRI_LAYERS];
  u64 pow_witness = 0;
  std::vector<u64> idx;
  LCP2_TRY(stage_fri_open(c, zeta, ch, proof, alpha, fri_betas, pow_witness, idx));
  const u32 Qn = p.num_query_rounds;
  // record the transcript for stage-wise parity tests
  u64 *lc = c->last_challenges;
  memset(lc, 0, sizeof c->last_challenges);
  memcpy(lc, betas, 32); memcpy(lc + 4, gammas, 32); memcpy(lc + 8, alphas, 32);
  lc[12] = zeta.c0; lc[13] = zeta.c1; lc[14] = alpha.c0; lc[15] = alpha.c1;
  for (u32 l = 0; l < p.num_fri_layers; l++) { lc[16 + 2 * l] = fri_betas[l].c0; lc[17 + 2 * l] = fri_betas[l].c1; }
  lc[32] = pow_witness;
  for (u32 q = 0; q < Qn; q++) lc[33 + q] = idx[q];
  return LCP2_OK;
}

// ---- staged host witnesses: the upload of the next witness overlaps the proof in flight (include/lcp2.h)
extern "C" int lcp2_witness_stage(lcp2_circuit *c, const uint64_t *wires, uint32_t slot) {
  if (!c || !wires || slot > 1) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  lcp2_ctx *ctx = c->ctx;
  if (c->sharded()) return ctx->fail(LCP2_E_INVALID, "lcp2_witness_stage: a sharded circuit takes its witness shard by shard");
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t)c->p.num_wires * 8 << c->p.degree_bits;
  if (!ctx->copy_stream) LCP2_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (!c->wit_ready[slot]) LCP2_HIP(ctx, hipEventCreateWithFlags(&c->wit_ready[slot], hipEventDisableTiming));
  LCP2_HIP(ctx, c->wit_slot[slot].ensure(bytes));
  LCP2_HIP(ctx, hipMemcpyAsync(c->wit_slot[slot].p, wires, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
  LCP2_HIP(ctx, hipEventRecord(c->wit_ready[slot], ctx->copy_stream));
  c->wit_staged[slot] = true;
  return LCP2_OK;
}
extern "C" int lcp2_prove_staged(lcp2_circuit *c, uint32_t slot, const uint64_t *public_inputs, size_t num_public_inputs, uint64_t *proof, size_t proof_words) {
  if (!c || slot > 1) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  if (!c->wit_staged[slot]) return c->ctx->fail(LCP2_E_INVALID, "lcp2_prove_staged: nothing has been staged into this slot");
  LCP2_HIP(c->ctx, hipStreamWaitEvent(c->ctx->stream, c->wit_ready[slot], 0));  // the stream waits for the upload, the host does not
  c->wit_staged[slot] = false;
  return lcp2_prove(c, (const uint64_t *)c->wit_slot[slot].p, LCP2_MEM_DEVICE, public_inputs, num_public_inputs, proof, proof_words);
}

// ---- C ABI of the seams (include/lcp2.h)
extern "C" int lcp2_commit_wires(lcp2_circuit *c, const uint64_t *wires, lcp2_mem mem, uint64_t *cap) {
  if (!c || !wires || !cap) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  return stage_wires(c, (const u64 *)wires, mem, nullptr, (u64 *)cap);
}
extern "C" int lcp2_commit_wires_coeffs(lcp2_circuit *c, const uint64_t *wires, const uint64_t *coeffs, uint64_t *cap) {
  if (!c || !wires || !coeffs || !cap) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  return stage_wires(c, (const u64 *)wires, LCP2_MEM_DEVICE, (const u64 *)coeffs, (u64 *)cap);
}
extern "C" int lcp2_commit_wires_rows(lcp2_circuit *c, const uint64_t *wire_rows, const uint64_t *coeffs, uint64_t *cap) {
  if (!c || !wire_rows || !coeffs || !cap) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  return stage_wires(c, (const u64 *)wire_rows, LCP2_MEM_DEVICE, (const u64 *)coeffs, (u64 *)cap, true);
}
// ---- the chunked form of lcp2_commit_wires_rows (include/lcp2.h)
extern "C" int lcp2_commit_wires_rows_begin(lcp2_circuit *c, const uint64_t *wire_rows) {
  if (!c || !wire_rows) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  lcp2_ctx *ctx = c->ctx;
  const lcp2_params &p = c->p;
  const u64 n = 1ull << p.degree_bits;
  if (!c->sharded() || n < c->world() || p.num_wires <= 4)
    return ctx->fail(LCP2_E_INVALID, "lcp2_commit_wires_rows_begin: needs a sharded circuit with at least one row per rank and more than 4 wires");
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  c->rows_mode = true; c->perm_phase = 0; c->fo.phase = 0; c->stage = lcp2_circuit::ST_NONE;
  unsigned long long *d_flag = (unsigned long long *)(c->small.u() + SMALL_NONCANON);
  launch_set_words(s, c->small.u() + SMALL_NONCANON, SmallWords{}, 1);
  launch_canon_copy(s, (const u64 *)wire_rows, nullptr, (u64)p.num_wires * c->rows(), d_flag);
  if (!c->cs_rows_ready) {  // the gate check reads the constants with the stride of the wires
    const u64 R = c->rows();
    LCP2_HIP(ctx, c->cs_rows.ensure((size_t)p.num_constants * R * 8));
    launch_copy_2d(s, c->cs_rows.u(), R, c->cs_values.u() + c->row0(), n, R, p.num_constants);
    c->cs_rows_ready = true;
  }
  lcp2_oracle *o = &c->wires;
  o->ctx = ctx; o->ncols = p.num_wires; o->log_n = p.degree_bits; o->rate_bits = p.rate_bits; o->cap_height = p.cap_height;
  LCP2_HIP(ctx, o->coeffs.ensure((size_t)p.num_wires * n * 8));
  LCP2_HIP(ctx, o->lde.ensure((size_t)p.num_wires * o->nleaves() * 8));
  LCP2_TRY(merkle_alloc_dev(ctx, o));
  LCP2_HIP(ctx, c->leaf_state.ensure((size_t)12 * o->nleaves() * 8));
  c->d_wires_cur = (const u64 *)wire_rows;
  c->chunk_next = 0;
  return LCP2_OK;
}
extern "C" int lcp2_commit_wires_chunk(lcp2_circuit *c, const uint64_t *coeffs, uint32_t first_col, uint32_t ncols) {
  if (!c || !coeffs) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  lcp2_ctx *ctx = c->ctx;
  const lcp2_params &p = c->p;
  if (c->chunk_next < 0 || (int)first_col != c->chunk_next) return ctx->fail(LCP2_E_INVALID, "lcp2_commit_wires_chunk: chunks come in column order after lcp2_commit_wires_rows_begin");
  if (ncols == 0 || first_col % 8 || first_col + ncols > p.num_wires || (ncols % 8 && first_col + ncols != p.num_wires))
    return ctx->fail(LCP2_E_INVALID, "lcp2_commit_wires_chunk: a chunk starts at a multiple of 8 columns and is a multiple of 8 long unless it is the last");
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  lcp2_oracle *o = &c->wires;
  const u64 n = 1ull << p.degree_bits, N = o->nleaves();
  hipStream_t s = ctx->stream;
  u64 *dst = o->coeffs.u() + (size_t)first_col * n;
  if ((const u64 *)coeffs != dst) LCP2_HIP(ctx, hipMemcpyAsync(dst, coeffs, (size_t)ncols * n * 8, hipMemcpyDeviceToDevice, s));
  DeviceNttBackend be{ctx};
  NttHost<DeviceNttBackend> ntt(be);
  {
    ProfScope ps(ctx, LCP2_K_LDE, (double)ncols * (8.0 * n + 8.0 * N));
    ntt.forward(dst, n, o->lde.u() + (size_t)first_col * N, N, p.degree_bits, ncols, GL_GENERATOR, p.rate_bits, o->block_first, o->block_count);
  }
  if (be.status) return be.status;
  const bool last = first_col + ncols == p.num_wires;
  {
    ProfScope ps(ctx, LCP2_K_LEAF_HASH, (double)N * (8.0 * ncols + (last ? 32.0 : 0.0)));
    launch_hash_leaves_absorb(s, o->lde.u() + (size_t)first_col * N, N, ncols, N, c->leaf_state.u(), first_col == 0, last, o->digests.u(), ctx->d_rc);
  }
  LCP2_HIP(ctx, hipGetLastError());
  c->chunk_next = (int)(first_col + ncols);
  return LCP2_OK;
}
extern "C" int lcp2_commit_wires_rows_finish(lcp2_circuit *c, uint64_t *cap) {
  if (!c || !cap) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  lcp2_ctx *ctx = c->ctx;
  if (c->chunk_next != (int)c->p.num_wires) return ctx->fail(LCP2_E_INVALID, "lcp2_commit_wires_rows_finish: not every column has been absorbed");
  c->chunk_next = -1;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  LCP2_TRY(merkle_levels_dev(ctx, &c->wires));
  Download d(ctx);
  LCP2_TRY(queue_cap(d, c, c->wires, (u64 *)cap));
  LCP2_TRY(d.add(&c->noncanon_host, c->small.u() + SMALL_NONCANON, 8));
  LCP2_TRY(d.wait());
  c->stage = lcp2_circuit::ST_WIRES;
  return LCP2_OK;
}
extern "C" int lcp2_perm_zs_rows_begin(lcp2_circuit *c, const uint64_t *betas, const uint64_t *gammas, uint64_t *block_products) {
  if (!c || !betas || !gammas || !block_products) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  if (!c->rows_mode) return c->ctx->fail(LCP2_E_INVALID, "lcp2_perm_zs_rows_begin: the wires were not committed with lcp2_commit_wires_rows");
  c->perm_phase = 0;
  LCP2_TRY(perm_begin(c, (const u64 *)betas, (const u64 *)gammas));
  {
    Download d(c->ctx);
    LCP2_TRY(queue_perm_wrap(d, c));
    LCP2_TRY(d.wait());
  }
  const u32 CH = c->p.num_challenges;
  memset(block_products, 0, (size_t)c->world() * CH * 8);
  for (u32 k = 0; k < CH; k++) block_products[(size_t)c->rank() * CH + k] = gl_mul(c->perm_wrap[2 * k], c->perm_wrap[2 * k + 1]);
  c->perm_phase = 1;
  return LCP2_OK;
}
extern "C" int lcp2_perm_zs_rows_finish(lcp2_circuit *c, const uint64_t *block_products, uint64_t **device_ptr, size_t *words) {
  if (!c || !block_products || !device_ptr || !words) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  if (!c->rows_mode || c->stage < lcp2_circuit::ST_WIRES || c->perm_phase != 1)
    return c->ctx->fail(LCP2_E_INVALID, "lcp2_perm_zs_rows_finish: lcp2_perm_zs_rows_begin has not run for this proof");
  const u32 CH = c->p.num_challenges;
  u64 prefix[QUOTIENT_MAX_CH];
  for (u32 k = 0; k < CH; k++) {
    u64 all = 1;
    prefix[k] = 1;
    for (u32 r = 0; r < c->world(); r++) {
      if (r == c->rank()) prefix[k] = all;
      all = gl_mul(all, gl_canon(block_products[(size_t)r * CH + k]));
    }
    // every rank sees the same products: all of them report the broken copy constraint (stage_perm_zs has the argument)
    if (all != 1) return c->ctx->fail(LCP2_E_UNSAT, "the witness violates a copy constraint (the permutation product does not return to 1)");
  }
  LCP2_TRY(perm_finish(c, prefix));
  c->perm_phase = 2;
  *device_ptr = (uint64_t *)c->zs_rows.p;
  *words = (size_t)CH * (1 + npp_of(c->p)) << c->p.degree_bits;
  return LCP2_OK;
}
extern "C" int lcp2_perm_zs_commit(lcp2_circuit *c, uint64_t *cap) {
  if (!c || !cap) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  if (!c->rows_mode || c->stage < lcp2_circuit::ST_WIRES || c->perm_phase != 2)
    return c->ctx->fail(LCP2_E_INVALID, "lcp2_perm_zs_commit: lcp2_perm_zs_rows_finish has not run for this proof");
  c->perm_phase = 0;
  return perm_commit(c, (u64 *)cap);
}
extern "C" int lcp2_perm_zs(lcp2_circuit *c, const uint64_t *betas, const uint64_t *gammas, uint64_t *cap) {
  if (!c || !betas || !gammas || !cap) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  return stage_perm_zs(c, (const u64 *)betas, (const u64 *)gammas, (u64 *)cap);
}
extern "C" int lcp2_quotient(lcp2_circuit *c, const uint64_t *alphas, const uint64_t public_inputs_hash[4], uint64_t *cap) {
  if (!c || !alphas || !cap || !public_inputs_hash) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  return stage_quotient(c, (const u64 *)alphas, (const u64 *)public_inputs_hash, (u64 *)cap);
}
extern "C" int lcp2_quotient_values(lcp2_circuit *c, const uint64_t *alphas, const uint64_t public_inputs_hash[4]) {
  if (!c || !alphas || !public_inputs_hash) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  return stage_quotient_values(c, (const u64 *)alphas, (const u64 *)public_inputs_hash);
}
extern "C" int lcp2_quotient_buffer(lcp2_circuit *c, uint64_t **device_ptr, size_t *words) {
  if (!c || !device_ptr || !words) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  *device_ptr = (uint64_t *)c->qvals.p;
  *words = ((size_t)c->p.num_challenges << (c->p.degree_bits + c->p.rate_bits));
  return LCP2_OK;
}
extern "C" int lcp2_quotient_commit(lcp2_circuit *c, uint64_t *cap) {
  if (!c || !cap) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  return stage_quotient_commit(c, (u64 *)cap);
}
// host-side transcript helpers (plonky2 Challenger / PoseidonHash::hash_no_pad) for callers without their own
extern "C" void lcp2_challenger_init(lcp2_challenger *ch) { if (ch) memset(ch, 0, sizeof *ch); }
extern "C" int lcp2_challenger_observe(lcp2_challenger *chs, const uint64_t *values, size_t count) {
  if (!chs || (count && !values) || chs->input_len >= 8 || chs->output_len > 8) return LCP2_E_INVALID;  // 8 buffered inputs duplex at once: never stored
  HostChallenger ch;
  ch.load((const u64 *)chs->sponge, (const u64 *)chs->input, chs->input_len, (const u64 *)chs->output, chs->output_len);
  ch.observe_n((const u64 *)values, count);
  ch.save((u64 *)chs->sponge, (u64 *)chs->input, chs->input_len, (u64 *)chs->output, chs->output_len);
  return LCP2_OK;
}
extern "C" int lcp2_challenger_get(lcp2_challenger *chs, uint64_t *out, size_t count) {
  if (!chs || (count && !out) || chs->input_len >= 8 || chs->output_len > 8) return LCP2_E_INVALID;
  HostChallenger ch;
  ch.load((const u64 *)chs->sponge, (const u64 *)chs->input, chs->input_len, (const u64 *)chs->output, chs->output_len);
  for (size_t i = 0; i < count; i++) out[i] = ch.get();
  ch.save((u64 *)chs->sponge, (u64 *)chs->input, chs->input_len, (u64 *)chs->output, chs->output_len);
  return LCP2_OK;
}
extern "C" int lcp2_hash_no_pad(const uint64_t *values, size_t count, uint64_t out[4]) {
  if ((count && !values) || !out) return LCP2_E_INVALID;
  std::vector<u64> v(std::max<size_t>(count, 1), 0);
  for (size_t i = 0; i < count; i++) v[i] = gl_canon(values[i]);
  u64 h[4];
  HostPoseidon::get().hash_no_pad(v.data(), count, h);
  memcpy(out, h, 32);
  return LCP2_OK;
}

extern "C" int lcp2_fri_open(lcp2_circuit *c, const uint64_t zeta[2], lcp2_challenger *chs, uint64_t *proof) {
  if (!c || !zeta || !chs || !proof) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  if (chs->input_len >= 8 || chs->output_len > 8) return LCP2_E_INVALID;
  HostChallenger ch;
  ch.load((const u64 *)chs->sponge, (const u64 *)chs->input, chs->input_len, (const u64 *)chs->output, chs->output_len);
  gl2 alpha, fri_betas[LCP2_MAX_FRI_LAYERS];
  u64 pow_witness = 0;
  std::vector<u64> idx;
  LCP2_TRY(stage_fri_open(c, gl2_make(gl_canon(zeta[0]), gl_canon(zeta[1])), ch, (u64 *)proof, alpha, fri_betas, pow_witness, idx));
  ch.save((u64 *)chs->sponge, (u64 *)chs->input, chs->input_len, (u64 *)chs->output, chs->output_len);
  return LCP2_OK;
}

// the same stage in its three phases, for a coset-sharded proof (and for callers that want the exchange points)
extern "C" int lcp2_fri_open_begin(lcp2_circuit *c, const uint64_t zeta[2], const lcp2_challenger *chs, uint64_t *proof) {
  if (!c || !zeta || !chs || !proof) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  if (chs->input_len >= 8 || chs->output_len > 8) return LCP2_E_INVALID;
  c->fo.phase = 0;
  c->fo.ch.load((const u64 *)chs->sponge, (const u64 *)chs->input, chs->input_len, (const u64 *)chs->output, chs->output_len);
  c->fo.zeta = gl2_make(gl_canon(zeta[0]), gl_canon(zeta[1]));
  return fri_open_openings(c, (u64 *)proof);
}
extern "C" int lcp2_fri_open_commit(lcp2_circuit *c, uint64_t *proof) {
  if (!c || !proof) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  return fri_open_commit(c, (u64 *)proof);
}
extern "C" int lcp2_fri_open_finish(lcp2_circuit *c, lcp2_challenger *chs, uint64_t *proof) {
  if (!c || !proof) return LCP2_E_INVALID;
  if (!c->ctx) return LCP2_E_NODEVICE;
  LCP2_TRY(fri_open_finish(c, (u64 *)proof));
  if (chs) c->fo.ch.save((u64 *)chs->sponge, (u64 *)chs->input, chs->input_len, (u64 *)chs->output, chs->output_len);
  return LCP2_OK;
}
extern "C" int lcp2_proof_section(const lcp2_circuit *c, int section, size_t *first_word, size_t *num_words) {
  if (!c || !first_word || !num_words) return LCP2_E_INVALID;
  const ProofLayout L(c->p);
  switch (section) {
    case LCP2_SECTION_OPENINGS: *first_word = L.op_constants; *num_words = L.fri_caps - L.op_constants; return LCP2_OK;
    case LCP2_SECTION_FRI_CAP0: *first_word = L.fri_caps; *num_words = c->p.num_fri_layers ? L.capw : 0; return LCP2_OK;
    case LCP2_SECTION_AFTER_CAPS: *first_word = L.op_constants; *num_words = L.total - L.op_constants; return LCP2_OK;
  }
  return LCP2_E_INVALID;
}
