// CircuitBuilder / PartialWitness / CircuitData of the host circuit layer (see lc_plonky2.hpp).
//
// Model: a Target is a variable; gadgets allocate gate rows and BIND wire cells to variables; `connect` merges
// variables (union-find).  build() turns every class of routed cells into one cycle of the copy-constraint
// permutation (the sigma polynomials), emits selector / constant columns and the gate programs.  The generators
// of plonky2 are a list of steps evaluated in creation order (gadgets are built in dependency order, so one
// forward pass replaces plonky2's worklist); a value set twice with different results is the UnsatisfiedError
// that makes the reference's #[should_panic] tests panic.
#include <cstdio>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <map>
#include <thread>
#include "host_internal.hpp"

namespace lc {

const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
const uint32_t SHA_IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

static inline uint32_t rotr(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

// ------------------------------------------------------------------ PartialWitness
void PartialWitness::set_target(Target t, F value) { entries_.push_back({t.id, value % GOLDILOCKS_P}); }
void PartialWitness::set_hash256_target(const Hash256Target &t, const uint8_t v[32]) {
  for (int i = 0; i < 8; i++)
    set_target(t[i].t, ((uint32_t)v[4 * i] << 24) | ((uint32_t)v[4 * i + 1] << 16) | ((uint32_t)v[4 * i + 2] << 8) | v[4 * i + 3]);
}

// ------------------------------------------------------------------ CircuitBuilder
struct CircuitBuilder::Impl {
  std::unique_ptr<CircuitData::Impl> d{new CircuitData::Impl()};
  std::map<F, uint32_t> const_vars;
  int const_row = -1, const_slot = 2;
  std::map<std::pair<F, F>, std::pair<uint32_t, uint32_t>> arith_open;  // (c0, c1) -> (row, next slot)
  bool built = false;

  uint32_t new_var() { d->parent.push_back((uint32_t)d->parent.size()); return (uint32_t)d->parent.size() - 1; }
  uint32_t new_row(uint32_t gate, F c0 = 0, F c1 = 0) {
    d->gate_of_row.push_back(gate);
    d->row_consts.push_back({c0, c1});
    return d->nrows++;
  }
  void bind(uint32_t row, uint32_t col, uint32_t var) { d->cells.push_back({row, col, var}); }
};

CircuitBuilder::CircuitBuilder(const CircuitConfig &config) : impl_(new Impl()) {
  impl_->d->config = config;
  impl_->new_row(G_PUBLIC_INPUT);  // row 0 carries the public inputs (PublicInputGate)
}
CircuitBuilder::~CircuitBuilder() = default;

Target CircuitBuilder::add_virtual_target() { return Target{impl_->new_var()}; }
BoolTarget CircuitBuilder::add_virtual_bool_target_safe() {
  BoolTarget b{add_virtual_target()};
  assert_bool(b);  // "safe": constrained to {0, 1}
  return b;
}
Hash256Target CircuitBuilder::add_virtual_hash256_target() {
  Hash256Target h;
  for (auto &l : h) l = U32Target{add_virtual_target()};
  return h;
}

Target CircuitBuilder::constant(F v) {
  v %= GOLDILOCKS_P;
  auto it = impl_->const_vars.find(v);
  if (it != impl_->const_vars.end()) return Target{it->second};
  if (impl_->const_slot == 2) { impl_->const_row = (int)impl_->new_row(G_CONSTANT); impl_->const_slot = 0; }
  uint32_t var = impl_->new_var();
  impl_->d->row_consts[impl_->const_row][impl_->const_slot] = v;
  impl_->bind((uint32_t)impl_->const_row, (uint32_t)impl_->const_slot, var);
  impl_->const_slot++;
  Op op; op.kind = Op::CONST; op.out = var; op.c0 = v;
  impl_->d->ops.push_back(op);
  impl_->const_vars[v] = var;
  return Target{var};
}

static Target arith_op(CircuitBuilder::Impl *b, F c0, Target x, Target y, F c1, Target z) {
  auto key = std::make_pair(c0, c1);
  auto it = b->arith_open.find(key);
  if (it == b->arith_open.end() || it->second.second == ARITH_OPS) {
    uint32_t row = b->new_row(G_ARITHMETIC, c0, c1);
    b->arith_open[key] = {row, 0};
    it = b->arith_open.find(key);
  }
  uint32_t row = it->second.first, k = it->second.second++;
  uint32_t out = b->new_var();
  b->bind(row, 4 * k, x.id); b->bind(row, 4 * k + 1, y.id); b->bind(row, 4 * k + 2, z.id); b->bind(row, 4 * k + 3, out);
  Op op; op.kind = Op::ARITH; op.x = x.id; op.y = y.id; op.z = z.id; op.out = out; op.c0 = c0; op.c1 = c1;
  b->d->ops.push_back(op);
  return Target{out};
}
Target CircuitBuilder::arithmetic(F c0, Target x, Target y, F c1, Target z) { return arith_op(impl_.get(), c0 % GOLDILOCKS_P, x, y, c1 % GOLDILOCKS_P, z); }
Target CircuitBuilder::mul_add(Target a, Target b, Target c) { return arith_op(impl_.get(), 1, a, b, 1, c); }
Target CircuitBuilder::mul(Target a, Target b) { return arith_op(impl_.get(), 1, a, b, 0, zero()); }
Target CircuitBuilder::add(Target a, Target b) { return arith_op(impl_.get(), 1, a, one(), 1, b); }
Target CircuitBuilder::sub(Target a, Target b) { return arith_op(impl_.get(), 1, a, one(), GOLDILOCKS_P - 1, b); }
Target CircuitBuilder::add_many(const std::vector<Target> &terms) {
  if (terms.empty()) return zero();
  Target acc = terms[0];
  for (size_t i = 1; i < terms.size(); i++) acc = add(acc, terms[i]);
  return acc;
}
BoolTarget CircuitBuilder::not_(BoolTarget b) { return BoolTarget{arith_op(impl_.get(), GOLDILOCKS_P - 1, b.target, one(), 1, one())}; }
Target CircuitBuilder::select(BoolTarget b, Target x, Target y) { return mul_add(b.target, sub(x, y), y); }
void CircuitBuilder::assert_bool(BoolTarget b) { connect(mul(b.target, b.target), b.target); }
Target CircuitBuilder::le_sum(const std::vector<BoolTarget> &bits, size_t first, size_t count) {
  if (first >= bits.size()) return zero();
  const size_t end = count == (size_t)-1 || first + count > bits.size() ? bits.size() : first + count;
  Target acc = bits[end - 1].target;
  for (size_t i = end - 1; i-- > first;) acc = arith_op(impl_.get(), 2, acc, one(), 1, bits[i].target);  // 2 acc + bit
  return acc;
}
std::vector<BoolTarget> CircuitBuilder::split_le(Target x, size_t nbits) {
  if (nbits == 0 || nbits > 63) throw std::runtime_error("split_le: 1..63 bits");
  Op op; op.kind = Op::BITS; op.x = x.id; op.c0 = nbits;
  std::vector<BoolTarget> bits;
  for (size_t i = 0; i < nbits; i++) { BoolTarget b{add_virtual_target()}; op.internal.push_back(b.target.id); bits.push_back(b); }
  impl_->d->ops.push_back(op);
  for (auto &b : bits) assert_bool(b);
  connect(le_sum(bits), x);
  return bits;
}

Target CircuitBuilder::inverse(Target x) {
  Op op; op.kind = Op::INV; op.x = x.id; op.out = impl_->new_var();
  impl_->d->ops.push_back(op);
  Target inv{op.out};
  connect(mul(x, inv), one());
  return inv;
}
BoolTarget CircuitBuilder::is_equal(Target a, Target b) {
  // d = a - b, e = 1 - d * (1 / d or 0): e is 1 exactly when d = 0 once d * e = 0 holds
  Target d = sub(a, b);
  Op op; op.kind = Op::INV; op.x = d.id; op.out = impl_->new_var();
  impl_->d->ops.push_back(op);
  Target e = arithmetic(GOLDILOCKS_P - 1, d, Target{op.out}, 1, one());
  connect(mul(d, e), zero());
  return BoolTarget{e};
}
std::array<Target, 2> CircuitBuilder::hint_ext_inverse(Target x0, Target x1) {
  Op op; op.kind = Op::EXT_INV; op.x = x0.id; op.y = x1.id;
  std::array<Target, 2> out{add_virtual_target(), add_virtual_target()};
  op.internal = {out[0].id, out[1].id};
  impl_->d->ops.push_back(op);
  return out;
}
std::array<Target, 2> CircuitBuilder::hint_split_32(Target x) {
  Op op; op.kind = Op::SPLIT32; op.x = x.id;
  std::array<Target, 2> out{add_virtual_target(), add_virtual_target()};
  op.internal = {out[0].id, out[1].id};
  impl_->d->ops.push_back(op);
  return out;
}
std::vector<Target> CircuitBuilder::hint(const std::vector<Target> &inputs, size_t num_outputs, std::function<void(const std::vector<F> &, std::vector<F> &)> fn) {
  Op op; op.kind = Op::HINT; op.hint_fn = std::move(fn);
  for (Target t : inputs) op.hint_in.push_back(t.id);
  std::vector<Target> out;
  for (size_t i = 0; i < num_outputs; i++) { out.push_back(add_virtual_target()); op.internal.push_back(out.back().id); }
  impl_->d->ops.push_back(std::move(op));
  return out;
}
CircuitBuilder::CanonicalSplit CircuitBuilder::split_canonical(Target x, uint32_t hi_bits) {
  if (hi_bits < 1 || hi_bits > 32) throw std::runtime_error("split_canonical: 1..32 high bits");
  std::array<Target, 2> h = hint_split_32(x);
  CanonicalSplit r{h[0], h[1], split_le(h[0], 32)};
  std::vector<BoolTarget> hi = split_le(h[1], hi_bits);
  connect(arithmetic(1ull << 32, h[1], one(), 1, h[0]), x);
  if (hi_bits == 32) {
    BoolTarget top = is_equal(h[1], constant(0xFFFFFFFFull));
    connect(mul(top.target, h[0]), zero());
  }
  r.bits.insert(r.bits.end(), hi.begin(), hi.end());
  return r;
}
std::array<Target, 12> CircuitBuilder::poseidon(const std::array<Target, 12> &inputs, BoolTarget swap) {
  Impl *b = impl_.get();
  const uint32_t row = b->new_row(G_POSEIDON);
  Op op; op.kind = Op::POSEIDON; op.first_row = row; op.x = swap.target.id;
  std::array<Target, 12> out;
  for (uint32_t i = 0; i < 12; i++) {
    op.in[i] = inputs[i].id;
    out[i] = Target{b->new_var()};
    op.internal.push_back(out[i].id);
    b->bind(row, POS_WIRE_INPUT + i, op.in[i]);
    b->bind(row, POS_WIRE_OUTPUT + i, out[i].id);
  }
  b->bind(row, POS_WIRE_SWAP, swap.target.id);
  b->d->ops.push_back(op);
  return out;
}

void CircuitBuilder::connect(Target a, Target b) {
  auto &p = impl_->d->parent;
  uint32_t ra = impl_->d->find(a.id), rb = impl_->d->find(b.id);
  if (ra != rb) p[std::max(ra, rb)] = std::min(ra, rb);
}
// the inputs are bound at build(): hashed in-circuit, the digest connected to the PublicInputGate of row 0
void CircuitBuilder::register_public_input(Target t) { impl_->d->public_inputs.push_back(t.id); }
size_t CircuitBuilder::num_gates() const { return impl_->d->nrows; }
void CircuitBuilder::print_gate_counts(int) const {
  std::map<uint32_t, size_t> counts;
  for (auto g : impl_->d->gate_of_row) counts[g]++;
  printf("Total gate counts:\n");
  for (auto &kv : counts) printf("- %zu instances of %s\n", kv.second, gate_name(kv.first));
}

// two_to_one_sha256(left, right) = SHA-256 of the 64-byte message: data block + constant padding block
Hash256Target CircuitBuilder::two_to_one_sha256(const Hash256Target &left, const Hash256Target &right) {
  Impl *b = impl_.get();
  Op op; op.kind = Op::SHA;
  for (int i = 0; i < 8; i++) { op.in[i] = left[i].t.id; op.in[8 + i] = right[i].t.id; }
  uint32_t zero_var = zero().id;
  uint32_t iv[8];
  for (int i = 0; i < 8; i++) iv[i] = constant(SHA_IV[i]).id;
  op.first_row = b->d->nrows;
  // constant schedule of the padding block
  uint32_t wpad[64] = {0};
  wpad[0] = 0x80000000u; wpad[15] = 512;
  for (int t = 16; t < 64; t++) {
    uint32_t s0 = rotr(wpad[t - 15], 7) ^ rotr(wpad[t - 15], 18) ^ (wpad[t - 15] >> 3);
    uint32_t s1 = rotr(wpad[t - 2], 17) ^ rotr(wpad[t - 2], 19) ^ (wpad[t - 2] >> 10);
    wpad[t] = wpad[t - 16] + s0 + wpad[t - 7] + s1;
  }
  std::vector<uint32_t> W(64);
  for (int i = 0; i < 16; i++) W[i] = op.in[i];
  // 48 schedule rows
  for (int t = 16; t < 64; t++) {
    uint32_t row = b->new_row(G_SHA_SCHED);
    W[t] = b->new_var();
    op.internal.push_back(W[t]);
    b->bind(row, 0, W[t - 2]); b->bind(row, 1, W[t - 7]); b->bind(row, 2, W[t - 15]); b->bind(row, 3, W[t - 16]); b->bind(row, 4, W[t]);
  }
  uint32_t state[8];
  for (int i = 0; i < 8; i++) state[i] = iv[i];
  uint32_t chain[8];
  for (int i = 0; i < 8; i++) chain[i] = iv[i];
  for (int c = 0; c < 2; c++) {
    for (int t = 0; t < 64; t++) {
      uint32_t t1 = b->new_var(), a_new = b->new_var(), e_new = b->new_var();
      op.internal.push_back(t1); op.internal.push_back(a_new); op.internal.push_back(e_new);
      F kconst = c == 0 ? (F)SHA_K[t] : (F)SHA_K[t] + (F)wpad[t];
      uint32_t re = b->new_row(G_SHA_ROUND_E, kconst, 0);
      b->bind(re, 0, state[4]); b->bind(re, 1, state[5]); b->bind(re, 2, state[6]); b->bind(re, 3, state[7]); b->bind(re, 4, state[3]);
      b->bind(re, 5, c == 0 ? W[t] : zero_var); b->bind(re, 6, e_new); b->bind(re, 7, t1);
      uint32_t ra = b->new_row(G_SHA_ROUND_A);
      b->bind(ra, 0, state[0]); b->bind(ra, 1, state[1]); b->bind(ra, 2, state[2]); b->bind(ra, 3, t1); b->bind(ra, 4, a_new);
      state[7] = state[6]; state[6] = state[5]; state[5] = state[4]; state[4] = e_new;
      state[3] = state[2]; state[2] = state[1]; state[1] = state[0]; state[0] = a_new;
    }
    // chaining value + working state, 8 range-checked additions in 3 rows
    uint32_t outv[8];
    uint32_t row = 0;
    for (int i = 0; i < 8; i++) {
      if (i % SHA_ADD_OPS == 0) row = b->new_row(G_SHA_ADD);
      int j = i % SHA_ADD_OPS;
      outv[i] = b->new_var();
      op.internal.push_back(outv[i]);
      b->bind(row, 3 * j, chain[i]); b->bind(row, 3 * j + 1, state[i]); b->bind(row, 3 * j + 2, outv[i]);
    }
    for (int i = 0; i < 8; i++) { chain[i] = outv[i]; state[i] = outv[i]; }
  }
  Hash256Target out;
  for (int i = 0; i < 8; i++) { op.out8[i] = chain[i]; out[i] = U32Target{Target{chain[i]}}; }
  b->d->ops.push_back(op);
  return out;
}

// ------------------------------------------------------------------ build()
std::unique_ptr<CircuitData> CircuitBuilder::build() {
  Impl *b = impl_.get();
  if (b->built) throw std::runtime_error("CircuitBuilder::build called twice");
  b->built = true;
  std::unique_ptr<CircuitData> data(new CircuitData());
  CircuitData::Impl *d = b->d.get();
  const CircuitConfig &cfg = d->config;
  {
    // circuit_builder.rs::build: public_inputs_hash = hash_n_to_hash_no_pad::<PoseidonHash>(public_inputs) in-circuit (an
    // overwrite-mode sponge of rate 8 over PoseidonGate rows, starting from the zero state, swap = false), connected to
    // the wires of the PublicInputGate, which the gate compares with the hash the transcript uses
    const Target zero_t = zero();
    std::array<Target, 12> state;
    state.fill(zero_t);
    const size_t npi_ = d->public_inputs.size();
    for (size_t off = 0; off < npi_; off += 8) {
      for (uint32_t i = 0; i < 8 && off + i < npi_; i++) state[i] = Target{d->public_inputs[off + i]};
      state = poseidon(state);
    }
    for (uint32_t i = 0; i < 4; i++) b->bind(0, i, state[i].id);
  }
  uint32_t degree_bits = 5;  // room for the cap-height-4 Merkle trees of every FRI layer
  while ((1u << degree_bits) < d->nrows) degree_bits++;
  const uint64_t n = 1ull << degree_bits;
  const uint32_t NR = cfg.num_routed_wires, npi = (uint32_t)d->public_inputs.size();
  GateSetLayout gs = build_gate_set(cfg.max_quotient_degree_factor + 1);
  CircuitDescription &D = data->desc_;
  if (lcp2_params_standard(degree_bits, gs.num_selectors + cfg.num_constants, &D.params) != LCP2_OK) throw std::runtime_error("bad circuit size");
  D.params.num_wires = cfg.num_wires; D.params.num_routed_wires = NR; D.params.rate_bits = cfg.rate_bits; D.params.cap_height = cfg.cap_height;
  D.params.num_challenges = cfg.num_challenges; D.params.quotient_degree_factor = cfg.max_quotient_degree_factor;
  D.params.proof_of_work_bits = cfg.proof_of_work_bits; D.params.num_query_rounds = cfg.num_query_rounds;
  D.num_selectors = gs.num_selectors; D.gates = gs.gates; D.code = gs.code; D.imm = gs.imm; D.num_public_inputs = npi; D.num_regs = gs.num_regs;
  const uint32_t NC = D.params.num_constants;
  D.constants_sigmas.assign((size_t)(NC + NR) * n, 0);
  // selectors and gate constants
  for (uint64_t r = 0; r < n; r++) {
    uint32_t g = r < d->nrows ? d->gate_of_row[r] : (uint32_t)G_NOOP;
    for (uint32_t s = 0; s < gs.num_selectors; s++)
      D.constants_sigmas[(size_t)s * n + r] = (gs.groups[s].first <= g && g < gs.groups[s].second) ? g : 0xFFFFFFFFull;
    if (r < d->nrows)
      for (uint32_t k = 0; k < cfg.num_constants; k++) D.constants_sigmas[(size_t)(gs.num_selectors + k) * n + r] = d->row_consts[r][k];
  }
  // k_is = g^j and the subgroup
  D.k_is.resize(NR);
  F acc = 1;
  for (uint32_t j = 0; j < NR; j++) { D.k_is[j] = acc; acc = f_mul(acc, 7); }
  std::vector<F> sub(n);
  F w = f_root_of_unity(degree_bits);
  sub[0] = 1;
  for (uint64_t i = 1; i < n; i++) sub[i] = f_mul(sub[i - 1], w);
  // no connect() after this point: flatten the union-find (every variable points at its root), so that the find() of every
  // generator input and wire cell during witness generation is one step instead of a walk
  for (uint32_t v = 0; v < d->parent.size(); v++) d->parent[v] = d->find(v);
  // identity permutation, then one cycle per variable class
  uint64_t *sig = D.constants_sigmas.data() + (size_t)NC * n;
  for (uint32_t j = 0; j < NR; j++)
    for (uint64_t i = 0; i < n; i++) sig[(size_t)j * n + i] = f_mul(D.k_is[j], sub[i]);
  std::vector<uint8_t> taken((size_t)NR * n, 0);
  std::map<uint32_t, std::vector<std::pair<uint32_t, uint32_t>>> classes;
  for (const CellBinding &c : d->cells) {
    if (c.col >= NR) throw std::runtime_error("a bound cell must be a routed wire");
    if (taken[(size_t)c.col * n + c.row]++) throw std::runtime_error("wire cell bound twice");
    classes[d->find(c.var)].push_back({c.row, c.col});
  }
  for (auto &kv : classes) {
    auto &cells = kv.second;
    for (size_t k = 0; k < cells.size(); k++) {
      auto &cur = cells[k];
      auto &nxt = cells[(k + 1) % cells.size()];
      sig[(size_t)cur.second * n + cur.first] = f_mul(D.k_is[nxt.second], sub[nxt.first]);
    }
  }
  data->impl_ = std::move(b->d);
  return data;
}

lcp2_circuit_desc CircuitDescription::c_desc() const {
  lcp2_circuit_desc d{};
  d.params = params;
  d.constants_sigmas = constants_sigmas.data(); d.constants_sigmas_mem = LCP2_MEM_HOST;
  d.k_is = k_is.data(); d.num_selectors = num_selectors; d.num_gates = (uint32_t)gates.size(); d.gates = gates.data();
  d.code = code.data(); d.code_words = code.size(); d.imm = imm.data(); d.num_imm = imm.size();
  d.num_public_inputs = num_public_inputs; d.num_regs = num_regs;
  return d;
}

// ------------------------------------------------------------------ witness generation
namespace {
struct Values {
  const CircuitData::Impl *d;
  F *val;
  uint32_t *stamp;
  uint32_t epoch;
  static constexpr uint32_t CLAIM = 0x80000000u;
  explicit Values(const CircuitData::Impl *dd) : d(dd) {
    if (dd->value_store.size() != dd->parent.size()) { dd->value_store.assign(dd->parent.size(), 0); dd->stamp_store.assign(dd->parent.size(), 0); dd->epoch = 0; }
    if (++dd->epoch >= CLAIM) { std::fill(dd->stamp_store.begin(), dd->stamp_store.end(), 0u); dd->epoch = 1; }  // once in 2^31 proofs
    val = dd->value_store.data(); stamp = dd->stamp_store.data(); epoch = dd->epoch;
  }
  // The generators of one phase may run on several threads (run_host_phase, LCP2_HOST_LANES > 1).  A slot is empty while its stamp
  // is not this proof's epoch, claimed (one writer is storing the value) with epoch | CLAIM, published with epoch; the claim is a
  // compare-and-swap, so exactly one generator writes val[r] and every other one - on any lane - compares with the published value:
  // two generators that disagree about a connected class are always reported, never a torn or lost write.
  bool known(uint32_t r) const { return __atomic_load_n(&stamp[r], __ATOMIC_ACQUIRE) == epoch; }
  void set(uint32_t var, F v, const char *what) {
    uint32_t r = d->find(var);
    uint32_t seen = __atomic_load_n(&stamp[r], __ATOMIC_ACQUIRE);
    while ((seen & ~CLAIM) != epoch) {  // empty (a stamp of an earlier proof): try to claim it
      if (__atomic_compare_exchange_n(&stamp[r], &seen, epoch | CLAIM, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
        val[r] = v;
        __atomic_store_n(&stamp[r], epoch, __ATOMIC_RELEASE);
        return;
      }
    }
    while (__atomic_load_n(&stamp[r], __ATOMIC_ACQUIRE) != epoch) {}  // another lane is between its claim and its store: a few cycles
    if (val[r] != v)
      throw UnsatisfiedError(std::string("witness conflict on a connected target (") + what + "): " + std::to_string(val[r]) + " vs " + std::to_string(v));
  }
  F get(uint32_t var, const char *what) const {
    uint32_t r = d->find(var);
    if (!known(r)) throw UnsatisfiedError(std::string("target has no value: ") + what);
    return val[r];
  }
};

inline void put_bits(std::vector<uint64_t> &wires, uint64_t n, uint32_t row, uint32_t base, uint32_t x) {
  for (int i = 0; i < 32; i++) wires[(size_t)(base + i) * n + row] = (x >> i) & 1;
}
inline uint32_t as_u32(F v, const char *what) {
  if (v > 0xFFFFFFFFull) throw UnsatisfiedError(std::string("value does not fit a U32Target: ") + what);
  return (uint32_t)v;
}

// fills the 310 rows of one two_to_one_sha256 and sets its internal / output variables
void eval_sha(const Op &op, Values &V, std::vector<uint64_t> &wires, uint64_t n) {
  auto cell = [&](uint32_t row, uint32_t col) -> uint64_t & { return wires[(size_t)col * n + row]; };
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = as_u32(V.get(op.in[i], "sha256 message word"), "sha256 message word");
  size_t iv = 0;  // index into op.internal
  uint32_t row = op.first_row;
  for (int t = 16; t < 64; t++, row++) {
    uint32_t w2 = w[t - 2], w15 = w[t - 15];
    uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3), s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
    uint64_t sum = (uint64_t)s1 + w[t - 7] + s0 + w[t - 16];
    w[t] = (uint32_t)sum;
    V.set(op.internal[iv++], w[t], "sha256 schedule word");
    cell(row, 0) = w2; cell(row, 1) = w[t - 7]; cell(row, 2) = w15; cell(row, 3) = w[t - 16]; cell(row, 4) = w[t];
    put_bits(wires, n, row, 8, w2); put_bits(wires, n, row, 40, w15);
    cell(row, 104) = (sum >> 32) & 1; cell(row, 105) = (sum >> 33) & 1;
  }
  uint32_t wpad[64] = {0};
  wpad[0] = 0x80000000u; wpad[15] = 512;
  for (int t = 16; t < 64; t++) {
    uint32_t s0 = rotr(wpad[t - 15], 7) ^ rotr(wpad[t - 15], 18) ^ (wpad[t - 15] >> 3);
    uint32_t s1 = rotr(wpad[t - 2], 17) ^ rotr(wpad[t - 2], 19) ^ (wpad[t - 2] >> 10);
    wpad[t] = wpad[t - 16] + s0 + wpad[t - 7] + s1;
  }
  uint32_t chain[8], st[8];
  for (int i = 0; i < 8; i++) chain[i] = st[i] = SHA_IV[i];
  for (int c = 0; c < 2; c++) {
    for (int t = 0; t < 64; t++) {
      uint32_t a = st[0], bb = st[1], cc = st[2], dd = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
      uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
      uint64_t kw = c == 0 ? (uint64_t)SHA_K[t] : (uint64_t)SHA_K[t] + wpad[t];
      uint32_t wv = c == 0 ? w[t] : 0;
      uint64_t sum1 = (uint64_t)h + S1 + ch + kw + wv;
      uint32_t t1 = (uint32_t)sum1;
      uint64_t sume = (uint64_t)dd + t1;
      uint32_t e_new = (uint32_t)sume;
      uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & bb) ^ (a & cc) ^ (bb & cc);
      uint64_t suma = (uint64_t)t1 + S0 + mj;
      uint32_t a_new = (uint32_t)suma;
      V.set(op.internal[iv++], t1, "sha256 t1"); V.set(op.internal[iv++], a_new, "sha256 a"); V.set(op.internal[iv++], e_new, "sha256 e");
      // round E row
      cell(row, 0) = e; cell(row, 1) = f; cell(row, 2) = g; cell(row, 3) = h; cell(row, 4) = dd; cell(row, 5) = wv; cell(row, 6) = e_new; cell(row, 7) = t1;
      put_bits(wires, n, row, 8, e); put_bits(wires, n, row, 40, f); put_bits(wires, n, row, 72, g);
      uint64_t k1 = sum1 >> 32;
      cell(row, 104) = k1 & 1; cell(row, 105) = (k1 >> 1) & 1; cell(row, 106) = (k1 >> 2) & 1; cell(row, 107) = sume >> 32;
      row++;
      // round A row
      cell(row, 0) = a; cell(row, 1) = bb; cell(row, 2) = cc; cell(row, 3) = t1; cell(row, 4) = a_new;
      put_bits(wires, n, row, 8, a); put_bits(wires, n, row, 40, bb); put_bits(wires, n, row, 72, cc);
      uint64_t k2 = suma >> 32;
      cell(row, 104) = k2 & 1; cell(row, 105) = (k2 >> 1) & 1;
      row++;
      st[7] = g; st[6] = f; st[5] = e; st[4] = e_new; st[3] = cc; st[2] = bb; st[1] = a; st[0] = a_new;
    }
    for (int i = 0; i < 8; i++) {
      if (i % SHA_ADD_OPS == 0 && i) row++;
      int j = i % SHA_ADD_OPS;
      uint64_t sum = (uint64_t)chain[i] + st[i];
      uint32_t out = (uint32_t)sum;
      V.set(op.internal[iv++], out, "sha256 chaining value");
      cell(row, 3 * j) = chain[i]; cell(row, 3 * j + 1) = st[i]; cell(row, 3 * j + 2) = out;
      put_bits(wires, n, row, 9 + 33 * j, out);
      cell(row, 9 + 33 * j + 32) = sum >> 32;
      chain[i] = out;
    }
    row++;
    for (int i = 0; i < 8; i++) st[i] = chain[i];
  }
}
}  // namespace

// PoseidonGenerator: the 12 outputs and every internal wire of the row (S-box inputs, deltas); `raw` collects the cells
// of the row that are not bound to a target
static void eval_poseidon(const Op &op, Values &V, std::vector<lcp2_cell> &raw) {
  F in[12], row[POS_GATE_WIRES];
  for (int i = 0; i < 12; i++) in[i] = V.get(op.in[i], "poseidon input");
  const F swap = V.get(op.x, "poseidon swap flag");
  if (swap > 1) throw UnsatisfiedError("poseidon swap flag is not boolean");
  poseidon_gate_row(in, swap == 1, row);
  for (int i = 0; i < 12; i++) V.set(op.internal[i], row[POS_WIRE_OUTPUT + i], "poseidon output");
  for (uint32_t c = POS_WIRE_DELTA; c < POS_GATE_WIRES; c++) raw.push_back(lcp2_cell{op.first_row, c, row[c]});
}

// The same generator when the rows are filled on the device (generate_witness_gpu): only the 12 outputs are computed here, the
// row is queued as (row, inputs, swap) for lcp2_poseidon_gate_rows
static void eval_poseidon_outputs(const Op &op, Values &V, std::vector<lcp2_poseidon_row> &rows) {
  lcp2_poseidon_row job{};
  F out[12];
  for (int i = 0; i < 12; i++) job.in[i] = V.get(op.in[i], "poseidon input");
  const F swap = V.get(op.x, "poseidon swap flag");
  if (swap > 1) throw UnsatisfiedError("poseidon swap flag is not boolean");
  job.row = op.first_row;
  job.swap = (uint32_t)swap;
  poseidon_gate_outputs(job.in, swap == 1, out);
  for (int i = 0; i < 12; i++) V.set(op.internal[i], out[i], "poseidon output");
  rows.push_back(job);
}

// split_le generator: the low c0 bits of x; a value that does not fit cannot satisfy the recomposition constraint
static void eval_bits(const Op &op, Values &V) {
  const F x = V.get(op.x, "split_le input");
  const unsigned nbits = (unsigned)op.c0;
  if (nbits < 64 && (x >> nbits) != 0) throw UnsatisfiedError("split_le: value does not fit in " + std::to_string(nbits) + " bits (range check fails)");
  for (unsigned i = 0; i < nbits; i++) V.set(op.internal[i], (x >> i) & 1, "split_le bit");
}

static inline F f_inv(F x) { return f_pow(x, GOLDILOCKS_P - 2); }

// every generator that runs on the host (everything but SHA): is it ready, and run it
static bool host_op_ready(const Op &op, const Values &V) {
  auto has = [&](uint32_t v) { return V.known(V.d->find(v)); };
  switch (op.kind) {
    case Op::ARITH: return has(op.x) && has(op.y) && has(op.z);
    case Op::BITS: case Op::INV: case Op::SPLIT32: return has(op.x);
    case Op::HINT: { for (uint32_t v : op.hint_in) if (!has(v)) return false; return true; }
    case Op::EXT_INV: return has(op.x) && has(op.y);
    case Op::POSEIDON: { if (!has(op.x)) return false; for (int i = 0; i < 12; i++) if (!has(op.in[i])) return false; return true; }
    case Op::SHA: { for (uint32_t v : op.in) if (!has(v)) return false; return true; }
    default: return true;
  }
}
static void host_op_eval(const Op &op, Values &V, std::vector<lcp2_cell> &raw) {
  switch (op.kind) {
    case Op::CONST: V.set(op.out, op.c0, "constant"); break;
    case Op::ARITH: {
      F x = V.get(op.x, "arithmetic input"), y = V.get(op.y, "arithmetic input"), z = V.get(op.z, "arithmetic input");
      V.set(op.out, f_add(f_mul(f_mul(x, y), op.c0), f_mul(z, op.c1)), "arithmetic output");
      break;
    }
    case Op::BITS: eval_bits(op, V); break;
    case Op::POSEIDON: eval_poseidon(op, V, raw); break;
    case Op::INV: { F x = V.get(op.x, "inverse input"); V.set(op.out, x ? f_inv(x) : 0, "inverse"); break; }
    case Op::EXT_INV: {  // 1 / (a + b X) = (a - b X) / (a^2 - 7 b^2)
      F a = V.get(op.x, "extension inverse input"), bb = V.get(op.y, "extension inverse input");
      F norm = f_add(f_mul(a, a), GOLDILOCKS_P - f_mul(7, f_mul(bb, bb)));
      if (norm >= GOLDILOCKS_P) norm -= GOLDILOCKS_P;
      F ni = norm ? f_inv(norm) : 0;
      V.set(op.internal[0], f_mul(a, ni), "extension inverse");
      V.set(op.internal[1], f_mul(bb ? GOLDILOCKS_P - bb : 0, ni), "extension inverse");
      break;
    }
    case Op::SPLIT32: { F x = V.get(op.x, "split input"); V.set(op.internal[0], x & 0xFFFFFFFFull, "low half"); V.set(op.internal[1], x >> 32, "high half"); break; }
    case Op::HINT: {
      std::vector<F> in, out(op.internal.size(), 0);
      for (uint32_t v : op.hint_in) in.push_back(V.get(v, "hint input"));
      op.hint_fn(in, out);
      for (size_t i = 0; i < out.size(); i++) V.set(op.internal[i], out[i] % GOLDILOCKS_P, "hint output");
      break;
    }
    case Op::SHA: break;  // the caller's
  }
}

void CircuitData::generate_witness(const PartialWitness &pw, std::vector<uint64_t> &wires, std::vector<F> &public_inputs) const {
  const Impl *d = impl_.get();
  const uint64_t n = 1ull << desc_.params.degree_bits;
  wires.assign((size_t)desc_.params.num_wires * n, 0);
  Values V(d);
  for (auto &e : pw.entries()) V.set(e.first, e.second, "PartialWitness");
  // plonky2 runs its generators from a worklist; here the steps are re-scanned until none is left waiting
  // (gadgets may be wired after they are built: ssz_sync_committee connects the leaves of an existing tree)
  std::vector<lcp2_cell> raw;
  std::vector<const Op *> pending;
  for (const Op &op : d->ops) pending.push_back(&op);
  while (!pending.empty()) {
    std::vector<const Op *> waiting;
    for (const Op *opp : pending) {
      const Op &op = *opp;
      if (!host_op_ready(op, V)) { waiting.push_back(opp); continue; }
      if (op.kind == Op::SHA) eval_sha(op, V, wires, n);
      else host_op_eval(op, V, raw);
    }
    if (waiting.size() == pending.size()) throw UnsatisfiedError("a generator is waiting for a target that is never set");
    pending.swap(waiting);
  }
  for (const lcp2_cell &c : raw) wires[(size_t)c.col * n + c.row] = c.value;
  for (const CellBinding &c : d->cells) wires[(size_t)c.col * n + c.row] = V.get(c.var, "wire cell");
  public_inputs.clear();
  for (uint32_t v : d->public_inputs) public_inputs.push_back(V.get(v, "public input"));
}

// ------------------------------------------------------------------ prove / verify through the C ABI
CircuitData::~CircuitData() {
  if (impl_) {
    if (impl_->cell_buf_pinned) lcp2_host_unregister(impl_->ctx, impl_->cell_buf.data());
    if (impl_->d_wires) lcp2_buffer_free(impl_->ctx, impl_->d_wires);
    if (impl_->gpu) lcp2_circuit_destroy(impl_->gpu);
    if (impl_->verifier) lcp2_circuit_destroy(impl_->verifier);
  }
}

void CircuitData::verifier_only_data(uint64_t digest[4], std::vector<uint64_t> &cap) const {
  if (!impl_->gpu) throw std::runtime_error("verifier_only_data needs attach_gpu()");
  cap.assign((size_t)4 << desc_.params.cap_height, 0);
  if (lcp2_circuit_digest(impl_->gpu, digest, cap.data()) != LCP2_OK) throw std::runtime_error("lcp2_circuit_digest failed");
}

void CircuitData::attach_gpu(lcp2_ctx *ctx) {
  lcp2_circuit_desc cd = desc_.c_desc();
  int rc = lcp2_circuit_create(ctx, &cd, &impl_->gpu);
  if (rc != LCP2_OK) throw std::runtime_error(std::string("lcp2_circuit_create: ") + lcp2_status_str(rc) + " (" + lcp2_last_error(ctx) + ")");
  impl_->ctx = ctx;
  const size_t bytes = (size_t)desc_.params.num_wires * ((size_t)1 << desc_.params.degree_bits) * 8;
  if (lcp2_buffer_alloc(ctx, bytes, &impl_->d_wires) != LCP2_OK || lcp2_buffer_zero(ctx, impl_->d_wires, bytes) != LCP2_OK)
    throw std::runtime_error(std::string("witness buffer: ") + lcp2_last_error(ctx));
}

// ---- One host phase of generate_witness_gpu: every host generator that is, or becomes, ready.
// The recursive verifier puts thousands of PoseidonGate rows into the light-client circuit - 3 152 of them the sponge over the
// inner proof's public inputs, sequential by construction, the rest Merkle paths and the Challenger, short chains that are
// independent of each other - and one permutation costs the host 4-6 us.  The generators are therefore spread over LANES (threads):
// PoseidonGate generators that feed one another directly form a chain and a chain stays on one lane (largest chains first, each
// to the least loaded lane), everything else (arithmetic, bit splits: cheap) shares the last lane.  Every lane sweeps its own list
// in creation order, runs what is ready and keeps what is not, as the single-threaded worklist does; a phase ends when every lane
// has either emptied its list or swept it without anything having happened anywhere in the meantime.
static HostLanes plan_host_lanes(const CircuitData::Impl *d, unsigned max_lanes) {
  HostLanes L;
  const size_t nops = d->ops.size();
  L.lane_of_op.assign(nops, 0);
  std::vector<uint32_t> pos;  // indices of the PoseidonGate generators
  for (size_t i = 0; i < nops; i++) if (d->ops[i].kind == Op::POSEIDON) pos.push_back((uint32_t)i);
  if (max_lanes < 2 || pos.size() < 512) return L;  // not worth a thread
  // chains: union-find over "an input of q is an output of p"
  std::map<uint32_t, uint32_t> producer;  // variable class -> position in pos
  for (uint32_t k = 0; k < pos.size(); k++)
    for (uint32_t v : d->ops[pos[k]].internal) producer[d->find(v)] = k;
  std::vector<uint32_t> parent(pos.size());
  for (uint32_t k = 0; k < pos.size(); k++) parent[k] = k;
  std::function<uint32_t(uint32_t)> find = [&](uint32_t x) { while (parent[x] != x) x = parent[x] = parent[parent[x]]; return x; };
  for (uint32_t k = 0; k < pos.size(); k++)
    for (int i = 0; i < 12; i++) {
      auto it = producer.find(d->find(d->ops[pos[k]].in[i]));
      if (it != producer.end()) parent[find(k)] = find(it->second);
    }
  std::map<uint32_t, std::vector<uint32_t>> chains;
  for (uint32_t k = 0; k < pos.size(); k++) chains[find(k)].push_back(k);
  std::vector<const std::vector<uint32_t> *> by_size;
  for (auto &c : chains) by_size.push_back(&c.second);
  std::sort(by_size.begin(), by_size.end(), [](const std::vector<uint32_t> *a, const std::vector<uint32_t> *b) { return a->size() != b->size() ? a->size() > b->size() : (*a)[0] < (*b)[0]; });
  L.lanes = max_lanes;
  std::vector<size_t> load(L.lanes, 0);
  load[L.lanes - 1] = (nops - pos.size()) / 40;  // the cheap generators: about 1/40 of a permutation each
  for (const auto *c : by_size) {
    const unsigned lane = (unsigned)(std::min_element(load.begin(), load.end()) - load.begin());
    for (uint32_t k : *c) L.lane_of_op[pos[k]] = (uint8_t)lane;
    load[lane] += c->size();
  }
  for (size_t i = 0; i < nops; i++) if (d->ops[i].kind != Op::POSEIDON) L.lane_of_op[i] = (uint8_t)(L.lanes - 1);
  return L;
}

struct HostPhaseStats { size_t sweeps = 0, visits = 0; };
// host_ops: the generators still waiting (in creation order); on return those that are waiting for something this phase cannot
// produce (a SHA-256 digest of the next device batch - or nothing at all: the caller tells these apart at the end)
static void run_host_phase(const CircuitData::Impl *d, const HostLanes &L, std::vector<const Op *> &host_ops, Values &V, std::vector<lcp2_cell> &cells,
                           std::vector<lcp2_poseidon_row> &pos_rows, HostPhaseStats &st) {
  auto sweep = [&](std::vector<const Op *> &mine, std::vector<lcp2_cell> &my_cells, std::vector<lcp2_poseidon_row> &my_rows, std::atomic<uint64_t> *epoch) {
    bool progress = false;
    std::vector<const Op *> waiting;
    for (const Op *op : mine) {
      if (!host_op_ready(*op, V)) { waiting.push_back(op); continue; }
      if (op->kind == Op::POSEIDON) eval_poseidon_outputs(*op, V, my_rows);
      else host_op_eval(*op, V, my_cells);
      if (epoch) epoch->fetch_add(1, std::memory_order_release);
      progress = true;
    }
    __atomic_fetch_add(&st.visits, mine.size(), __ATOMIC_RELAXED);  // shared by the lanes
    __atomic_fetch_add(&st.sweeps, (size_t)1, __ATOMIC_RELAXED);
    mine.swap(waiting);
    return progress;
  };
  size_t npos = 0;
  for (const Op *op : host_ops) npos += op->kind == Op::POSEIDON;
  if (L.lanes < 2 || npos < 512) {  // the single-threaded worklist
    while (!host_ops.empty() && sweep(host_ops, cells, pos_rows, nullptr)) {}
    return;
  }
  const unsigned K = L.lanes;
  std::vector<std::vector<const Op *>> mine(K);
  std::vector<std::vector<lcp2_cell>> lane_cells(K);
  std::vector<std::vector<lcp2_poseidon_row>> lane_rows(K);
  for (const Op *op : host_ops) mine[L.lane_of_op[(size_t)(op - d->ops.data())]].push_back(op);
  std::atomic<uint64_t> epoch{0};
  std::atomic<int> active{(int)K}, idle{0};
  std::atomic<bool> stop{false};
  std::vector<std::exception_ptr> failure(K);
  auto lane_main = [&](unsigned k) {
    try {
      unsigned backoff = 0;
      while (!stop.load(std::memory_order_acquire) && !mine[k].empty()) {
        const uint64_t e0 = epoch.load(std::memory_order_acquire);
        if (sweep(mine[k], lane_cells[k], lane_rows[k], &epoch)) { backoff = 0; continue; }
        if (epoch.load(std::memory_order_acquire) != e0) {
          // other lanes are producing values, none of them for this lane yet (the verifier's Merkle paths wait for query indices that
          // wait for the sponge): look again after a pause that doubles up to 256 us, instead of re-reading the whole list (and the
          // cache lines the busy lanes write) for every value somebody else produces
          const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(1u << backoff);
          while (!stop.load(std::memory_order_acquire) && std::chrono::steady_clock::now() < until) std::this_thread::yield();
          if (backoff < 8) backoff++;
          continue;
        }
        // nothing happened anywhere while this lane looked at all it has: wait for news, or find that every lane is in this state
        idle.fetch_add(1, std::memory_order_acq_rel);
        while (!stop.load(std::memory_order_acquire) && epoch.load(std::memory_order_acquire) == e0) {
          if (idle.load(std::memory_order_acquire) >= active.load(std::memory_order_acquire)) stop.store(true, std::memory_order_release);
          else std::this_thread::yield();
        }
        idle.fetch_sub(1, std::memory_order_acq_rel);
      }
    } catch (...) {
      failure[k] = std::current_exception();
      stop.store(true, std::memory_order_release);
    }
    active.fetch_sub(1, std::memory_order_acq_rel);
  };
  std::vector<std::thread> threads;
  for (unsigned k = 1; k < K; k++) threads.emplace_back(lane_main, k);
  lane_main(0);
  for (auto &t : threads) t.join();
  for (unsigned k = 0; k < K; k++) if (failure[k]) std::rethrow_exception(failure[k]);
  host_ops.clear();
  for (unsigned k = 0; k < K; k++) {
    host_ops.insert(host_ops.end(), mine[k].begin(), mine[k].end());
    cells.insert(cells.end(), lane_cells[k].begin(), lane_cells[k].end());
    pos_rows.insert(pos_rows.end(), lane_rows[k].begin(), lane_rows[k].end());
  }
  std::sort(host_ops.begin(), host_ops.end());  // creation order again
}

// generate_partial_witness with the SHA-256 generators on the device.  Host and device alternate until every
// generator has run: the host evaluates constants / arithmetic from its worklist, then every two_to_one_sha256
// whose message is known (directly, or as the digest of another hash of the same batch) goes to the GPU as one
// levelled batch; the digests come back for the copy-constraint conflict check and for generators that use them.
void CircuitData::generate_witness_gpu(const PartialWitness &pw, std::vector<F> &public_inputs) {
  Impl *d = impl_.get();
  if (!d->gpu) throw std::runtime_error("generate_witness_gpu needs attach_gpu()");
  const uint64_t n = 1ull << desc_.params.degree_bits;
  Values V(d);
  for (auto &e : pw.entries()) V.set(e.first, e.second, "PartialWitness");
  std::vector<const Op *> host_ops, sha_ops;
  for (const Op &op : d->ops) (op.kind == Op::SHA ? sha_ops : host_ops).push_back(&op);
  std::vector<lcp2_cell> cells;               // what the host writes cell by cell: the bound cells of the non-SHA, non-Poseidon rows
  std::vector<lcp2_poseidon_row> pos_rows;    // PoseidonGate rows: generated on the device from (row, inputs, swap)
  const bool prof = getenv("LCP2_PROF") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  double t_host = 0;
  if (d->host_lanes.lane_of_op.size() != d->ops.size()) {
    // One lane unless LCP2_HOST_LANES asks for more.  Measured on the light-client circuit with the recursive verifier (EPYC 9575F,
    // profiles/r03_host_witness.md): one lane 9.8 ms, eight lanes 8.4-9.5 ms with outliers of 17 ms - everything the verifier does
    // waits for the sponge over the inner public inputs (it is observed into the transcript first), so the lanes have only the 40 %
    // after it to share, and thread start-up and hand-offs eat most of that.
    unsigned want = 1;
    if (const char *e = getenv("LCP2_HOST_LANES")) want = (unsigned)atoi(e);
    d->host_lanes = plan_host_lanes(d, std::max(1u, std::min(want, 8u)));
  }
  HostPhaseStats st;
  // the first proof of a circuit records its device batches and the host's cell list (GpuWitnessPlan); later proofs replay them
  bool recording = !d->gpu_plan.ready;
  GpuWitnessPlan recorded;
  size_t n_batches = 0;
  const auto t_begin = now();
  while (true) {
    const auto t0 = now();
    run_host_phase(d, d->host_lanes, host_ops, V, cells, pos_rows, st);
    t_host += ms(t0, now());
    if (sha_ops.empty()) break;
    const size_t batch_no = n_batches++;
    if (d->gpu_plan.ready && batch_no < d->gpu_plan.batches.size()) {
      // the batch as planned at the first proof: only the message words are fetched
      const ShaBatchPlan &B = d->gpu_plan.batches[batch_no];
      bool applies = B.ops_left.size() < sha_ops.size();
      std::vector<uint32_t> words(B.word_vars.size());
      for (size_t k = 0; k < words.size() && applies; k++) {
        if (!V.known(B.word_vars[k])) applies = false;
        else words[k] = as_u32(V.val[B.word_vars[k]], "sha256 message word");
      }
      if (applies) {
        std::vector<uint32_t> digests(B.jobs.size() * 8);
        int rc = lcp2_sha256_witness(d->ctx, B.jobs.data(), B.jobs.size(), B.level_start.data(), (uint32_t)B.level_start.size() - 1, words.data(), words.size(),
                                     (uint64_t *)d->d_wires, n, digests.data());
        if (rc != LCP2_OK) throw std::runtime_error(std::string("lcp2_sha256_witness: ") + lcp2_status_str(rc) + " (" + lcp2_last_error(d->ctx) + ")");
        for (size_t k = 0; k < B.jobs.size(); k++)
          for (int w = 0; w < 8; w++) V.set(B.out[k][w], digests[k * 8 + w], "sha256 digest");
        sha_ops.clear();
        for (uint32_t k : B.ops_left) sha_ops.push_back(&d->ops[k]);
        continue;
      }
      d->gpu_plan = GpuWitnessPlan();  // another set of witness targets than at the first proof: plan afresh, from this batch on
      recording = false;
    }
    // plan one device batch
    std::map<uint32_t, uint32_t> produced;  // variable class -> slot * 8 + word
    std::vector<lcp2_sha_job> jobs;
    std::vector<uint32_t> level, words, word_vars;
    std::vector<const Op *> batch, later;
    for (const Op *op : sha_ops) {
      lcp2_sha_job job{};
      job.first_row = op->first_row;
      uint32_t lvl = 0;
      bool ok = true;
      size_t words_mark = words.size();
      for (int i = 0; i < 16 && ok; i++) {
        uint32_t r = d->find(op->in[i]);
        if (V.known(r)) { job.in_src[i] = (int32_t)words.size(); words.push_back(as_u32(V.val[r], "sha256 message word")); word_vars.push_back(r); }
        else {
          auto it = produced.find(r);
          if (it == produced.end()) ok = false;
          else { job.in_src[i] = ~(int32_t)it->second; lvl = std::max(lvl, level[it->second >> 3] + 1); }
        }
      }
      if (!ok) { words.resize(words_mark); word_vars.resize(words_mark); later.push_back(op); continue; }
      uint32_t slot = (uint32_t)jobs.size();
      for (int w = 0; w < 8; w++) produced[d->find(op->out8[w])] = slot * 8 + w;
      jobs.push_back(job); level.push_back(lvl); batch.push_back(op);
    }
    if (jobs.empty()) throw UnsatisfiedError("a generator is waiting for a target that is never set");
    // order by level, remap the digest references
    uint32_t nlev = 0;
    for (uint32_t l : level) nlev = std::max(nlev, l + 1);
    std::vector<uint32_t> order(jobs.size()), newslot(jobs.size()), level_start(nlev + 1, 0);
    for (uint32_t l : level) level_start[l + 1]++;
    for (uint32_t l = 0; l < nlev; l++) level_start[l + 1] += level_start[l];
    {
      std::vector<uint32_t> fill(level_start.begin(), level_start.end() - 1);
      for (uint32_t s = 0; s < jobs.size(); s++) { newslot[s] = fill[level[s]]++; order[newslot[s]] = s; }
    }
    std::vector<lcp2_sha_job> sorted(jobs.size());
    for (uint32_t s = 0; s < jobs.size(); s++) {
      lcp2_sha_job j = jobs[s];
      for (int i = 0; i < 16; i++)
        if (j.in_src[i] < 0) { uint32_t ref = (uint32_t)~j.in_src[i]; j.in_src[i] = ~(int32_t)(newslot[ref >> 3] * 8 + (ref & 7)); }
      sorted[newslot[s]] = j;
    }
    std::vector<uint32_t> digests(jobs.size() * 8);
    int rc = lcp2_sha256_witness(d->ctx, sorted.data(), sorted.size(), level_start.data(), nlev, words.data(), words.size(),
                                 (uint64_t *)d->d_wires, n, digests.data());
    if (rc != LCP2_OK) throw std::runtime_error(std::string("lcp2_sha256_witness: ") + lcp2_status_str(rc) + " (" + lcp2_last_error(d->ctx) + ")");
    for (uint32_t s = 0; s < batch.size(); s++)
      for (int w = 0; w < 8; w++) V.set(batch[s]->out8[w], digests[(size_t)newslot[s] * 8 + w], "sha256 digest");
    if (recording) {
      ShaBatchPlan B;
      B.jobs = sorted; B.word_vars = word_vars; B.level_start = level_start;
      for (const Op *op : later) B.ops_left.push_back((uint32_t)(op - d->ops.data()));
      B.out.resize(jobs.size());
      for (uint32_t s = 0; s < batch.size(); s++)
        for (int w = 0; w < 8; w++) B.out[newslot[s]][w] = d->find(batch[s]->out8[w]);
      recorded.batches.push_back(std::move(B));
    }
    sha_ops.swap(later);
  }
  if (!host_ops.empty()) throw UnsatisfiedError("a generator is waiting for a target that is never set");
  const auto t_rows = now();
  // the PoseidonGate rows: every wire of a row from its inputs (the values bound to its input / output / swap cells are the ones
  // the jobs were made from: a conflicting connection has already thrown in V.set)
  int rc = lcp2_poseidon_gate_rows(d->ctx, pos_rows.data(), pos_rows.size(), (uint64_t *)d->d_wires, n);
  if (rc != LCP2_OK) throw std::runtime_error(std::string("lcp2_poseidon_gate_rows: ") + lcp2_status_str(rc) + " (" + lcp2_last_error(d->ctx) + ")");
  // the cells of the remaining rows (the list is made once: most of a light-client circuit's bindings sit on SHA-256 rows)
  if (!d->gpu_plan.ready) {
    GpuWitnessPlan fresh;
    for (const CellBinding &c : d->cells) {
      uint32_t g = d->gate_of_row[c.row];
      if (g == G_SHA_ADD || g == G_SHA_ROUND_A || g == G_SHA_ROUND_E || g == G_SHA_SCHED || g == G_POSEIDON) continue;
      fresh.host_cells.push_back(CellBinding{c.row, c.col, d->find(c.var)});
    }
    if (recording) { fresh.batches = std::move(recorded.batches); fresh.ready = true; }
    d->gpu_plan = std::move(fresh);  // not ready if this proof left the recorded plan half way: the next one records again
  }
  // the planned cells: rows and columns were filled when the plan was made, only the values change from proof to proof
  std::vector<lcp2_cell> &buf = d->cell_buf;
  if (buf.size() != d->gpu_plan.host_cells.size()) {
    if (d->cell_buf_pinned) { lcp2_host_unregister(d->ctx, buf.data()); d->cell_buf_pinned = false; }
    buf.resize(d->gpu_plan.host_cells.size());
    for (size_t k = 0; k < buf.size(); k++) { buf[k].row = d->gpu_plan.host_cells[k].row; buf[k].col = d->gpu_plan.host_cells[k].col; }
    if (!buf.empty()) d->cell_buf_pinned = lcp2_host_register(d->ctx, buf.data(), buf.size() * sizeof(lcp2_cell)) == LCP2_OK;  // (unpinned it still works)
  }
  for (size_t k = 0; k < buf.size(); k++) {
    const uint32_t var = d->gpu_plan.host_cells[k].var;
    if (!V.known(var)) throw UnsatisfiedError("target has no value: wire cell");
    buf[k].value = V.val[var];
  }
  const auto t_scatter = now();
  rc = lcp2_scatter_cells(d->ctx, buf.data(), buf.size(), (uint64_t *)d->d_wires, n);
  if (rc == LCP2_OK) rc = lcp2_scatter_cells(d->ctx, cells.data(), cells.size(), (uint64_t *)d->d_wires, n);  // what the generators of this proof wrote cell by cell
  if (rc != LCP2_OK) throw std::runtime_error(std::string("lcp2_scatter_cells: ") + lcp2_status_str(rc));
  if (prof)
    printf("  witness generation %.2f ms: host generators %.2f ms on %u lane(s) (%zu sweeps, %zu visits of %zu ops, %zu PoseidonGate rows), SHA-256 batches %.2f ms, "
           "PoseidonGate rows on the device + cell list %.2f ms, scatter of %zu cells %.2f ms\n",
           ms(t_begin, now()), t_host, d->host_lanes.lanes, st.sweeps, st.visits, d->ops.size(), pos_rows.size(), ms(t_begin, t_rows) - t_host, ms(t_rows, t_scatter), cells.size() + buf.size(), ms(t_scatter, now()));
  public_inputs.clear();
  for (uint32_t v : d->public_inputs) public_inputs.push_back(V.get(v, "public input"));
}

void CircuitData::read_device_witness(std::vector<uint64_t> &wires) const {
  const size_t words = (size_t)desc_.params.num_wires << desc_.params.degree_bits;
  wires.resize(words);
  if (!impl_->d_wires || lcp2_buffer_read(impl_->ctx, wires.data(), impl_->d_wires, words * 8) != LCP2_OK)
    throw std::runtime_error("read_device_witness failed");
}

ProofWithPublicInputs CircuitData::prove(const PartialWitness &pw) {
  if (!impl_->gpu) throw std::runtime_error("CircuitData::prove needs attach_gpu(): there is no CPU prover in this library");
  ProofWithPublicInputs out;
  generate_witness_gpu(pw, out.public_inputs);  // throws UnsatisfiedError = plonky2's Err; the witness stays in HBM
  out.proof.assign(lcp2_proof_words(&desc_.params), 0);
  int rc = lcp2_prove(impl_->gpu, (const uint64_t *)impl_->d_wires, LCP2_MEM_DEVICE, out.public_inputs.data(), out.public_inputs.size(),
                      out.proof.data(), out.proof.size());
  // the generators above already reject an inconsistent witness; LCP2_E_UNSAT is the device-side check saying the same
  if (rc == LCP2_E_UNSAT) throw UnsatisfiedError(std::string("lcp2_prove: ") + lcp2_last_error(impl_->ctx));
  if (rc != LCP2_OK) throw std::runtime_error(std::string("lcp2_prove: ") + lcp2_status_str(rc) + " (" + lcp2_last_error(impl_->ctx) + ")");
  return out;
}

void CircuitData::verify(const ProofWithPublicInputs &proof) const {
  const lcp2_circuit *c = impl_->gpu ? impl_->gpu : impl_->verifier;
  if (!c) throw std::runtime_error("CircuitData::verify needs the circuit digest: attach_gpu() first");
  int failed = 0;
  int rc = lcp2_verify(c, proof.proof.data(), proof.proof.size(), proof.public_inputs.data(), proof.public_inputs.size(), &failed);
  if (rc == LCP2_E_VERIFY) throw VerifyError("proof rejected (check " + std::to_string(failed) + ")");
  if (rc != LCP2_OK) throw std::runtime_error(std::string("lcp2_verify: ") + lcp2_status_str(rc));
}

}  // namespace lc

// diagnostics for tools (how a circuit's witness generation is made up)
namespace lc {
struct OpStats { size_t kinds[16]; size_t ops, vars, cells, const_cells; };
OpStats op_stats(const CircuitData &data) {
  const CircuitData::Impl *d = data.impl_for_tools();
  OpStats s{};
  s.ops = d->ops.size(); s.vars = d->parent.size(); s.cells = d->cells.size();
  std::vector<uint8_t> is_const(d->parent.size(), 0);
  for (const Op &op : d->ops) { s.kinds[op.kind]++; if (op.kind == Op::CONST) is_const[d->find(op.out)] = 1; }
  for (const CellBinding &c : d->cells) {
    const uint32_t g = d->gate_of_row[c.row];
    if (g == G_SHA_ADD || g == G_SHA_ROUND_A || g == G_SHA_ROUND_E || g == G_SHA_SCHED || g == G_POSEIDON) { s.cells--; continue; }
    s.const_cells += is_const[d->find(c.var)];
  }
  return s;
}
}  // namespace lc
