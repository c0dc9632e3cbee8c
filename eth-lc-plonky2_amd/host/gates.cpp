// Gate set of the own circuit layout and its constraint programs (instruction format: include/lcp2.h).
//
// plonky2_crypto's SHA-256 circuit (U32 add / arithmetic / base-sum gates, ~1400 rows per compression) is not
// visible from the reference, so SHA-256 is laid out with dedicated row types, one row per round half:
//   sha_round_e  e,f,g,h,d,w -> e_new, t1     t1 + 2^32 k = h + S1(e) + Ch(e,f,g) + (K_t [+ W_t]) + w ; e_new + 2^32 c = d + t1
//   sha_round_a  a,b,c,t1    -> a_new         a_new + 2^32 k = t1 + S0(a) + Maj(a,b,c)
//   sha_sched    w2,w7,w15,w16 -> wt          wt + 2^32 k = s1(w2) + w7 + s0(w15) + w16
//   sha_add      3 x (x, y -> out)            out + 2^32 c = x + y, out range-checked by its bits
// Words are single field elements on routed wires; the bit decompositions live in the same row (booleanity +
// recomposition constraints).  Every equation holds modulo 2^32 and every word that leaves a compression is
// range-checked by a decomposition, which pins the values (DESIGN.md section "SHA-256 layout").
// 310 rows per two_to_one_sha256 (179 for the data block, 131 for the constant padding block).
#include "host_internal.hpp"

namespace lc {

const char *gate_name(uint32_t k) {
  static const char *names[G_COUNT] = {"NoopGate", "ConstantGate", "PublicInputGate", "ShaAddGate", "ArithmeticGate",
                                       "ShaRoundAGate", "ShaRoundEGate", "ShaScheduleGate", "PoseidonGate"};
  return k < G_COUNT ? names[k] : "?";
}
const uint32_t GATE_DEGREE[G_COUNT] = {0, 1, 1, 2, 3, 3, 3, 3, 7};

namespace {
enum { OP_ADD = LCP2_OP_ADD, OP_SUB = LCP2_OP_SUB, OP_MUL = LCP2_OP_MUL, OP_EMIT = LCP2_OP_EMIT, OP_XOR = LCP2_OP_XOR, OP_DBLADD = LCP2_OP_DBLADD,
       OP_EMITBOOL = LCP2_OP_EMITBOOL, OP_MULADD = LCP2_OP_MULADD, OP_SBOX = LCP2_OP_SBOX, OP_PMDS = LCP2_OP_PMDS };
enum { K_REG = 0, K_WIRE = 1, K_CONST = 2, K_IMM = 3, K_PI = 4 };
struct Opnd { uint32_t kind, idx; };
inline Opnd R(uint32_t i) { return {K_REG, i}; }
inline Opnd W(uint32_t i) { return {K_WIRE, i}; }
inline Opnd C(uint32_t i) { return {K_CONST, i}; }
inline Opnd PI(uint32_t i) { return {K_PI, i}; }

struct Asm {
  std::vector<uint32_t> &code;
  std::vector<uint64_t> &imm;
  uint32_t max_reg = 0, nconstraints = 0, flags = 0;
  // constraints are collected per index and written out last-to-first (EMIT is a Horner step); a gate that sets
  // LCP2_GATE_EMIT_FORWARD writes its blocks in order instead
  std::vector<std::vector<uint32_t>> blocks;
  std::vector<uint32_t> *cur = nullptr;
  Asm(std::vector<uint32_t> &c, std::vector<uint64_t> &i) : code(c), imm(i) {}
  Opnd IMM(uint64_t v) {
    for (size_t k = 0; k < imm.size(); k++) if (imm[k] == v) return {K_IMM, (uint32_t)k};
    imm.push_back(v);
    return {K_IMM, (uint32_t)imm.size() - 1};
  }
  void begin() { blocks.emplace_back(); cur = &blocks.back(); }
  void op(uint32_t o, uint32_t dst, Opnd a, Opnd b) {
    if (dst + 1 > max_reg) max_reg = dst + 1;
    cur->push_back(o | dst << 8 | a.kind << 16 | b.kind << 20);
    cur->push_back(a.idx | b.idx << 16);
  }
  void add(uint32_t d, Opnd a, Opnd b) { op(OP_ADD, d, a, b); }
  void sub(uint32_t d, Opnd a, Opnd b) { op(OP_SUB, d, a, b); }
  void mul(uint32_t d, Opnd a, Opnd b) { op(OP_MUL, d, a, b); }
  void xor_(uint32_t d, Opnd a, Opnd b) { op(OP_XOR, d, a, b); }        // bits: a ^ b = a + b - 2ab
  void dbladd(uint32_t d, Opnd a, Opnd b) { op(OP_DBLADD, d, a, b); }   // 2a + b: one Horner step of a bit recomposition
  void muladd(uint32_t d, Opnd a, Opnd b) { op(OP_MULADD, d, a, b); }   // r[d] += a * b
  void emit(Opnd a) { cur->push_back(OP_EMIT | a.kind << 16); cur->push_back(a.idx); nconstraints++; }
  void emit_bool(Opnd a) { cur->push_back(OP_EMITBOOL | a.kind << 16); cur->push_back(a.idx); nconstraints++; }
  void sbox(uint32_t d, Opnd a) { op(OP_SBOX, d, a, R(0)); }             // r[d] = a^7
  // r[d .. d+12) = MDS * r[src .. src+12) + the 12 constants (a contiguous block of immediates)
  void pmds(uint32_t d, uint32_t src, const uint64_t constants[12]) {
    const uint32_t base = (uint32_t)imm.size();
    imm.insert(imm.end(), constants, constants + 12);
    if (d + 12 > max_reg) max_reg = d + 12;
    cur->push_back(OP_PMDS | d << 8 | K_REG << 16 | K_IMM << 20);
    cur->push_back(src | base << 16);
  }
  void finish() {
    if (flags & LCP2_GATE_EMIT_FORWARD) { for (auto &b : blocks) code.insert(code.end(), b.begin(), b.end()); return; }
    for (size_t k = blocks.size(); k-- > 0;) code.insert(code.end(), blocks[k].begin(), blocks[k].end());
  }
  // ---- constraint helpers (each is one constraint block)
  void boolean(uint32_t wire) { begin(); emit_bool(W(wire)); }
  void recompose(uint32_t bits, uint32_t word) {  // sum 2^i bit_i - word
    begin();
    dbladd(0, W(bits + 31), W(bits + 30));
    for (int i = 29; i >= 0; i--) dbladd(0, R(0), W(bits + i));
    sub(0, R(0), W(word));
    emit(R(0));
  }
  // r[dst] = x ^ y ^ z for bits (z may be absent: z_wire < 0)
  void xor3(uint32_t dst, uint32_t, int x, int y, int z) {
    xor_(dst, W(x), W(y));
    if (z >= 0) xor_(dst, R(dst), W(z));
  }
};

constexpr uint64_t TWO32 = 1ull << 32;

void prog_noop(Asm &) {}
// plonky2's own gates, constraint for constraint (gates/constant.rs, public_input.rs, arithmetic_base.rs, poseidon.rs [RECALL])
void prog_constant(Asm &a) {
  for (int i = 0; i < 2; i++) { a.begin(); a.sub(0, C(i), W(i)); a.emit(R(0)); }
}
void prog_public_input(Asm &a) {  // wires 0..4 against public_inputs_hash
  for (uint32_t i = 0; i < 4; i++) { a.begin(); a.sub(0, W(i), PI(i)); a.emit(R(0)); }
}
void prog_arithmetic(Asm &a) {
  a.flags |= LCP2_GATE_NATIVE_ARITHMETIC;
  for (int k = 0; k < 20; k++) {
    a.begin();
    a.mul(0, W(4 * k), W(4 * k + 1));
    a.mul(0, R(0), C(0));
    a.mul(1, W(4 * k + 2), C(1));
    a.add(0, R(0), R(1));
    a.sub(0, W(4 * k + 3), R(0));
    a.emit(R(0));
  }
}
// PoseidonGate: 123 constraints in the order of eval_unfiltered, emitted first to last (one forward pass over the rounds):
// swap booleanity, 4 delta equations, `state - sbox_in` for every S-box that has a wire, the 12 outputs.  The partial rounds
// are in the naive form; plonky2's fast-partial-round form is the same polynomial in the wires.
void prog_poseidon(Asm &a) {
  const uint64_t *rc = poseidon_round_constants();
  const uint64_t zeros[12] = {0};
  a.flags |= LCP2_GATE_EMIT_FORWARD | LCP2_GATE_NATIVE_POSEIDON;  // the library checks the native claim at build()
  const uint32_t T = 12;  // scratch register; r0..r11 hold the state
  a.begin(); a.emit_bool(W(POS_WIRE_SWAP));
  for (uint32_t i = 0; i < 4; i++) {
    a.begin();
    a.sub(T, W(POS_WIRE_INPUT + i + 4), W(POS_WIRE_INPUT + i));
    a.mul(T, R(T), W(POS_WIRE_SWAP));
    a.sub(T, R(T), W(POS_WIRE_DELTA + i));
    a.emit(R(T));
  }
  a.begin();
  for (uint32_t i = 0; i < 4; i++) {
    a.add(i, W(POS_WIRE_INPUT + i), W(POS_WIRE_DELTA + i));
    a.add(i, R(i), a.IMM(rc[i]));
    a.sub(i + 4, W(POS_WIRE_INPUT + i + 4), W(POS_WIRE_DELTA + i));
    a.add(i + 4, R(i + 4), a.IMM(rc[i + 4]));
  }
  for (uint32_t i = 8; i < 12; i++) a.add(i, W(POS_WIRE_INPUT + i), a.IMM(rc[i]));
  uint32_t rnd = 0;
  auto next = [&]() { return rnd + 1 < 30 ? rc + 12 * (rnd + 1) : zeros; };
  for (uint32_t r = 0; r < 4; r++, rnd++) {
    for (uint32_t i = 0; i < 12; i++) {
      if (r) {
        const Opnd w = W(pos_wire_full_sbox_0(r, i));
        a.sub(T, R(i), w); a.emit(R(T)); a.begin();
        a.sbox(i, w);
      } else a.sbox(i, R(i));
    }
    a.pmds(0, 0, next());
  }
  for (uint32_t r = 0; r < 22; r++, rnd++) {
    const Opnd w = W(POS_WIRE_PARTIAL + r);
    a.sub(T, R(0), w); a.emit(R(T)); a.begin();
    a.sbox(0, w);
    a.pmds(0, 0, next());
  }
  for (uint32_t r = 0; r < 4; r++, rnd++) {
    for (uint32_t i = 0; i < 12; i++) {
      const Opnd w = W(pos_wire_full_sbox_1(r, i));
      a.sub(T, R(i), w); a.emit(R(T)); a.begin();
      a.sbox(i, w);
    }
    a.pmds(0, 0, next());
  }
  for (uint32_t i = 0; i < 12; i++) { a.sub(T, R(i), W(POS_WIRE_OUTPUT + i)); a.emit(R(T)); a.begin(); }
}
// The four SHA-256 gates have straight-line device forms generated from these very programs (tools/gen/run.sh ->
// csrc/generated_gates.hpp, in this order); the flag is the claim, lcp2_circuit_create checks it.
void prog_sha_add(Asm &a) {
  a.flags |= LCP2_GATE_NATIVE_GENERATED(0);
  for (int j = 0; j < SHA_ADD_OPS; j++) {
    const uint32_t x = 3 * j, y = 3 * j + 1, out = 3 * j + 2, bits = 9 + 33 * j, carry = bits + 32;
    for (int i = 0; i < 32; i++) a.boolean(bits + i);
    a.recompose(bits, out);
    a.begin();
    a.add(0, W(x), W(y));
    a.sub(0, R(0), W(out));
    a.mul(1, W(carry), a.IMM(TWO32));
    a.sub(0, R(0), R(1));
    a.emit(R(0));
    a.boolean(carry);
  }
}
// tail of a sum equation: acc(r1) + extras - out - 2^32 * (c0 + 2 c1 [+ 4 c2])
void carry_tail(Asm &a, uint32_t carry0, int ncarry) {
  a.dbladd(2, W(carry0 + ncarry - 1), W(carry0 + ncarry - 2));
  for (int k = ncarry - 3; k >= 0; k--) a.dbladd(2, R(2), W(carry0 + k));
  a.mul(2, R(2), a.IMM(TWO32));
  a.sub(1, R(1), R(2));
  a.emit(R(1));
}
void prog_sha_round_e(Asm &a) {
  a.flags |= LCP2_GATE_NATIVE_GENERATED(2);
  const uint32_t e = 0, f = 1, g = 2, h = 3, d = 4, w = 5, e_new = 6, t1 = 7, be = 8, bf = 40, bg = 72, c0 = 104, c3 = 107;
  for (int i = 0; i < 32; i++) a.boolean(be + i);
  for (int i = 0; i < 32; i++) a.boolean(bf + i);
  for (int i = 0; i < 32; i++) a.boolean(bg + i);
  a.recompose(be, e); a.recompose(bf, f); a.recompose(bg, g);
  a.begin();  // T1
  for (int i = 31; i >= 0; i--) {
    a.xor3(3, 4, be + (i + 6) % 32, be + (i + 11) % 32, be + (i + 25) % 32);  // S1(e) bit i
    a.sub(5, W(bf + i), W(bg + i));
    a.muladd(3, R(5), W(be + i));
    a.add(3, R(3), W(bg + i));  // + Ch bit i = g + e (f - g)
    if (i == 31) a.add(1, R(3), a.IMM(0));
    else a.dbladd(1, R(1), R(3));
  }
  a.add(1, R(1), W(h)); a.add(1, R(1), C(0)); a.add(1, R(1), W(w)); a.sub(1, R(1), W(t1));
  carry_tail(a, c0, 3);
  for (int k = 0; k < 3; k++) a.boolean(c0 + k);
  a.begin();  // e_new
  a.add(0, W(d), W(t1));
  a.sub(0, R(0), W(e_new));
  a.mul(1, W(c3), a.IMM(TWO32));
  a.sub(0, R(0), R(1));
  a.emit(R(0));
  a.boolean(c3);
}
void prog_sha_round_a(Asm &a) {
  a.flags |= LCP2_GATE_NATIVE_GENERATED(1);
  const uint32_t wa = 0, wb = 1, wc = 2, t1 = 3, a_new = 4, ba = 8, bb = 40, bc = 72, c0 = 104;
  for (int i = 0; i < 32; i++) a.boolean(ba + i);
  for (int i = 0; i < 32; i++) a.boolean(bb + i);
  for (int i = 0; i < 32; i++) a.boolean(bc + i);
  a.recompose(ba, wa); a.recompose(bb, wb); a.recompose(bc, wc);
  a.begin();
  for (int i = 31; i >= 0; i--) {
    a.xor3(3, 4, ba + (i + 2) % 32, ba + (i + 13) % 32, ba + (i + 22) % 32);  // S0(a) bit i
    a.muladd(3, W(ba + i), W(bb + i));   // + ab
    a.xor_(5, W(ba + i), W(bb + i));     // a ^ b
    a.muladd(3, R(5), W(bc + i));        // + c (a ^ b): Maj = ab + c (a ^ b)
    if (i == 31) a.add(1, R(3), a.IMM(0));
    else a.dbladd(1, R(1), R(3));
  }
  a.add(1, R(1), W(t1)); a.sub(1, R(1), W(a_new));
  carry_tail(a, c0, 2);
  for (int k = 0; k < 2; k++) a.boolean(c0 + k);
}
void prog_sha_sched(Asm &a) {
  a.flags |= LCP2_GATE_NATIVE_GENERATED(3);
  const uint32_t w2 = 0, w7 = 1, w15 = 2, w16 = 3, wt = 4, b2 = 8, b15 = 40, c0 = 104;
  for (int i = 0; i < 32; i++) a.boolean(b2 + i);
  for (int i = 0; i < 32; i++) a.boolean(b15 + i);
  a.recompose(b2, w2); a.recompose(b15, w15);
  a.begin();
  for (int i = 31; i >= 0; i--) {
    a.xor3(3, 4, b15 + (i + 7) % 32, b15 + (i + 18) % 32, i + 3 < 32 ? (int)(b15 + i + 3) : -1);   // s0(w15) bit i
    a.xor3(5, 4, b2 + (i + 17) % 32, b2 + (i + 19) % 32, i + 10 < 32 ? (int)(b2 + i + 10) : -1);   // s1(w2) bit i
    a.add(3, R(3), R(5));
    if (i == 31) a.add(1, R(3), a.IMM(0));
    else a.dbladd(1, R(1), R(3));
  }
  a.add(1, R(1), W(w7)); a.add(1, R(1), W(w16)); a.sub(1, R(1), W(wt));
  carry_tail(a, c0, 2);
  for (int k = 0; k < 2; k++) a.boolean(c0 + k);
}
}  // namespace

// selector groups: plonky2 gates/selectors.rs greedy grouping with max_degree = quotient_degree_factor + 1
GateSetLayout build_gate_set(uint32_t max_degree) {
  GateSetLayout gs;
  uint32_t n = G_COUNT;
  std::vector<std::pair<uint32_t, uint32_t>> groups;
  if (GATE_DEGREE[n - 1] + n - 1 <= max_degree) groups.push_back({0, n});
  else {
    uint32_t start = 0;
    while (start < n) {
      uint32_t size = 0;
      while (start + size < n && size + GATE_DEGREE[start + size] < max_degree) size++;
      if (size == 0) throw std::runtime_error("gate degree too high for the quotient degree factor");
      groups.push_back({start, start + size});
      start += size;
    }
  }
  gs.groups = groups;
  gs.num_selectors = (uint32_t)groups.size();
  gs.imm.push_back(0);
  for (uint32_t g = 0; g < n; g++) {
    Asm a(gs.code, gs.imm);
    uint32_t off = (uint32_t)gs.code.size() / 2;
    switch (g) {
      case G_NOOP: prog_noop(a); break;
      case G_CONSTANT: prog_constant(a); break;
      case G_PUBLIC_INPUT: prog_public_input(a); break;
      case G_SHA_ADD: prog_sha_add(a); break;
      case G_ARITHMETIC: prog_arithmetic(a); break;
      case G_SHA_ROUND_A: prog_sha_round_a(a); break;
      case G_SHA_ROUND_E: prog_sha_round_e(a); break;
      case G_SHA_SCHED: prog_sha_sched(a); break;
      case G_POSEIDON: prog_poseidon(a); break;
    }
    a.finish();
    uint32_t sel = 0;
    for (uint32_t s = 0; s < groups.size(); s++) if (groups[s].first <= g && g < groups[s].second) sel = s;
    lcp2_gate G{sel, g, groups[sel].first, groups[sel].second, off, (uint32_t)gs.code.size() / 2 - off, a.nconstraints, a.flags};
    gs.gates.push_back(G);
    if (a.max_reg > gs.num_regs) gs.num_regs = a.max_reg;
  }
  if (gs.code.empty()) { gs.code.push_back(0); gs.code.push_back(0); }
  return gs;
}

}  // namespace lc
