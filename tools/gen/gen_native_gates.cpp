// Offline generator: turns gate programs into straight-line device functions (csrc/generated_gates*.hpp).
//   sources   the host layer's own SHA-256 gate set (eth-lc-plonky2_amd/host/gates.cpp, programs 0-3) and the text dump of the
//             Python gate libraries (tools/gen/reference_gate_programs.txt, written by tools/gen/dump_reference_programs.py: the
//             plonky2_u32 / comparison / coset-interpolation gates the reference's circuit is made of, and plonky2's recursion gates)
// The interpreter executes a program one instruction at a time with its registers in LDS; the generated form is the same
// instructions as plain C++ expressions in single-assignment form, scheduled for register pressure (below): values are VGPRs, a wire
// is loaded one window before its first use, immediates are literals, and the constraints are combined as weighted terms
// (csrc/quotient_common.hpp QTermsLds: alpha^e from a limb table, no reduction per constraint) instead of a Horner chain.  A gate
// that carries LCP2_GATE_NATIVE_GENERATED(k) in its flags claims to be program k; lcp2_circuit_create checks the claim on random
// points against the interpreter, so a stale generated file cannot produce a wrong proof, only a refused build().
//
// Programs of the dump are additionally rewritten where the arithmetic allows (the SHA-256 gates keep the forms measured in round 3):
//   lazy values     a product that only feeds products, constraint terms or the lazy side of an add / sub is not canonicalised
//                   (gl_mul_nc, gl_add_nc, gl_sub_nc: any u64 congruent to the element);
//   constants       x * 2^s is a shift-reduce (gl_shl), x * c with c < 2^32 two 32-bit multiply-adds and one fold (gl_mul_u32);
//   range products  EMIT(((x - 1) x (x - 2)) (x - 3)) - the base-4 digit check of every plonky2_u32 limb - is y (y + 2) with
//                   y = x (x - 3): two multiplications instead of three (the same polynomial, and the build()-time check says so);
//   PMDS            the Poseidon MDS layer on a register window becomes 12 row sums on 32-bit halves (q_mds_row).
//
//     tools/gen/run.sh        regenerates the headers in eth-lc-plonky2_amd/csrc/
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>
#include "../../eth-lc-plonky2_amd/host/host_internal.hpp"

using namespace lc;

enum : uint32_t { OP_RANGE4 = 32, OP_MDS_ROW = 33 };  // generator-internal node kinds

// One instruction in single-assignment form: every register write gets its own name (v<node>), so that any order respecting the
// data dependences is a valid program - the constraints carry their alpha exponent explicitly (weighted terms), so their order is free.
struct Node {
  uint32_t op = 0;
  std::string a, b, acc;            // operand expressions; acc: the old value of dst (MULADD)
  int na = -1, nb = -1, nacc = -1;  // the nodes behind a / b / acc (-1: a wire, constant, immediate or literal zero)
  bool imm_a = false, imm_b = false;
  uint64_t va = 0, vb = 0;          // immediate values
  std::vector<int> deps;            // nodes whose values are read
  std::vector<std::string> loads;   // wires / gate constants read ("w12", "k0")
  int exponent = -1;                // EMIT / EMITBOOL / RANGE4: the alpha exponent of the constraint
  std::vector<int> users;
  int height = 0;                   // longest dependence chain below this node
  int block = 0;                    // program position of the first constraint this value feeds (rewriting programs: scheduling order)
  bool done = false, dead = false;
  bool lazy = false;                // the value may be any u64 congruent to the element (every use tolerates it)
  bool swap = false;                // ADD: operands exchanged at emission (the lazy one goes first)
  std::vector<int> mds_in;          // MDS_ROW: the 12 window values (-1: zero)
  uint32_t mds_row = 0;
};

struct Program {
  std::string name;
  uint32_t flags = 0, m = 0;
  std::vector<uint32_t> code;       // 2 words per instruction
  const std::vector<uint64_t> *imm = nullptr;
  size_t window_loads = 8;          // 0: the plain form (every wire loaded at the top)
  uint32_t waves = 0;               // waves per SIMD the kernel is built for; 0: from the peak of live values
  bool rewrite = false;             // lazy values / constant multiplications / range products (programs of the dump)
  std::string group;
};

struct Result { std::string text; size_t peak = 0; uint32_t waves = 2; size_t instructions = 0, wires = 0; };

static bool is_emit(uint32_t op) { return op == LCP2_OP_EMIT || op == LCP2_OP_EMITBOOL || op == OP_RANGE4; }

static std::string hexlit(uint64_t v) {
  char buf[40];
  snprintf(buf, sizeof buf, "0x%llxull", (unsigned long long)v);
  return buf;
}

static void appendf(std::string &s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
static void appendf(std::string &s, const char *fmt, ...) {
  char buf[4096];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  s += buf;
}

// ---- program -> nodes
static bool build_nodes(const Program &P, std::vector<Node> &nodes, std::set<std::string> &all_loads, std::set<std::string> &pis) {
  std::vector<int> cur(256, -1);  // the node holding the current value of a register (-1: still zero)
  std::map<std::string, int> seen;  // rewriting programs: an instruction that repeats an earlier one (same operation on the same values) reuses its value
  const bool fwd = (P.flags & LCP2_GATE_EMIT_FORWARD) != 0;
  uint32_t emitted = 0;
  for (size_t pc = 0; pc < P.code.size() / 2; pc++) {
    const uint32_t w0 = P.code[2 * pc], w1 = P.code[2 * pc + 1], op = w0 & 0xF, dst = (w0 >> 8) & 0xFF;
    const uint32_t kk[2] = {(w0 >> 16) & 0xF, (w0 >> 20) & 0xF}, ii[2] = {w1 & 0xFFFF, w1 >> 16};
    const bool emits = op == LCP2_OP_EMIT || op == LCP2_OP_EMITBOOL;
    if (op == LCP2_OP_PMDS) {
      if (!P.rewrite) { fprintf(stderr, "generator: PMDS in a program without rewriting\n"); return false; }
      std::vector<int> window(12);
      for (int j = 0; j < 12; j++) window[j] = cur[ii[0] + j];
      std::vector<int> made(12);
      for (uint32_t r = 0; r < 12; r++) {
        Node n;
        n.op = OP_MDS_ROW;
        n.mds_row = r;
        n.mds_in = window;
        n.vb = (*P.imm)[ii[1] + r];
        for (int d : window) if (d >= 0 && std::find(n.deps.begin(), n.deps.end(), d) == n.deps.end()) n.deps.push_back(d);
        made[r] = (int)nodes.size();
        nodes.push_back(n);
      }
      for (uint32_t r = 0; r < 12; r++) cur[dst + r] = made[r];
      continue;
    }
    if (op > LCP2_OP_MULADD) { fprintf(stderr, "generator: op %u is not supported (%s)\n", op, P.name.c_str()); return false; }
    Node n;
    n.op = op;
    std::string *out[2] = {&n.a, &n.b};
    int *outn[2] = {&n.na, &n.nb};
    for (int s = 0; s < (emits ? 1 : 2); s++) {
      char buf[64];
      switch (kk[s]) {
        case 0:
          if (cur[ii[s]] < 0) { snprintf(buf, sizeof buf, "0ull"); (s ? n.imm_b : n.imm_a) = true; }
          else { snprintf(buf, sizeof buf, "v%d", cur[ii[s]]); n.deps.push_back(cur[ii[s]]); *outn[s] = cur[ii[s]]; }
          break;
        case 1: snprintf(buf, sizeof buf, "w%u", ii[s]); n.loads.push_back(buf); break;
        case 2: snprintf(buf, sizeof buf, "k%u", ii[s]); n.loads.push_back(buf); break;
        case 3:
          snprintf(buf, sizeof buf, "0x%llxull", (unsigned long long)(*P.imm)[ii[s]]);
          (s ? n.imm_b : n.imm_a) = true;
          (s ? n.vb : n.va) = (*P.imm)[ii[s]];
          break;
        default: snprintf(buf, sizeof buf, "pi%u", ii[s]); pis.insert(buf); break;
      }
      *out[s] = buf;
    }
    if (op == LCP2_OP_MULADD) {
      if (cur[dst] < 0) n.acc = "0ull";
      else { n.acc = "v" + std::to_string(cur[dst]); n.deps.push_back(cur[dst]); n.nacc = cur[dst]; }
    }
    // a gate lists its constraints last to first unless it is a forward gate; either way the j-th constraint has the weight alpha^j
    if (emits) { n.exponent = fwd ? (int)emitted : (int)(P.m - 1 - emitted); emitted++; }
    else {
      if (P.rewrite) {
        const std::string key = std::to_string(op) + "|" + n.a + "|" + n.b + "|" + n.acc;
        auto it = seen.find(key);
        if (it != seen.end()) { cur[dst] = it->second; continue; }
        seen[key] = (int)nodes.size();
      }
      cur[dst] = (int)nodes.size();
    }
    for (const std::string &l : n.loads) all_loads.insert(l);
    nodes.push_back(n);
  }
  if (emitted != P.m) { fprintf(stderr, "generator: %u constraints emitted, %s declares %u\n", emitted, P.name.c_str(), P.m); return false; }
  return true;
}

static void link_users(std::vector<Node> &nodes) {
  for (Node &n : nodes) n.users.clear();
  for (size_t j = 0; j < nodes.size(); j++)
    if (!nodes[j].dead)
      for (int d : nodes[j].deps) nodes[d].users.push_back((int)j);
}

// EMIT(MUL(MUL(MUL(SUB(x, 1), x), SUB(x, 2)), SUB(x, 3))) -> RANGE4(x); x a wire
static void rewrite_range_products(std::vector<Node> &nodes) {
  link_users(nodes);
  auto single_use = [&](int j) { return j >= 0 && nodes[j].users.size() == 1; };
  auto sub_of = [&](int j, const std::string &x, uint64_t k) {
    return j >= 0 && nodes[j].op == LCP2_OP_SUB && nodes[j].na < 0 && nodes[j].a == x && nodes[j].imm_b && nodes[j].vb == k && !nodes[j].imm_a;
  };
  for (size_t e = 0; e < nodes.size(); e++) {
    Node &E = nodes[e];
    if (E.op != LCP2_OP_EMIT || E.na < 0 || !single_use(E.na)) continue;
    const int m3 = E.na;
    if (nodes[m3].op != LCP2_OP_MUL) continue;
    const int m2 = nodes[m3].na, s3 = nodes[m3].nb;
    if (m2 < 0 || s3 < 0 || !single_use(m2) || !single_use(s3) || nodes[m2].op != LCP2_OP_MUL) continue;
    const int m1 = nodes[m2].na, s2 = nodes[m2].nb;
    if (m1 < 0 || s2 < 0 || !single_use(m1) || !single_use(s2) || nodes[m1].op != LCP2_OP_MUL) continue;
    const int s1 = nodes[m1].na;
    if (s1 < 0 || !single_use(s1) || nodes[m1].nb >= 0 || nodes[m1].imm_b) continue;
    const std::string x = nodes[m1].b;
    if (x.empty() || x[0] != 'w') continue;
    if (!sub_of(s1, x, 1) || !sub_of(s2, x, 2) || !sub_of(s3, x, 3)) continue;
    for (int d : {m3, m2, m1, s1, s2, s3}) { nodes[d].dead = true; nodes[d].done = true; }
    E.op = OP_RANGE4;
    E.a = x; E.na = -1;
    E.deps.clear();
    E.loads = {x};
  }
  link_users(nodes);
}

// which values may stay lazy: decided from the last node to the first (a node's users come after it)
static void mark_lazy(std::vector<Node> &nodes) {
  link_users(nodes);
  auto is_node = [](int j) { return j >= 0; };
  for (size_t jj = nodes.size(); jj-- > 0;) {
    Node &n = nodes[jj];
    if (n.dead || is_emit(n.op)) continue;
    if (n.users.empty()) { n.lazy = false; continue; }
    bool ok = true;
    for (int u : n.users) {
      const Node &U = nodes[u];
      const bool at_a = U.na == (int)jj, at_b = U.nb == (int)jj, at_acc = U.nacc == (int)jj;
      bool tol = false;
      switch (U.op) {
        case LCP2_OP_MUL: tol = !at_acc; break;
        case LCP2_OP_EMIT: tol = true; break;
        case LCP2_OP_MULADD: tol = !at_acc; break;  // the product takes any u64; the accumulator is added canonically
        case LCP2_OP_ADD:
          // one lazy operand: a if b is not a node (or is a canonical node: decided already? no - b's own decision comes later, so
          // only the first node operand may be lazy); b if a is not a node
          if (U.lazy && !(at_a && at_b)) tol = at_a ? true : (at_b && !is_node(U.na));
          break;
        case LCP2_OP_SUB: tol = U.lazy && at_a && !at_b; break;
        default: tol = false; break;  // XOR, DBLADD, EMITBOOL, MDS_ROW inputs (taken as halves: any u64 is fine, but keep them canonical)
      }
      if (U.op == OP_MDS_ROW) tol = true;  // the row sums split their inputs into 32-bit halves: any u64
      ok = ok && tol;
    }
    n.lazy = ok;
  }
  // an ADD whose lazy operand sits at b is emitted with its operands exchanged
  for (Node &n : nodes)
    if (!n.dead && n.op == LCP2_OP_ADD && n.lazy && n.nb >= 0 && nodes[n.nb].lazy && n.na < 0) n.swap = true;
}

static int log2_exact(uint64_t v) {
  if (!v || (v & (v - 1))) return -1;
  int s = 0;
  while (!(v & 1)) { v >>= 1; s++; }
  return s;
}

// the C++ statement of a node (rewriting programs)
static std::string statement(const std::vector<Node> &nodes, int j) {
  const Node &n = nodes[j];
  const char *A = n.a.c_str(), *B = n.b.c_str();
  std::string s;
  switch (n.op) {
    case LCP2_OP_ADD:
      if (n.lazy) appendf(s, "  u64 v%d = gl_add_nc(%s, %s);\n", j, n.swap ? B : A, n.swap ? A : B);
      else appendf(s, "  u64 v%d = gl_add(%s, %s);\n", j, A, B);
      break;
    case LCP2_OP_SUB:
      if (n.lazy) appendf(s, "  u64 v%d = gl_sub_nc(%s, %s);\n", j, A, B);
      else appendf(s, "  u64 v%d = gl_sub(%s, %s);\n", j, A, B);
      break;
    case LCP2_OP_MUL: {
      const bool ia = n.imm_a && n.a != "0ull", ib = n.imm_b && n.b != "0ull";
      const char *X = ia ? B : A;
      const uint64_t c = ia ? n.va : n.vb;
      const int sh = (ia || ib) ? log2_exact(c) : -1;
      if ((ia || ib) && sh >= 1 && sh <= 32) appendf(s, "  u64 v%d = %s<%d>(%s);\n", j, n.lazy ? "gl_shl_nc" : "gl_shl", sh, X);
      else if ((ia || ib) && c < (1ull << 32) && c > 1) appendf(s, "  u64 v%d = %s(%s, 0x%llxu);\n", j, n.lazy ? "gl_mul_u32_nc" : "gl_mul_u32", X, (unsigned long long)c);
      else appendf(s, "  u64 v%d = %s(%s, %s);\n", j, n.lazy ? "gl_mul_nc" : "gl_mul", A, B);
      break;
    }
    case LCP2_OP_EMIT: appendf(s, "  terms.add<%d>(%s);\n", n.exponent, A); break;
    case LCP2_OP_XOR: appendf(s, "  u64 t%d = gl_mul(%s, %s), v%d = gl_sub(gl_sub(gl_add(%s, %s), t%d), t%d);\n", j, A, B, j, A, B, j, j); break;
    case LCP2_OP_DBLADD: appendf(s, "  u64 v%d = gl_add(gl_add(%s, %s), %s);\n", j, A, A, B); break;
    case LCP2_OP_EMITBOOL: appendf(s, "  terms.add<%d>(gl_mul_nc(%s, gl_sub(%s, 1)));\n", n.exponent, A, A); break;  // x^2 - x = x (x - 1), lazily
    case OP_RANGE4: appendf(s, "  { const u64 y = gl_mul(%s, gl_sub(%s, 3)); terms.add<%d>(gl_mul_nc(y, gl_add_nc(y, 2))); }\n", A, A, n.exponent); break;
    case OP_MDS_ROW: {
      appendf(s, "  u64 v%d = q_mds_row<%u>(", j, n.mds_row);
      for (int t = 0; t < 12; t++) {
        if (n.mds_in[t] < 0) s += "0ull, ";
        else appendf(s, "v%d, ", n.mds_in[t]);
      }
      appendf(s, "%s);\n", hexlit(n.vb).c_str());
      break;
    }
    default:  // MULADD
      if (n.lazy) appendf(s, "  u64 v%d = gl_add_nc(gl_mul_nc(%s, %s), %s);\n", j, A, B, n.acc.c_str());
      else appendf(s, "  u64 v%d = gl_add(%s, gl_mul(%s, %s));\n", j, n.acc.c_str(), A, B);
      break;
  }
  return s;
}

static bool generate_with(const Program &P, int k, Result &R, bool greedy_all) {
  std::vector<Node> nodes;
  std::set<std::string> all_loads, pis;
  if (!build_nodes(P, nodes, all_loads, pis)) return false;
  if (P.rewrite) { rewrite_range_products(nodes); mark_lazy(nodes); }
  link_users(nodes);
  for (size_t j = nodes.size(); j-- > 0;) {
    int h = 0;
    for (int u : nodes[j].users) h = std::max(h, nodes[u].height + 1);
    nodes[j].height = h;
  }
  // the constraint a value belongs to: the first one (in program order) that it feeds.  Programs of the dump are scheduled constraint
  // by constraint - their constraints are many, short and independent, and taking the highest node first would open all of them at
  // once (129 live values in the ReducingExtensionGate) - with the longest chain first inside a constraint.
  for (size_t j = nodes.size(); j-- > 0;) {
    int b = is_emit(nodes[j].op) ? (int)j : (int)nodes.size();
    for (int u : nodes[j].users) b = std::min(b, nodes[u].block);
    nodes[j].block = b;
  }
  size_t live_nodes = 0;
  for (const Node &n : nodes) live_nodes += !n.dead;
  // ---- list scheduling
  std::vector<int> order;
  std::set<std::string> loaded;
  auto ready = [&](const Node &n) {
    if (n.done) return false;
    for (int d : n.deps) if (!nodes[d].done) return false;
    return true;
  };
  auto free_now = [&](const Node &n) {  // every wire it reads is in a register already
    for (const std::string &l : n.loads) if (!loaded.count(l)) return false;
    return true;
  };
  auto take = [&](int j) {
    nodes[j].done = true;
    for (const std::string &l : nodes[j].loads) loaded.insert(l);
    order.push_back(j);
  };
  while (order.size() < live_nodes) {
    int best = -1;
    for (size_t j = 0; j < nodes.size(); j++)
      if (ready(nodes[j]) && (best < 0 || (P.rewrite ? (nodes[j].block < nodes[best].block || (nodes[j].block == nodes[best].block && nodes[j].height > nodes[best].height))
                                                     : nodes[j].height > nodes[best].height)))
        best = (int)j;
    take(best);
    const int cur_block = nodes[best].block;
    for (bool again = true; again;) {  // what the loaded wires make computable at no new load: booleanity terms, recomposition steps
      again = false;                   // (rewriting programs: only constraint terms and values of the constraint in progress - nothing that would sit in a register for long)
      for (size_t j = 0; j < nodes.size(); j++)
        if (ready(nodes[j]) && free_now(nodes[j]) && (!P.rewrite || greedy_all || is_emit(nodes[j].op) || nodes[j].block <= cur_block)) { take((int)j); again = true; }
    }
  }
  size_t peak = 0;
  {  // the number of values live across each instruction of the schedule (wires until their last use)
    std::map<std::string, int> last_use;
    std::vector<int> last_val(nodes.size(), -1);
    for (size_t t = 0; t < order.size(); t++) {
      for (const std::string &l : nodes[order[t]].loads) last_use[l] = (int)t;
      for (int d : nodes[order[t]].deps) last_val[d] = (int)t;
    }
    std::set<std::string> live_w;
    std::set<int> live_v;
    for (size_t t = 0; t < order.size(); t++) {
      const int j = order[t];
      for (const std::string &l : nodes[j].loads) live_w.insert(l);
      if (nodes[j].exponent < 0) live_v.insert(j);
      peak = std::max(peak, live_w.size() + live_v.size());
      for (const std::string &l : nodes[j].loads) if (last_use[l] == (int)t) live_w.erase(l);
      for (int d : nodes[j].deps) if (last_val[d] == (int)t) live_v.erase(d);
      if (nodes[j].exponent < 0 && last_val[j] < 0) live_v.erase(j);
    }
  }
  R.peak = peak;
  R.instructions = P.code.size() / 2;
  R.wires = all_loads.size();
  std::string &o = R.text;
  if (P.window_loads == 0) {
    // ---- the plain form: program order, every wire one load that hipcc hoists to the top (it does), the limb table read through
    // the constant address space.  The ScheduleGate keeps two rotated 32-bit words alive throughout (66 live values): with two
    // waves per SIMD either way, all of its loads in flight at once beat the windows (1.28 against 1.36 ms at 2^19 rows).
    R.waves = 2;
    appendf(o, "// %s: %zu instructions, %u constraints, %zu wires and gate constants (plain form)\n", P.name.c_str(), P.code.size() / 2, P.m, all_loads.size());
    appendf(o, "template <> __device__ __forceinline__ void q_generated<%d>(const QuotientArgs &a, u64 i, QEmit &emit) {\n", k);
    o += "  const u64 *W = a.wires + i;\n  const u64 st = a.stride;\n";
    for (const std::string &l : all_loads) {
      if (l[0] == 'w') appendf(o, "  const u64 %s = W[%sull * st];\n", l.c_str(), l.c_str() + 1);
      else appendf(o, "  const u64 %s = a.consts[(u64)(a.num_selectors + %s) * st + i];\n", l.c_str(), l.c_str() + 1);
    }
    for (const std::string &p : pis) appendf(o, "  const u64 %s = konst(a.pis)[%s];\n", p.c_str(), p.c_str() + 2);
    o += "  QTerms &terms = emit.t;\n  emit.begin_terms();\n";
    for (size_t j = 0; j < nodes.size(); j++) {
      const Node &n = nodes[j];
      const char *A = n.a.c_str(), *B = n.b.c_str();
      const int jj = (int)j;
      switch (n.op) {
        case LCP2_OP_ADD: appendf(o, "  const u64 v%d = gl_add(%s, %s);\n", jj, A, B); break;
        case LCP2_OP_SUB: appendf(o, "  const u64 v%d = gl_sub(%s, %s);\n", jj, A, B); break;
        case LCP2_OP_MUL: appendf(o, "  const u64 v%d = gl_mul(%s, %s);\n", jj, A, B); break;
        case LCP2_OP_EMIT: appendf(o, "  terms.add<%d>(a, %s);\n", n.exponent, A); break;
        case LCP2_OP_XOR: appendf(o, "  const u64 t%d = gl_mul(%s, %s), v%d = gl_sub(gl_sub(gl_add(%s, %s), t%d), t%d);\n", jj, A, B, jj, A, B, jj, jj); break;
        case LCP2_OP_DBLADD: appendf(o, "  const u64 v%d = gl_add(gl_add(%s, %s), %s);\n", jj, A, A, B); break;
        case LCP2_OP_EMITBOOL: appendf(o, "  terms.add<%d>(a, gl_mul_nc(%s, gl_sub(%s, 1)));\n", n.exponent, A, A); break;
        default: appendf(o, "  const u64 v%d = gl_add(%s, gl_mul(%s, %s));\n", jj, n.acc.c_str(), A, B); break;
      }
    }
    o += "  emit.finish_terms();\n}\n\n";
    return true;
  }
  // ---- windows: WINDOW_LOADS new wires each; the loads of window k + 1 are issued at the top of window k
  const size_t WINDOW_LOADS = P.window_loads;
  std::vector<std::vector<int>> windows(1);
  std::vector<std::vector<std::string>> window_loads(1);
  {
    std::set<std::string> declared;
    for (int j : order) {
      std::vector<std::string> fresh;
      for (const std::string &l : nodes[j].loads)
        if (!declared.count(l) && std::find(fresh.begin(), fresh.end(), l) == fresh.end()) fresh.push_back(l);
      if (window_loads.back().size() + fresh.size() > WINDOW_LOADS && !windows.back().empty()) { windows.emplace_back(); window_loads.emplace_back(); }
      for (const std::string &l : fresh) { declared.insert(l); window_loads.back().push_back(l); }
      windows.back().push_back(j);
    }
  }
  // occupancy the kernel asks for (k_q_gen's launch bounds): 2 VGPRs per live value, 24 for the column sums, 16 for the loads in
  // flight, ~25 of temporaries and addresses; 512 VGPRs per SIMD lane
  {  // 2 VGPRs per live value, two windows of loads in flight, 24 for the column sums, ~36 of temporaries and addresses; 512 per SIMD lane
    const size_t vg = 2 * peak + 4 * WINDOW_LOADS + 60;  // (measured: 16-load windows are no faster than 8 for the plonky2_u32 gates, profiles/r04_k6_experiments.md)
    R.waves = P.waves ? P.waves : vg <= 140 ? 4 : vg <= 192 ? 3 : 2;
  }
  // ---- emission
  appendf(o, "// %s: %zu instructions, %u constraints, %zu wires and gate constants; at most %zu 64-bit values live in this schedule\n",
          P.name.c_str(), P.code.size() / 2, P.m, all_loads.size(), peak);
  appendf(o, "template <> __device__ __forceinline__ void q_generated<%d>(const QuotientArgs &a, u64 i, QEmit &emit) {\n", k);
  o += "  const u64 *W = a.wires + i;\n  const u64 st = a.stride;\n";
  for (const std::string &p : pis) appendf(o, "  const u64 %s = konst(a.pis)[%s];\n", p.c_str(), p.c_str() + 2);
  o += "  QTermsLds &terms = emit.tl;\n  emit.begin_terms_lds();\n";
  auto print_loads = [&](const std::vector<std::string> &ls) {
    for (const std::string &l : ls) {
      if (l[0] == 'w') appendf(o, "  u64 %s = W[%sull * st];\n", l.c_str(), l.c_str() + 1);
      else appendf(o, "  u64 %s = a.consts[(u64)(a.num_selectors + %s) * st + i];\n", l.c_str(), l.c_str() + 1);
    }
  };
  print_loads(window_loads[0]);
  for (size_t wdw = 0; wdw < windows.size(); wdw++) {
    if (wdw + 1 < windows.size()) print_loads(window_loads[wdw + 1]);
    for (int j : windows[wdw]) {
      if (P.rewrite) { o += statement(nodes, j); continue; }
      const Node &n = nodes[j];
      const char *A = n.a.c_str(), *B = n.b.c_str();
      switch (n.op) {
        case LCP2_OP_ADD: appendf(o, "  u64 v%d = gl_add(%s, %s);\n", j, A, B); break;
        case LCP2_OP_SUB: appendf(o, "  u64 v%d = gl_sub(%s, %s);\n", j, A, B); break;
        case LCP2_OP_MUL: appendf(o, "  u64 v%d = gl_mul(%s, %s);\n", j, A, B); break;
        case LCP2_OP_EMIT: appendf(o, "  terms.add<%d>(%s);\n", n.exponent, A); break;
        case LCP2_OP_XOR: appendf(o, "  u64 t%d = gl_mul(%s, %s), v%d = gl_sub(gl_sub(gl_add(%s, %s), t%d), t%d);\n", j, A, B, j, A, B, j, j); break;
        case LCP2_OP_DBLADD: appendf(o, "  u64 v%d = gl_add(gl_add(%s, %s), %s);\n", j, A, A, B); break;
        case LCP2_OP_EMITBOOL: appendf(o, "  terms.add<%d>(gl_mul_nc(%s, gl_sub(%s, 1)));\n", n.exponent, A, A); break;  // x^2 - x = x (x - 1), lazily
        default: appendf(o, "  u64 v%d = gl_add(%s, gl_mul(%s, %s));\n", j, n.acc.c_str(), A, B); break;
      }
    }
    if (wdw + 1 < windows.size()) {
      // everything alive across the boundary goes through an empty asm: hipcc's DAG scheduler would otherwise move a chain whose
      // result is needed late (a recomposition) down to that use, with all its wires alive until there.  (The next window's
      // wires, just requested, are left alone: pinning them would wait for the loads.)
      std::set<int> in_or_before;
      std::set<std::string> used_wires;
      for (size_t u = 0; u <= wdw; u++)
        for (int j : windows[u]) { in_or_before.insert(j); for (const std::string &l : nodes[j].loads) used_wires.insert(l); }
      std::set<int> live_vals;
      std::set<std::string> live_wires;
      for (size_t u = wdw + 1; u < windows.size(); u++)
        for (int j : windows[u]) {
          for (int d : nodes[j].deps) if (in_or_before.count(d)) live_vals.insert(d);
          for (const std::string &l : nodes[j].loads) if (used_wires.count(l)) live_wires.insert(l);
        }
      o += "  terms.pin();";
      for (int d : live_vals) appendf(o, " Q_PIN(v%d);", d);
      for (const std::string &l : live_wires) appendf(o, " Q_PIN(%s);", l.c_str());
      o += "\n  Q_WINDOW_BARRIER();\n";
    }
  }
  o += "  emit.finish_terms_lds();\n}\n\n";
  return true;
}

// programs of the dump: two greedy policies of the list scheduler (everything the loaded wires make computable / only what belongs to
// the constraint in progress), the schedule with fewer live values wins
static bool generate(const Program &P, int k, Result &R) {
  if (!P.rewrite) return generate_with(P, k, R, true);
  Result A, B;
  if (!generate_with(P, k, A, true) || !generate_with(P, k, B, false)) return false;
  R = B.peak < A.peak ? B : A;
  return true;
}

// ---- the text dump of the Python gate libraries
struct Dump {
  std::vector<std::vector<uint64_t>> imm_tables;  // one per gate set (kept alive: programs point into them)
  std::vector<Program> programs;
};
static bool read_dump(const char *path, Dump &D) {
  FILE *f = fopen(path, "r");
  if (!f) { fprintf(stderr, "generator: cannot open %s\n", path); return false; }
  D.imm_tables.reserve(16);
  char tok[256];
  std::string set_name;
  while (fscanf(f, "%255s", tok) == 1) {
    if (tok[0] == '#') { int c; while ((c = fgetc(f)) != EOF && c != '\n') {} continue; }
    if (!strcmp(tok, "end")) break;
    if (!strcmp(tok, "gateset")) {
      unsigned long long n;
      if (fscanf(f, "%255s %llx", tok, &n) != 2) return false;
      set_name = tok;
      D.imm_tables.emplace_back(n);
      for (auto &v : D.imm_tables.back()) { unsigned long long x; if (fscanf(f, "%llx", &x) != 1) return false; v = x; }
      continue;
    }
    if (!strcmp(tok, "gate")) {
      Program P;
      unsigned flags, m, len;
      if (fscanf(f, "%255s %x %x %x", tok, &flags, &m, &len) != 4) return false;
      P.name = tok; P.flags = flags; P.m = m;
      P.code.resize(2 * (size_t)len);
      for (auto &w : P.code) { unsigned x; if (fscanf(f, "%x", &x) != 1) return false; w = x; }
      P.imm = &D.imm_tables.back();
      P.rewrite = true;
      if (const char *w = getenv("GEN_WINDOW")) P.window_loads = (size_t)atoi(w);  // experiments only: the committed headers are made without it
      P.group = set_name;
      D.programs.push_back(P);
      continue;
    }
    fprintf(stderr, "generator: unexpected token %s in %s\n", tok, path);
    return false;
  }
  fclose(f);
  return true;
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: gen_native_gates <reference_gate_programs.txt> <output directory>\n"); return 2; }
  const std::string outdir = argv[2];
  GateSetLayout gs = build_gate_set(9);
  std::vector<Program> programs;
  {
    const uint32_t targets[] = {G_SHA_ADD, G_SHA_ROUND_A, G_SHA_ROUND_E, G_SHA_SCHED};
    // wires per window: more loads in flight where the register budget of the gate's kernel allows (the ShaAddGate runs 4 waves per
    // SIMD, the round gates 3, the ScheduleGate - two rotated words alive throughout - 2, in the plain form)
    const size_t window_loads_of[] = {8, 8, 8, 0};
    const uint32_t waves_of[] = {4, 3, 3, 2};  // as measured in round 3 (102 / 168 / 165 / 185 VGPRs)
    int t = 0;
    for (uint32_t g : targets) {
      const lcp2_gate &G = gs.gates[g];
      Program P;
      P.name = gate_name(g);
      P.flags = G.flags & LCP2_GATE_EMIT_FORWARD; P.m = G.num_constraints;
      P.code.assign(gs.code.begin() + 2 * (size_t)G.code_offset, gs.code.begin() + 2 * ((size_t)G.code_offset + G.code_len));
      P.imm = &gs.imm;
      P.window_loads = window_loads_of[t];
      P.waves = waves_of[t++];
      if (getenv("GEN_SHA_REWRITE")) { P.rewrite = true; P.waves = 0; if (!P.window_loads) P.window_loads = 8; }  // experiments only
      P.group = "sha";
      programs.push_back(P);
    }
  }
  Dump D;
  if (!read_dump(argv[1], D)) return 1;
  {
    // the compile units of the dump's programs (csrc/kernels_gates_*.hip): three evaluators each, so that they build side by side
    const char *group_of_reference[] = {"u32a", "u32a", "u32a", "u32b", "u32b", "u32b"};
    size_t r = 0, c = 0;
    for (Program &P : D.programs) {
      if (P.group == "reference") P.group = r < 6 ? group_of_reference[r] : "u32b", r++;
      else P.group = (c++ < 4) ? "reca" : "recb";
      programs.push_back(P);
    }
  }
  const char *banner =
      "// GENERATED by tools/gen/gen_native_gates.cpp from the gate programs of eth-lc-plonky2_amd/host/gates.cpp and\n"
      "// tools/gen/reference_gate_programs.txt -- do not edit.  Straight-line device forms of gate programs (DESIGN.md section \"K6\");\n"
      "// checked against the programs at build().\n"
      "// Instructions are in single-assignment form and SCHEDULED for register pressure: the longest dependence chain first, and\n"
      "// whatever else has become computable from the wires already loaded right behind it; the schedule is cut into windows, a\n"
      "// window's wires are loaded at the top of the window before it, and Q_WINDOW_BARRIER keeps hipcc from undoing that.\n";
  std::map<std::string, std::string> files;
  std::vector<std::string> group_order;
  std::vector<Result> results(programs.size());
  for (size_t k = 0; k < programs.size(); k++) {
    if (!generate(programs[k], (int)k, results[k])) return 1;
    const std::string &g = programs[k].group;
    if (!files.count(g)) { group_order.push_back(g); files[g] = std::string(banner) + "#pragma once\n\nnamespace lcp2 {\n\n"; }
    files[g] += results[k].text;
  }
  std::string unit_ranges;
  for (const std::string &g : group_order) {
    size_t lo = programs.size(), n = 0;
    for (size_t k = 0; k < programs.size(); k++) if (programs[k].group == g) { lo = std::min(lo, k); n++; }
    std::string upper = g;
    for (char &ch : upper) ch = (char)toupper(ch);
    appendf(unit_ranges, "constexpr u32 Q_GENERATED_%s_FIRST = %zu, Q_GENERATED_%s_COUNT = %zu;  // generated_gates_%s.hpp -> kernels_gates_%s.hip\n", upper.c_str(), lo,
            upper.c_str(), n, g.c_str(), g.c_str());
    files[g] += "}  // namespace lcp2\n";
    const std::string path = outdir + "/generated_gates_" + g + ".hpp";
    FILE *f = fopen(path.c_str(), "w");
    if (!f) { fprintf(stderr, "generator: cannot write %s\n", path.c_str()); return 1; }
    fputs(files[g].c_str(), f);
    fclose(f);
  }
  {  // the index: names, occupancy, which compile unit holds which program
    std::string o = banner;
    o += "// Index of the generated evaluators: program k of this table is what LCP2_GATE_NATIVE_GENERATED(k) claims (include/lcp2.h);\n"
         "// eth-lc-plonky2_amd/circuit.py GENERATED_GATE_NAMES lists the same names in the same order.\n"
         "#pragma once\n\nnamespace lcp2 {\n\n";
    appendf(o, "constexpr u32 Q_GENERATED_COUNT = %zu;\n", programs.size());
    o += "// waves per SIMD the kernel of program k is built for (from the peak of live values of its schedule)\n";
    o += "constexpr u32 Q_GENERATED_WAVES[Q_GENERATED_COUNT] = {";
    for (size_t k = 0; k < programs.size(); k++) appendf(o, "%s%u", k ? ", " : "", results[k].waves);
    o += "};\n";
    o += "// the compile units (device code only: a unit includes its file inside #if defined(__HIP_DEVICE_COMPILE__))\n" + unit_ranges;
    for (size_t k = 0; k < programs.size(); k++)
      appendf(o, "// %2zu %-24s %-5s %5zu instructions %4u constraints %4zu columns, peak %3zu live values%s\n", k, programs[k].name.c_str(), programs[k].group.c_str(),
              results[k].instructions, programs[k].m, results[k].wires, results[k].peak, (programs[k].flags & LCP2_GATE_EMIT_FORWARD) ? ", forward" : "");
    o += "\n}  // namespace lcp2\n";
    const std::string path = outdir + "/generated_gates.hpp";
    FILE *f = fopen(path.c_str(), "w");
    if (!f) { fprintf(stderr, "generator: cannot write %s\n", path.c_str()); return 1; }
    fputs(o.c_str(), f);
    fclose(f);
  }
  return 0;
}
