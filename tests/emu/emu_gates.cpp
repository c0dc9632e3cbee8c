// TEST HARNESS (not product code): the generated straight-line gate evaluators (csrc/generated_gates_*.hpp) compiled for the CPU.
// The device-only pieces they are written against - the weighted-term accumulators with their LDS limb table, the scheduling pins - are
// replaced by plain field arithmetic here, so what is checked is the generator's output itself: operand wiring, lazy / canonical
// forms, constant multiplications, the rewritten range products, the alpha exponent of every constraint.  tests/test_generated_gates.py
// compares sum_j alpha^j c_j of every program with a Python interpretation of the program on random points.
// Built by tests/emu_lib.py with g++; never loaded by the package.
#include <utility>
#include "../../eth-lc-plonky2_amd/csrc/gate_helpers.hpp"

namespace lcp2 {

constexpr u32 QUOTIENT_MAX_CH = 2, QUOTIENT_TERM_POWS = 128;
struct QuotientArgs {
  const u64 *wires, *consts, *pis;
  u64 stride;
  u32 num_selectors;
  const u64 *alpha_pow;  // [QUOTIENT_MAX_CH][QUOTIENT_TERM_POWS]
};
template <class T> const T *konst(const T *p) { return p; }

struct QTermsEmu {
  u64 sum[QUOTIENT_MAX_CH];
  const u64 *pw;
  void init(const u64 *p) { pw = p; sum[0] = sum[1] = 0; }
  void pin() {}
  template <u32 E> void add(u64 x) {  // x: any u64
    static_assert(E < QUOTIENT_TERM_POWS, "too many constraints for the alpha power table");
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) sum[c] = gl_add(sum[c], gl_mul(x, pw[c * QUOTIENT_TERM_POWS + E]));
  }
  template <u32 E> void add(const QuotientArgs &, u64 x) { add<E>(x); }
};
typedef QTermsEmu QTerms;
typedef QTermsEmu QTermsLds;
struct QEmit {
  QTermsEmu t, tl;
  const u64 *pw;
  u64 acc[QUOTIENT_MAX_CH];
  void begin_terms() { t.init(pw); }
  void finish_terms() { acc[0] = t.sum[0]; acc[1] = t.sum[1]; }
  void begin_terms_lds() { tl.init(pw); }
  void finish_terms_lds() { acc[0] = tl.sum[0]; acc[1] = tl.sum[1]; }
};
template <u32 K> void q_generated(const QuotientArgs &a, u64 i, QEmit &emit);

}  // namespace lcp2

#define __device__
#define __forceinline__ inline
#define Q_PIN(x) (void)(x)
#define Q_WINDOW_BARRIER() do {} while (0)
#include "../../eth-lc-plonky2_amd/csrc/generated_gates.hpp"
#include "../../eth-lc-plonky2_amd/csrc/generated_gates_sha.hpp"
#include "../../eth-lc-plonky2_amd/csrc/generated_gates_u32a.hpp"
#include "../../eth-lc-plonky2_amd/csrc/generated_gates_u32b.hpp"
#include "../../eth-lc-plonky2_amd/csrc/generated_gates_reca.hpp"
#include "../../eth-lc-plonky2_amd/csrc/generated_gates_recb.hpp"

using namespace lcp2;

template <u32 K> static bool run_one(u32 k, const QuotientArgs &a, u64 i, QEmit &e) {
  if (k != K) return false;
  q_generated<K>(a, i, e);
  return true;
}
template <size_t... I> static bool run_any(std::index_sequence<I...>, u32 k, const QuotientArgs &a, u64 i, QEmit &e) {
  return (run_one<(u32)I>(k, a, i, e) || ...);
}

extern "C" {
unsigned emu_generated_count() { return Q_GENERATED_COUNT; }
unsigned emu_generated_waves(unsigned k) { return k < Q_GENERATED_COUNT ? Q_GENERATED_WAVES[k] : 0; }
// wires [num_wires][count], consts [num_constants incl. selectors][count] (canonical), pis[4], alphas[2] -> out[count][2] = sum_j alpha_c^j constraint_j
int emu_generated_gate(unsigned k, const u64 *wires, const u64 *consts, const u64 *pis, unsigned num_selectors, u64 count, const u64 *alphas, u64 *out) {
  u64 pw[QUOTIENT_MAX_CH * QUOTIENT_TERM_POWS];
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) {
    u64 v = 1;
    for (u32 e = 0; e < QUOTIENT_TERM_POWS; e++) { pw[c * QUOTIENT_TERM_POWS + e] = v; v = gl_mul(v, alphas[c]); }
  }
  QuotientArgs a{wires, consts, pis, count, num_selectors, pw};
  for (u64 i = 0; i < count; i++) {
    QEmit e;
    e.pw = pw;
    if (!run_any(std::make_index_sequence<Q_GENERATED_COUNT>{}, k, a, i, e)) return -1;
    out[2 * i] = e.acc[0];
    out[2 * i + 1] = e.acc[1];
  }
  return 0;
}
u64 emu_gl_mul_u32(u64 x, u32 c) { return gl_mul_u32(x, c); }
u64 emu_gl_shl_nc(u64 x, unsigned s) {
  switch (s) {
    case 1: return gl_canon(gl_shl_nc<1>(x));
    case 2: return gl_canon(gl_shl_nc<2>(x));
    case 31: return gl_canon(gl_shl_nc<31>(x));
    case 32: return gl_canon(gl_shl_nc<32>(x));
    default: return ~0ull;
  }
}
}
