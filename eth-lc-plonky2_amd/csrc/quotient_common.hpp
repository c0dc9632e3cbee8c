// K6 device code shared by the compile units that hold gate evaluators (kernels_prover.hip: interpreter, native plonky2 gates, light
// gates, permutation pass; kernels_gates_*.hip: the generated straight-line evaluators of csrc/generated_gates_*.hpp): operand access,
// the weighted-term accumulators, the gate-program interpreter and the filter.  Device functions only; nothing here is a kernel.
#pragma once
#include "internal.hpp"
#include "poseidon.hpp"
#include "prover_kernels.hpp"
#include "gate_helpers.hpp"

namespace lcp2 {

// Read-only tables reached through an argument struct carry no `__restrict__`, so hipcc cannot prove a uniform load from them
// invariant and emits a vector load + v_readfirstlane (a full memory round trip on the critical path of every interpreter
// instruction).  Viewed through the constant address space the same load is an s_load from the scalar cache.  Only for data
// that no kernel writes while this one runs (code, gate table, immediates, challenges: all uploaded before the launch).
template <class T> using const_as = const T __attribute__((address_space(4))) *;
template <class T> __device__ __forceinline__ const_as<T> konst(const T *p) { return (const_as<T>)(unsigned long long)p; }

// ------------------------------------------------------------------ K6: quotient polynomial values on the LDE coset
// Gate-program interpreter: registers live in LDS (reg r of thread t at lds[r * T + t]: conflict free), decode is
// wave-uniform (scalar unit), operands come from the wires / constants LDE columns at this thread's point.
__device__ __forceinline__ u64 q_operand(const QuotientArgs &a, u32 kind, u32 idx, const u64 *lds, u32 T, u32 tid, u64 i) {
  switch (kind) {
    case 0: return lds[idx * T + tid];
    case 1: return a.wires[(u64)idx * a.stride + i];
    case 2: return a.consts[(u64)(a.num_selectors + idx) * a.stride + i];
    case 3: return konst(a.imm)[idx];
    case 4: return konst(a.pis)[idx];
    default: return lds[(a.num_regs + idx) * T + tid];  // QKIND_STAGE
  }
}

// LDG: `cnt` column values of this thread's point into the staging slots.  All QUOTIENT_STAGE loads are issued back to back
// (lanes past `cnt` repeat entry 0) before the first one is used: one HBM round trip for the whole group.
__device__ __forceinline__ void q_stage(const QuotientArgs &a, u32 cnt, const_as<u32> lst, u64 *lds, u32 T, u32 tid, u64 i) {
  u64 v[QUOTIENT_STAGE];
#pragma unroll
  for (u32 j = 0; j < QUOTIENT_STAGE; j++) {
    const u32 e = lst[j < cnt ? j : 0];
    const u64 *col = e < a.num_wires ? a.wires + (u64)e * a.stride : a.consts + (u64)(e - a.num_wires) * a.stride;
    v[j] = col[i];
  }
#pragma unroll
  for (u32 j = 0; j < QUOTIENT_STAGE; j++)
    if (j < cnt) lds[(a.num_regs + j) * T + tid] = v[j];
}

// Poseidon MDS layer on a window of 12 LDS registers (LCP2_OP_PMDS): the small-constant circulant on 32-bit halves with one
// fold per element, exactly the layer of the hash kernels (poseidon.hpp), the 12 constants riding in the accumulators.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void q_pmds(u64 *lds, u32 T, u32 tid, u32 dst, u32 src, const_as<u64> add) {
  u32 lo[12], hi[12];
  u64 rc[12];
#pragma unroll
  for (int j = 0; j < 12; j++) rc[j] = add[j];
#pragma unroll
  for (int j = 0; j < 12; j++) { const u64 v = lds[(src + j) * T + tid]; lo[j] = (u32)v; hi[j] = (u32)(v >> 32); }
  pos_mds_h(lo, hi, rc);
#pragma unroll
  for (int j = 0; j < 12; j++) lds[(dst + j) * T + tid] = gl_canon(((u64)hi[j] << 32) | lo[j]);
}
__device__ __forceinline__ u64 q_sbox(u64 x) {
  u32 x0 = (u32)x, x1 = (u32)(x >> 32);
  pos_sbox_h(x0, x1);
  return gl_canon(((u64)x1 << 32) | x0);
}
#else  // host pass of hipcc: declarations only
__device__ void q_pmds(u64 *lds, u32 T, u32 tid, u32 dst, u32 src, const_as<u64> add);
__device__ u64 q_sbox(u64 x);
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// The running combination of a gate's constraints.  acc is kept LAZY (any u64 congruent to the value): a Horner step is a
// 17-instruction multiply-reduce plus a 3-instruction add of the canonical constraint value, instead of 21 + 8 for canonical
// arithmetic; whoever reads acc multiplies it with gl_mul, which accepts any u64 and returns a canonical value.
// The constraint combination of a native gate, sum_j alpha^(e_j) c_j, without a multiply-REDUCE per constraint (a Horner chain
// costs 17 + 3 instructions per constraint and challenge): alpha^e comes from a table as three 22-bit limbs, a constraint value
// (any u64: it need not even be canonical) is two 32-bit halves, and each of the six half x limb products - below 2^54 - is one
// v_mad_u64_u32 into a 64-bit column sum that 128 terms cannot overflow.  12 instructions per constraint and two challenges
// instead of 40; the columns are folded (sum_l (col_l + col_(3+l) 2^32) 2^(22 l) mod p) once per point.
struct QTerms {
  u64 col[QUOTIENT_MAX_CH][6];
  u32 CH;
  __device__ __forceinline__ void init(u32 ch) {
    CH = ch;
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
#pragma unroll
      for (u32 l = 0; l < 6; l++) col[c][l] = 0;
  }
  template <u32 E>  // x * alpha^E
  __device__ __forceinline__ void add(const QuotientArgs &a, u64 x) {
    static_assert(E < QUOTIENT_TERM_POWS, "too many constraints for the alpha power table");
    add(a, E, x);
  }
  __device__ __forceinline__ void add(const QuotientArgs &a, u32 e /* wave-uniform, < QUOTIENT_TERM_POWS */, u64 x) {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) {
      if (c < CH) {
        const_as<u32> L = konst(a.alpha_limbs) + ((size_t)c * QUOTIENT_TERM_POWS + e) * 4;
#pragma unroll
        for (u32 l = 0; l < 3; l++) {
          const u32 w = L[l];
          col[c][l] += (u64)x0 * w;
          col[c][3 + l] += (u64)x1 * w;
        }
      }
    }
  }
  __device__ __forceinline__ void fold(u64 out[QUOTIENT_MAX_CH]) const {
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) {
      if (c < CH) {
        u64 v[3];
#pragma unroll
        for (u32 l = 0; l < 3; l++) {  // col_l + col_(3+l) 2^32 as a 128-bit value
          const u64 lo = col[c][l] + (col[c][3 + l] << 32);
          const u64 hi = (col[c][3 + l] >> 32) + (lo < col[c][l] ? 1 : 0);
          v[l] = gl_reduce128(lo, hi);
        }
        out[c] = gl_add(v[0], gl_add(gl_shl<22>(v[1]), gl_shl<44>(v[2])));
      }
    }
  }
};
// The same for the generated gates, with the limb table in LDS (the kernel stages it: 4 KB) and no branch on the challenge count
// (with one challenge the second slot's limbs are zeros).  Why: a branch per constraint cuts an evaluator into hundreds of basic
// blocks, across which hipcc sinks every recomposition chain to its use at the end of the function, with all its wires alive until
// there; and in ONE block it hoists the ~600 scalar loads of a constant-space table to the top.  LDS reads are ordered by the
// window barriers of the generated code (Q_WINDOW_BARRIER) like the wire loads are, so the generator's schedule survives.
typedef const __attribute__((address_space(3))) u32 *lds_u32_ptr;
struct QTermsLds {
  u64 col[QUOTIENT_MAX_CH][6];
  lds_u32_ptr limbs;
  __device__ __forceinline__ void init() {
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
#pragma unroll
      for (u32 l = 0; l < 6; l++) col[c][l] = 0;
  }
  // The column sums are plain integer additions: in one basic block hipcc reassociates the whole sum of a gate's ~100 terms into
  // an order of its own, with every term (or its operands) alive until the end.  Passing the sums through an empty asm at the
  // window boundaries of the generated code cuts the expression trees there.
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
#pragma unroll
      for (u32 l = 0; l < 6; l++) asm volatile("" : "+v"(col[c][l]));
  }
  template <u32 E>  // x * alpha^E
  __device__ __forceinline__ void add(u64 x) {
    static_assert(E < QUOTIENT_TERM_POWS, "too many constraints for the alpha power table");
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) {
      lds_u32_ptr L = limbs + ((size_t)c * QUOTIENT_TERM_POWS + E) * 4;
#pragma unroll
      for (u32 l = 0; l < 3; l++) {
        const u32 w = L[l];
        col[c][l] += (u64)x0 * w;
        col[c][3 + l] += (u64)x1 * w;
      }
    }
  }
  __device__ __forceinline__ void fold(u64 out[QUOTIENT_MAX_CH]) const {
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) {
      u64 v[3];
#pragma unroll
      for (u32 l = 0; l < 3; l++) {
        const u64 lo = col[c][l] + (col[c][3 + l] << 32);
        const u64 hi = (col[c][3 + l] >> 32) + (lo < col[c][l] ? 1 : 0);
        v[l] = gl_reduce128(lo, hi);
      }
      out[c] = gl_add(v[0], gl_add(gl_shl<22>(v[1]), gl_shl<44>(v[2])));
    }
  }
};
// (an evaluator that uses the weighted terms never touches acc / step until finish_terms(): the compiler keeps only what is used)
struct QEmit {
  u64 acc[QUOTIENT_MAX_CH], step[QUOTIENT_MAX_CH];
  u32 CH, emitted;
  bool weighted = false;  // acc is the finished combination sum_j alpha^j c_j (no rescaling by the caller)
  QTerms t;
  QTermsLds tl;
  __device__ __forceinline__ void begin_terms_lds() { tl.init(); }
  __device__ __forceinline__ void finish_terms_lds() { tl.fold(acc); weighted = true; }
  __device__ __forceinline__ void begin_terms() { t.init(CH); }
  __device__ __forceinline__ void term(const QuotientArgs &a, u32 e, u64 x) { t.add(a, e, x); }  // x alpha^e; x any u64
  __device__ __forceinline__ void finish_terms() { t.fold(acc); weighted = true; }
  __device__ __forceinline__ void operator()(u64 x) {  // forward gates: Horner with 1 / alpha (step = 0 encodes alpha = 0)
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
      if (c < CH) acc[c] = step[c] == 0 ? (emitted ? acc[c] : x) : gl_add_nc(gl_mul_nc(acc[c], step[c]), x);
    emitted++;
  }
  __device__ __forceinline__ void horner(u64 x) {      // constraints listed last to first: Horner with alpha
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
      if (c < CH) acc[c] = gl_add_nc(gl_mul_nc(acc[c], step[c]), x);
  }
};
#endif

}  // namespace lcp2

// window boundary of a generated gate: no memory access (wire loads, LDS limb reads) and no instruction moves across it
#define Q_PIN(x) asm volatile("" : "+v"(x))
#define Q_WINDOW_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#include "generated_gates.hpp"   // the index: Q_GENERATED_COUNT, Q_GENERATED_WAVES (the evaluators themselves: generated_gates_<unit>.hpp)

namespace lcp2 {
static_assert(Q_GENERATED_COUNT == QUOTIENT_GENERATED_GATES, "prover_kernels.hpp and generated_gates.hpp disagree");
#if defined(__HIP_DEVICE_COMPILE__)
template <u32 K> __device__ __forceinline__ void q_generated(const QuotientArgs &a, u64 i, QEmit &emit);
#endif

// the alpha-limb table into LDS for the generated gates (every thread of the workgroup must call it: it ends in a barrier)
constexpr u32 Q_LIMB_WORDS32 = QUOTIENT_MAX_CH * QUOTIENT_TERM_POWS * 4;
__device__ __forceinline__ void q_stage_limbs(const QuotientArgs &a, u64 *lds, u32 T, u32 tid) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 *dst = (u32 *)(lds + a.limbs_lds_word);
  for (u32 j = tid; j < Q_LIMB_WORDS32; j += T) dst[j] = konst(a.alpha_limbs)[j];
  __syncthreads();
#endif
}

// ---- the pieces of "val[c] <- filter_g(point) * sum_i alpha_c^i constraint_{g,i}(point)" for gate g at the point whose operands sit at
// index i: q_emit_begin, then ONE evaluator (q_interpret, a native plonky2 gate, a generated program), then q_gate_finish.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void q_emit_begin(const QuotientArgs &a, const GateDev &G, QEmit &emit) {
  const u32 CH = a.num_challenges;
  const bool fwd = (G.flags & LCP2_GATE_EMIT_FORWARD) != 0;
  // Horner step with alpha (constraints listed last to first) or with 1 / alpha (first to last; rescaled in q_gate_finish).
  // alpha = 0 in a forward gate (step = 0): the sum is the first constraint alone.
  emit.CH = CH; emit.emitted = 0;
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) { emit.acc[c] = 0; emit.step[c] = c < CH ? (fwd ? konst(a.alpha_inv)[c] : konst(a.alphas)[c]) : 0; }
}
// the gate-program interpreter: registers in LDS (reg r of thread t at lds[r * T + t]), decode wave-uniform
__device__ __forceinline__ void q_interpret(const QuotientArgs &a, const GateDev &G, u64 i, u64 *lds, u32 T, u32 tid, QEmit &emit) {
  const bool fwd = (G.flags & LCP2_GATE_EMIT_FORWARD) != 0;
  // the instruction words are wave-uniform scalar loads: fetch one instruction ahead so that the scalar-cache round
  // trip overlaps the arithmetic of the current instruction (the code array is padded by one instruction)
  const_as<u64> code2 = konst((const u64 *)a.code);  // one instruction = two 32-bit words
  u64 nxt = code2[G.code_offset];
  for (u32 pc = G.code_offset; pc < G.code_offset + G.code_len; pc++) {
    const u32 w0 = (u32)nxt, w1 = (u32)(nxt >> 32);
    nxt = code2[pc + 1];
    const u32 op = w0 & 0xF, dst = (w0 >> 8) & 0xFF, ka = (w0 >> 16) & 0xF, kb = (w0 >> 20) & 0xF, ia = w1 & 0xFFFF, ib = w1 >> 16;
    if (op == QOP_LDG) { q_stage(a, dst, konst(a.stage_list) + w1, lds, T, tid, i); continue; }
    if (op == LCP2_OP_PMDS) { q_pmds(lds, T, tid, dst, ia, konst(a.imm) + ib); continue; }
    u64 x = q_operand(a, ka, ia, lds, T, tid, i);
    if (op == LCP2_OP_EMIT || op == LCP2_OP_EMITBOOL) {
      if (op == LCP2_OP_EMITBOOL) x = gl_sub(gl_mul(x, x), x);
      if (fwd) emit(x); else emit.horner(x);
      continue;
    }
    if (op == LCP2_OP_SBOX) { lds[dst * T + tid] = q_sbox(x); continue; }
    u64 y = q_operand(a, kb, ib, lds, T, tid, i);
    u64 r;
    switch (op) {  // uniform across the wave: the code stream is the same for every point
      case LCP2_OP_ADD: r = gl_add(x, y); break;
      case LCP2_OP_SUB: r = gl_sub(x, y); break;
      case LCP2_OP_MUL: r = gl_mul(x, y); break;
      case LCP2_OP_XOR: { const u64 xy = gl_mul(x, y); r = gl_sub(gl_sub(gl_add(x, y), xy), xy); break; }
      case LCP2_OP_DBLADD: r = gl_add(gl_add(x, x), y); break;
      default: r = gl_add(lds[dst * T + tid], gl_mul(x, y)); break;  // LCP2_OP_MULADD
    }
    lds[dst * T + tid] = r;
  }
}
__device__ __forceinline__ void q_gate_finish(const QuotientArgs &a, u32 g, const GateDev &G, u64 i, const QEmit &emit, u64 val[QUOTIENT_MAX_CH]) {
  const u32 CH = a.num_challenges;
  const bool fwd = (G.flags & LCP2_GATE_EMIT_FORWARD) != 0;
  const u64 s = a.consts[(u64)G.selector_index * a.stride + i];
  u64 f = 1;
  for (u32 j = G.group_start; j < G.group_end; j++)
    if (j != G.selector_value) f = gl_mul(f, gl_sub((u64)j, s));
  if (a.num_selectors > 1) f = gl_mul(f, gl_sub(0xFFFFFFFFull, s));
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
    if (c < CH) {
      const u64 sum = (fwd && !emit.weighted && emit.step[c] != 0) ? gl_mul(emit.acc[c], konst(a.gate_scale)[g * QUOTIENT_MAX_CH + c]) : emit.acc[c];
      val[c] = gl_mul(f, sum);
    }
}
#endif
__device__ __forceinline__ GateDev q_load_gate(const QuotientArgs &a, u32 g) {
  GateDev G;  // field by field: an address-space-qualified struct has no implicit copy
  const_as<u32> gw = konst((const u32 *)a.gates) + (size_t)g * (sizeof(GateDev) / 4);
  G.selector_index = gw[0]; G.selector_value = gw[1]; G.group_start = gw[2]; G.group_end = gw[3];
  G.code_offset = gw[4]; G.code_len = gw[5]; G.num_constraints = gw[6]; G.flags = gw[7];
  return G;
}
}  // namespace lcp2
