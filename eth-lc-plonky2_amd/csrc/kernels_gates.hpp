// Kernels of the generated straight-line gate evaluators (csrc/generated_gates_<unit>.hpp, tools/gen/gen_native_gates.cpp).  Every
// compile unit kernels_gates_<unit>.hip includes its own generated header and this file and defines, through
// LCP2_DEFINE_GENERATED_UNIT, the launcher of its programs; kernels_prover.hip asks the units one after the other
// (launch_generated).  One kernel per program and mode: the quotient values on the LDE points, the same evaluation over the rows of H
// (LCP2_E_UNSAT check), and the build()-time check of the LCP2_GATE_NATIVE_GENERATED claim against the interpreted program.
#pragma once
#include <utility>
#include "quotient_common.hpp"

namespace lcp2 {

enum GeneratedMode : u32 { GEN_QUOTIENT = 0, GEN_ROW_CHECK = 1, GEN_CLAIM_CHECK = 2 };

// CHECK = false: out[c][point] (+)= filter * constraints; CHECK = true: the rows of H, a wave skips a gate none of its rows holds
template <u32 K, bool CHECK>
__global__ __launch_bounds__(QUOTIENT_THREADS, Q_GENERATED_WAVES[K]) void k_q_gen(QuotientArgs a, u32 g, u32 accumulate, unsigned long long *flag) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 T = QUOTIENT_THREADS, tid = threadIdx.x;
  const u64 i0 = (u64)blockIdx.x * T + tid;
  q_stage_limbs(a, lds, T, tid);
  if (!CHECK && i0 >= a.count) return;            // no barrier is used below
  const u64 i = i0 < a.count ? i0 : a.count - 1;  // CHECK: the tail re-checks the last row so that every lane votes
  const GateDev G = q_load_gate(a, g);
  if (CHECK) {
    const u64 sel = a.consts[(u64)G.selector_index * a.stride + i];
    if (!__any(sel == G.selector_value)) return;
  }
  QEmit emit;
  q_emit_begin(a, G, emit);
  emit.tl.limbs = (lds_u32_ptr)(lds + a.limbs_lds_word);
  q_generated<K>(a, i, emit);
  u64 val[QUOTIENT_MAX_CH];
  q_gate_finish(a, g, G, i, emit, val);
  if (CHECK) {
    bool bad = false;
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
      if (c < a.num_challenges && val[c] != 0) bad = true;
    if (bad) atomicMin(flag, (unsigned long long)i + 1);
  } else {
    const u64 ig = a.leaf0 + i;
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
      if (c < a.num_challenges) a.out[(u64)c * a.N + ig] = accumulate ? gl_add(a.out[(u64)c * a.N + ig], val[c]) : val[c];
  }
#endif
}

// build()-time check of the claim: program (interpreted) against generated evaluator on random points in a.wires / a.consts
template <u32 K>
__global__ __launch_bounds__(QUOTIENT_THREADS, 2) void k_claim_check_gen(QuotientArgs a, u32 g, unsigned long long *flag) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 T = QUOTIENT_THREADS, tid = threadIdx.x;
  const u64 i = (u64)blockIdx.x * T + tid;
  q_stage_limbs(a, lds, T, tid);
  if (i >= a.count) return;
  const GateDev G = q_load_gate(a, g);
  u64 r0[QUOTIENT_MAX_CH], r1[QUOTIENT_MAX_CH];
  {
    QEmit emit;
    q_emit_begin(a, G, emit);
    q_interpret(a, G, i, lds, T, tid, emit);
    q_gate_finish(a, g, G, i, emit, r0);
  }
  {
    QEmit emit;
    q_emit_begin(a, G, emit);
    emit.tl.limbs = (lds_u32_ptr)(lds + a.limbs_lds_word);
    q_generated<K>(a, i, emit);
    q_gate_finish(a, g, G, i, emit, r1);
  }
  bool bad = false;
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
    if (c < a.num_challenges && r0[c] != r1[c]) bad = true;
  if (bad) atomicMin(flag, (unsigned long long)i + 1);
#endif
}

// a: the arguments of the launch as the caller built them (limbs_lds_word / dynamic LDS are set here)
template <u32 K>
void launch_generated_one(hipStream_t s, const QuotientArgs &a, u32 g, u32 accumulate, unsigned long long *flag, u32 mode) {
  const dim3 grid((unsigned)((a.count + QUOTIENT_THREADS - 1) / QUOTIENT_THREADS)), block(QUOTIENT_THREADS);
  const size_t limbs = Q_LIMB_WORDS32 * 4;  // the LDS copy of the alpha-limb table
  QuotientArgs ag = a;
  if (mode == GEN_CLAIM_CHECK) {  // behind the interpreter's registers and staging slots
    ag.limbs_lds_word = (a.num_regs + QUOTIENT_STAGE) * QUOTIENT_THREADS;
    const size_t lds = (size_t)(a.num_regs + QUOTIENT_STAGE) * QUOTIENT_THREADS * sizeof(u64) + limbs;
    hipLaunchKernelGGL((k_claim_check_gen<K>), grid, block, lds, s, ag, g, flag);
    return;
  }
  ag.limbs_lds_word = 0;
  if (mode == GEN_ROW_CHECK) hipLaunchKernelGGL((k_q_gen<K, true>), grid, block, limbs, s, ag, g, accumulate, flag);
  else hipLaunchKernelGGL((k_q_gen<K, false>), grid, block, limbs, s, ag, g, accumulate, flag);
}

template <u32 FIRST, size_t... I>
bool launch_generated_unit(std::index_sequence<I...>, hipStream_t s, const QuotientArgs &a, u32 k, u32 g, u32 accumulate, unsigned long long *flag, u32 mode) {
  bool hit = false;
  ((k == FIRST + I ? (launch_generated_one<FIRST + (u32)I>(s, a, g, accumulate, flag, mode), hit = true) : false), ...);
  return hit;
}

// true if program k belongs to this unit (and has been launched)
#define LCP2_DEFINE_GENERATED_UNIT(name, FIRST, COUNT)                                                                                          \
  bool launch_generated_##name(hipStream_t s, const QuotientArgs &a, u32 k, u32 g, u32 accumulate, unsigned long long *flag, u32 mode) {       \
    return launch_generated_unit<FIRST>(std::make_index_sequence<COUNT>{}, s, a, k, g, accumulate, flag, mode);                                 \
  }

bool launch_generated_sha(hipStream_t s, const QuotientArgs &a, u32 k, u32 g, u32 accumulate, unsigned long long *flag, u32 mode);
bool launch_generated_u32a(hipStream_t s, const QuotientArgs &a, u32 k, u32 g, u32 accumulate, unsigned long long *flag, u32 mode);
bool launch_generated_u32b(hipStream_t s, const QuotientArgs &a, u32 k, u32 g, u32 accumulate, unsigned long long *flag, u32 mode);
bool launch_generated_reca(hipStream_t s, const QuotientArgs &a, u32 k, u32 g, u32 accumulate, unsigned long long *flag, u32 mode);
bool launch_generated_recb(hipStream_t s, const QuotientArgs &a, u32 k, u32 g, u32 accumulate, unsigned long long *flag, u32 mode);

}  // namespace lcp2
