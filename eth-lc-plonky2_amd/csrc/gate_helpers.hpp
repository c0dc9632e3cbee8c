// Arithmetic helpers of the generated gate evaluators that are plain C++ (the CPU emulation of tests/emu/emu_gates.cpp runs the same code).
#pragma once
#include "gl64.hpp"

namespace lcp2 {

// One row of the Poseidon MDS layer on twelve values given as u64 (any u64: they are taken as 32-bit halves), plus a constant: the PMDS
// instruction of a gate program in a generated evaluator.  Row r = sum_i x[(i + r) % 12] CIRC[i] + x[r] DIAG[r] + c; the
// two half sums stay below 2^41, the value lo + hi 2^32 + c is folded once.
template <u32 R>
LCP2_HD u64 q_mds_row(u64 x0, u64 x1, u64 x2, u64 x3, u64 x4, u64 x5, u64 x6, u64 x7, u64 x8, u64 x9, u64 x10, u64 x11, u64 c) {
  constexpr u32 CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  const u64 x[12] = {x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11};
  u64 lo = 0, hi = 0;
#pragma unroll
  for (u32 i = 0; i < 12; i++) {
    const u64 v = x[(i + R) % 12];
    const u32 k = CIRC[i] + (i == 0 && R == 0 ? 8u : 0u);  // DIAG = [8, 0, ..., 0]
    lo += (u64)(u32)v * k;
    hi += (v >> 32) * k;
  }
  u64 l = lo + (hi << 32);
  u64 h = (hi >> 32) + (l < lo ? 1 : 0);
  const u64 l2 = l + c;
  h += l2 < l ? 1 : 0;
  return gl_reduce128(l2, h);
}

}  // namespace lcp2
