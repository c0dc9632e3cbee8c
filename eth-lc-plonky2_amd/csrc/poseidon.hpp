// Poseidon-12 over Goldilocks (x^7, 4 + 22 + 4 rounds), one permutation per lane.
//
// Replaces plonky2 0.1.4 hash/poseidon.rs + poseidon_goldilocks.rs
// (`PoseidonPermutation`, `PoseidonHash::{hash_no_pad, two_to_one, hash_or_noop}`;
// un-vendored dependency, /root/reference/Cargo.lock:2347-2350; reached through
// `PoseidonGoldilocksConfig`, eth-lc-plonky2/src/main.rs:75).
//
// The state lives in 24 VGPRs of one lane; round constants are wave-uniform
// (scalar loads).  Values stay "lazy" (any u64 congruent to the element) through
// the 30 rounds and are canonicalised once at the end.
#pragma once
#include <utility>
#include "gl64.hpp"

namespace lcp2 {

constexpr int POS_W = 12;
constexpr int POS_RATE = 8;
constexpr int POS_FULL_HALF = 4;
constexpr int POS_PARTIAL = 22;
constexpr int POS_ROUNDS = 30;

// ---- Partial rounds three at a time --------------------------------------------------------------------------------------
// In a partial round only element 0 goes through the S-box; elements 1..11 pass through the round linearly.  With u = state after the
// constant layer, y = u[1..12), w = u[0]^7 and the MDS matrix M cut into m00 = M[0][0], a = M[0][1..], b = M[1..][0], A = M[1..][1..]:
//     next u0 = a.y + m00 w + c0          next y = A y + b w + c[1..]
// so three rounds in a row are (w0, w1, w2 the three S-box outputs, u1, u2, u3 the element 0 after each round)
//     u1 = a.y            + m00 w0                          + k1          w1 = u1^7
//     u2 = (aA).y         + (a.b) w0   + m00 w1             + k2          w2 = u2^7
//     u3 = (aA^2).y       + (aAb) w0   + (a.b) w1 + m00 w2  + k3
//     y3 = A^3 y + A^2 b w0 + A b w1   + b w2               + kv
// with k1, k2, k3, kv[11] affine images of the three rounds' constants (computed once on the host, pos_extend_round_constants).
// The entries of A^3 stay below 2^21 (the MDS entries are at most 41; A^4 would overflow the 64-bit accumulators of the 32-bit
// halves), so every product is still ONE v_mad_u64_u32 per half: 386 of them and 14 folds for three rounds, against 3 x (290 + 12)
// for three plain MDS layers.  21 of the 22 partial rounds run as 7 such groups: 15.9 k VALU instructions per permutation instead
// of 20.5 k.  (plonky2's own "fast partial rounds" reach the same end with full 64-bit constants, which cost 17 instructions per
// product on this ISA and come out even; this form keeps the constants small.)
constexpr int POS_GROUP = 3;                        // partial rounds per group
constexpr int POS_GROUPS = POS_PARTIAL / POS_GROUP;  // 7 groups = 21 rounds, the 22nd runs as a plain round
constexpr int POS_GROUP_CONSTS = 3 + 11;            // k1, k2, k3, kv[11]
constexpr int POS_RC_WORDS = POS_ROUNDS * POS_W + POS_GROUPS * POS_GROUP_CONSTS;  // the device table: round constants, then the group constants

struct PosPartialTables {
  u32 m00, ab, aAb;
  u32 a[11], aA[11], aA2[11], b[11], Ab[11], A2b[11];
  u32 A[11][11], A3[11][11];
};
constexpr PosPartialTables pos_partial_tables() {
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  PosPartialTables t{};
  u64 M[12][12] = {};
  for (int r = 0; r < 12; r++)
    for (int c = 0; c < 12; c++) M[r][c] = C[(c - r + 12) % 12] + (r == 0 && c == 0 ? 8 : 0);
  u64 A[11][11] = {}, A2[11][11] = {}, A3[11][11] = {}, a[11] = {}, b[11] = {};
  for (int i = 0; i < 11; i++) {
    a[i] = M[0][1 + i];
    b[i] = M[1 + i][0];
    for (int j = 0; j < 11; j++) A[i][j] = M[1 + i][1 + j];
  }
  for (int i = 0; i < 11; i++)
    for (int j = 0; j < 11; j++) {
      u64 acc = 0;
      for (int k = 0; k < 11; k++) acc += A[i][k] * A[k][j];
      A2[i][j] = acc;
    }
  for (int i = 0; i < 11; i++)
    for (int j = 0; j < 11; j++) {
      u64 acc = 0;
      for (int k = 0; k < 11; k++) acc += A2[i][k] * A[k][j];
      A3[i][j] = acc;
    }
  t.m00 = (u32)M[0][0];
  u64 ab = 0, aAb = 0;
  for (int i = 0; i < 11; i++) {
    u64 Ab = 0, A2b = 0, aA = 0, aA2 = 0;
    for (int k = 0; k < 11; k++) { Ab += A[i][k] * b[k]; A2b += A2[i][k] * b[k]; aA += a[k] * A[k][i]; aA2 += a[k] * A2[k][i]; }
    t.a[i] = (u32)a[i]; t.b[i] = (u32)b[i]; t.Ab[i] = (u32)Ab; t.A2b[i] = (u32)A2b; t.aA[i] = (u32)aA; t.aA2[i] = (u32)aA2;
    ab += a[i] * b[i];
    aAb += a[i] * Ab;
    for (int j = 0; j < 11; j++) { t.A[i][j] = (u32)A[i][j]; t.A3[i][j] = (u32)A3[i][j]; }
  }
  t.ab = (u32)ab; t.aAb = (u32)aAb;
  return t;
}
constexpr u32 pos_partial_max_entry() {
  constexpr PosPartialTables t = pos_partial_tables();
  u32 m = t.aAb > t.ab ? t.aAb : t.ab;
  for (int i = 0; i < 11; i++) {
    if (t.A2b[i] > m) m = t.A2b[i];
    if (t.aA2[i] > m) m = t.aA2[i];
    for (int j = 0; j < 11; j++) if (t.A3[i][j] > m) m = t.A3[i][j];
  }
  return m;
}
// 14 terms of (32-bit half) x (entry) plus a 32-bit constant half must fit the 64-bit accumulator, and the fold needs sums below 2^59
static_assert(pos_partial_max_entry() < (1u << 22), "grouped partial rounds: a table entry is too large for one multiply-accumulate per half");

// x^7 on lazy (non-canonical) values: two squarings + two products, no canonicalisation in between
LCP2_HD u64 pos_sbox(u64 x) {
  u64 x2 = gl_sqr_nc(x);
  u64 x4 = gl_sqr_nc(x2);
  u64 x3 = gl_mul_nc(x, x2);
  return gl_mul_nc(x3, x4);
}

// MDS layer: out[r] = sum_i s[(i + r) % 12] * CIRC[i] + s[r] * DIAG[r],  CIRC = [17 15 41 16 2 28 13 13 39 18 34 20],
// DIAG = [8, 0, ...].  On gfx950 every VOP3 integer instruction issues in ~4.4 cycles, v_mad_u64_u32 included
// (tools/ubench/int_rates.hip), so the cheapest form is the one with the fewest instructions: the state is split
// into 32-bit halves, each half is accumulated in 64 bits with one v_mad_u64_u32 per product (sums < 2^41) and
// the two sums are folded with one reduction per lane.  (A 22/21/21-bit limb variant built on 24-bit multiplies
// was measured 1.4x SLOWER: 24-bit multiplies are not cheaper here and it needs 1.5x the instructions.)
// Result is lazy (any u64 congruent to the element).
LCP2_HD void pos_mds(u64 s[12]) {
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { lo[i] = (u32)s[i]; hi[i] = (u32)(s[i] >> 32); }
#pragma unroll
  for (int r = 0; r < 12; r++) {
    u64 al = 0, ah = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
      al += (u64)lo[(i + r) % 12] * C[i];
      ah += (u64)hi[(i + r) % 12] * C[i];
    }
    if (r == 0) { al += (u64)lo[0] * 8u; ah += (u64)hi[0] * 8u; }
    // al + ah * 2^32 with al, ah < 2^42:  = al + (ah_hi * 2^64) + (ah_lo << 32),  2^64 = 2^32 - 1 (mod p)
    u64 ah_hi = ah >> 32;
    u64 t = al + ((ah_hi << 32) - ah_hi);  // < 2^43
    u64 u = ah << 32;                      // (ah_lo << 32): the high half of ah is shifted out
    u64 v = t + u;
    s[r] = v < t ? v + GL_EPS : v;
  }
}

// Portable form (host: challenger, verifier, CPU emulation of the kernels).
// rc: 30 * 12 round constants, canonical.  s: any u64 values in, canonical out.
LCP2_HD void pos_permute_portable(u64 s[12], const u64 *__restrict__ rc) {
  int round = 0;
#pragma unroll 1
  for (int r = 0; r < POS_FULL_HALF; r++, round++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = pos_sbox(gl_add_nc(s[i], rc[round * 12 + i]));
    pos_mds(s);
  }
#pragma unroll 1
  for (int r = 0; r < POS_PARTIAL; r++, round++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_nc(s[i], rc[round * 12 + i]);
    s[0] = pos_sbox(s[0]);
    pos_mds(s);
  }
#pragma unroll 1
  for (int r = 0; r < POS_FULL_HALF; r++, round++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = pos_sbox(gl_add_nc(s[i], rc[round * 12 + i]));
    pos_mds(s);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}

// The group constants of partial rounds r, r+1, r+2 (r = 4 + 3 g), appended to the 360 round constants, with c1 / c2 / c3 the
// constants of rounds r+1 / r+2 / r+3 (the constant layer of a round rides in the accumulators of the layer before):
//   k1 = c1[0]   k2 = a.c1[1..] + c2[0]   k3 = a.(A c1[1..] + c2[1..]) + c3[0]   kv = A (A c1[1..] + c2[1..]) + c3[1..]
inline void pos_extend_round_constants(u64 rc[POS_RC_WORDS]) {
  constexpr PosPartialTables T = pos_partial_tables();
  for (int g = 0; g < POS_GROUPS; g++) {
    const int r = POS_FULL_HALF + POS_GROUP * g;
    const u64 *c1 = rc + 12 * (r + 1), *c2 = rc + 12 * (r + 2), *c3 = rc + 12 * (r + 3);
    u64 *k = rc + POS_ROUNDS * POS_W + POS_GROUP_CONSTS * g;
    u64 v2[11];
    u64 k2 = c2[0], k3 = c3[0];
    for (int i = 0; i < 11; i++) {
      u64 acc = c2[1 + i];
      for (int j = 0; j < 11; j++) acc = gl_add(acc, gl_mul(T.A[i][j], c1[1 + j]));
      v2[i] = acc;
      k2 = gl_add(k2, gl_mul(T.a[i], c1[1 + i]));
    }
    for (int i = 0; i < 11; i++) {
      u64 acc = c3[1 + i];
      for (int j = 0; j < 11; j++) acc = gl_add(acc, gl_mul(T.A[i][j], v2[j]));
      k[3 + i] = acc;
      k3 = gl_add(k3, gl_mul(T.a[i], v2[i]));
    }
    k[0] = c1[0]; k[1] = k2; k[2] = k3;
  }
}

// The grouped form in portable arithmetic (what the gfx950 form computes, on canonical values): tests/emu checks it against the
// plain definition above and the oracle.  rc: POS_RC_WORDS words (pos_derive_round_constants + pos_extend_round_constants).
LCP2_HD void pos_permute_grouped_portable(u64 s[12], const u64 *__restrict__ rc) {
  constexpr PosPartialTables T = pos_partial_tables();
  for (int i = 0; i < 12; i++) s[i] = gl_add(gl_canon(s[i]), rc[i]);
  int round = 0;
  for (int r = 0; r < POS_FULL_HALF; r++, round++) {
    for (int i = 0; i < 12; i++) s[i] = gl_canon(pos_sbox(s[i]));
    pos_mds(s);
    for (int i = 0; i < 12; i++) s[i] = gl_add(gl_canon(s[i]), rc[(round + 1) * 12 + i]);
  }
  for (int g = 0; g < POS_GROUPS; g++, round += POS_GROUP) {
    const u64 *k = rc + POS_ROUNDS * POS_W + POS_GROUP_CONSTS * g;
    const u64 *y = s + 1;
    const u64 w0 = gl_canon(pos_sbox(s[0]));
    u64 u1 = gl_add(k[0], gl_mul(T.m00, w0)), u2 = gl_add(k[1], gl_mul(T.ab, w0)), u3 = gl_add(k[2], gl_mul(T.aAb, w0));
    for (int j = 0; j < 11; j++) {
      u1 = gl_add(u1, gl_mul(T.a[j], y[j]));
      u2 = gl_add(u2, gl_mul(T.aA[j], y[j]));
      u3 = gl_add(u3, gl_mul(T.aA2[j], y[j]));
    }
    const u64 w1 = gl_canon(pos_sbox(u1));
    u2 = gl_add(u2, gl_mul(T.m00, w1));
    const u64 w2 = gl_canon(pos_sbox(u2));
    u3 = gl_add(gl_add(u3, gl_mul(T.ab, w1)), gl_mul(T.m00, w2));
    u64 ny[11];
    for (int i = 0; i < 11; i++) {
      u64 acc = gl_add(k[3 + i], gl_add(gl_mul(T.A2b[i], w0), gl_add(gl_mul(T.Ab[i], w1), gl_mul(T.b[i], w2))));
      for (int j = 0; j < 11; j++) acc = gl_add(acc, gl_mul(T.A3[i][j], y[j]));
      ny[i] = acc;
    }
    s[0] = u3;
    for (int i = 0; i < 11; i++) s[1 + i] = ny[i];
  }
  for (; round < POS_FULL_HALF + POS_PARTIAL; round++) {  // the partial round the groups leave over
    s[0] = gl_canon(pos_sbox(s[0]));
    pos_mds(s);
    for (int i = 0; i < 12; i++) s[i] = gl_add(gl_canon(s[i]), rc[(round + 1) * 12 + i]);
  }
  for (int r = 0; r < POS_FULL_HALF; r++, round++) {
    for (int i = 0; i < 12; i++) s[i] = gl_canon(pos_sbox(s[i]));
    pos_mds(s);
    for (int i = 0; i < 12; i++) s[i] = round + 1 < POS_ROUNDS ? gl_add(gl_canon(s[i]), rc[(round + 1) * 12 + i]) : gl_canon(s[i]);
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---------------------------------------------------------------------------------------------------------
// gfx950 form.  Measured (tools/ubench): every VOP3-encoded or vcc-touching integer op (v_mad_u64_u32, add / sub with carry,
// v_cndmask) costs ~4.3 cycles per wave-instruction, plain 32-bit VOP1/VOP2 ops ~2.5.  So the state is kept as 32-bit halves,
// products and conditional corrections are v_mad_u64_u32 / v_mad_i64_i32 (gl_mul_halves, gl64.hpp: 12 + 2 moves per multiply; the
// compiler's own lowering of a 64x64 multiply + reduction is 27).  hipcc pads nothing inside an asm string, so the wait states it
// emits itself for the same pairs on gfx950 (a VALU write of vcc -> any VALU read of it: 2) are written out as s_nop 1 here;
// tools/check_hazards.py checks the built library.
// All values are lazy (any u64 congruent to the element); the final state is canonicalised.

__device__ __forceinline__ void pos_sbox_h(u32 &x0, u32 &x1) {
  u32 a0, a1, b0, b1, c0, c1;
  gl_mul_halves(x0, x1, x0, x1, a0, a1);  // x^2
  gl_mul_halves(a0, a1, a0, a1, b0, b1);  // x^4
  gl_mul_halves(x0, x1, a0, a1, c0, c1);  // x^3
  gl_mul_halves(c0, c1, b0, b1, x0, x1);  // x^7
}

// al + ah * 2^32 (al, ah < 2^42) -> lazy 64-bit value:  t = al + ah_hi * (2^64 mod p) ;  v = t + (ah_lo << 32), on carry += 2^32 - 1
// (as a multiply-add of the carry: v < 2^43 after a wrap, so no second carry)
__device__ __forceinline__ void pos_fold_h(u64 al, u64 ah, u32 &r0, u32 &r1) {
  u64 t = (u64)(u32)(ah >> 32) * 0xFFFFFFFFu + al;
  u32 t1 = (u32)(t >> 32), c;
  asm("v_add_co_u32 %0, vcc, %0, %2\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %1, 0, 1, vcc"
      : "+v"(t1), "=&v"(c) : "v"((u32)ah) : "vcc");
  t = ((u64)t1 << 32) | (u32)t;
  u64 q;
  asm("v_mad_u64_u32 %0, vcc, %1, -1, %2" : "=v"(q) : "v"(c), "v"(t) : "vcc");
  r0 = (u32)q;
  r1 = (u32)(q >> 32);
}

// One row of the MDS layer: al = k_lo + sum_i lo_i C_i, ah = k_hi + sum_i hi_i C_i as 24 v_mad_u64_u32 with the constants inline, the
// two chains interleaved.  Left to itself hipcc turns the constants 2, 8 and 16 into 64-bit shift-adds, for which the element has to be
// zero-extended into a register pair (two moves and a VOP3 op instead of one multiply-add), and it adds the round constant with a separate
// 64-bit add at the END of each chain; here the wave-uniform round constant is the addend of the FIRST multiply-add (an SGPR pair: one
// constant-bus read next to the inline constant).  One asm statement per row: between separate statements hipcc pads every dependent
// pair with a wait state it cannot know to be unnecessary.  x0..x11: the row's elements in the order of the circulant's constants
// 17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20 (row 0: 25 first - the diagonal's 8).
#define LCP2_POS_ROW_OPERANDS                                                                                                        \
  : "=&v"(al), "=&v"(ah)                                                                                                             \
  : "v"(l[0]), "v"(l[1]), "v"(l[2]), "v"(l[3]), "v"(l[4]), "v"(l[5]), "v"(l[6]), "v"(l[7]), "v"(l[8]), "v"(l[9]), "v"(l[10]), "v"(l[11]), \
    "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(h[4]), "v"(h[5]), "v"(h[6]), "v"(h[7]), "v"(h[8]), "v"(h[9]), "v"(h[10]), "v"(h[11]), \
    "s"(klo), "s"(khi)                                                                                                               \
  : "vcc"
template <bool ROW0>
__device__ __forceinline__ void pos_mds_row(const u32 (&l)[12], const u32 (&h)[12], u64 klo /* wave-uniform */, u64 khi, u64 &al, u64 &ah) {
  if constexpr (ROW0) {
    asm(
        "v_mad_u64_u32 %0, vcc, %2, 25, %26\n\t"
        "v_mad_u64_u32 %1, vcc, %14, 25, %27\n\t"
        "v_mad_u64_u32 %0, vcc, %3, 15, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %15, 15, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %4, 41, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %16, 41, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %5, 16, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %17, 16, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %6, 2, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %18, 2, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %7, 28, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %19, 28, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %8, 13, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %20, 13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %9, 13, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %21, 13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, 39, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %22, 39, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %11, 18, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %23, 18, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %12, 34, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %24, 34, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %13, 20, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %25, 20, %1"
        LCP2_POS_ROW_OPERANDS);
  } else {
    asm(
        "v_mad_u64_u32 %0, vcc, %2, 17, %26\n\t"
        "v_mad_u64_u32 %1, vcc, %14, 17, %27\n\t"
        "v_mad_u64_u32 %0, vcc, %3, 15, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %15, 15, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %4, 41, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %16, 41, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %5, 16, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %17, 16, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %6, 2, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %18, 2, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %7, 28, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %19, 28, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %8, 13, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %20, 13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %9, 13, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %21, 13, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %10, 39, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %22, 39, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %11, 18, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %23, 18, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %12, 34, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %24, 34, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %13, 20, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %25, 20, %1"
        LCP2_POS_ROW_OPERANDS);
  }
}
#undef LCP2_POS_ROW_OPERANDS

// x * C + k with the wave-uniform k as the addend of the multiply-add (C inline): the first term of an accumulator chain whose other
// constants are too large to be inline (the grouped partial rounds): saves the separate 64-bit add of k
template <u32 C>
__device__ __forceinline__ u64 pos_mac_first(u32 x, u64 k /* wave-uniform */) {
  static_assert(C <= 64, "inline constants only");
  u64 r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(x), "n"(C), "s"(k) : "vcc");
  return r;
}
// Accumulator chains of the grouped partial rounds, whose constants (entries of A^2, A^3 < 2^21) are too large to be inline and sit in
// SGPRs: the chain is ONE dependent sequence of multiply-adds, the wave-uniform round constant k the addend of the first one (whose
// constant C0 is inline: two SGPR operands in one instruction are not encodable).  As C, hipcc splits every chain in two, starts both
// from zero and joins them and k with 64-bit adds.
template <u32 C0>
__device__ __forceinline__ u64 pos_chain_first7(u64 k /* wave-uniform */, u32 x0, const u32 (&x)[6], const u32 (&c)[6]) {
  static_assert(C0 <= 64, "inline constants only");
  u64 acc;
  asm(
      "v_mad_u64_u32 %0, vcc, %1, %8, %15\n\t"
      "v_mad_u64_u32 %0, vcc, %2, %9, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %3, %10, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %4, %11, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %12, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %6, %13, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %7, %14, %0"
      : "=&v"(acc)
      : "v"(x0), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]),
        "n"(C0), "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]), "s"(c[5]), "s"(k)
      : "vcc");
  return acc;
}
template <int N>
__device__ __forceinline__ void pos_chain_add(u64 &acc, const u32 (&x)[N], const u32 (&c)[N]) {
  static_assert(N == 6 || N == 7, "six or seven terms");
  if constexpr (N == 6) {
    asm(
      "v_mad_u64_u32 %0, vcc, %1, %7, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %2, %8, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %3, %9, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %4, %10, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %11, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %6, %12, %0"
        : "+v"(acc)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]), "s"(c[5])
        : "vcc");
  } else {
    asm(
      "v_mad_u64_u32 %0, vcc, %1, %8, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %2, %9, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %3, %10, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %4, %11, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %12, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %6, %13, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %7, %14, %0"
        : "+v"(acc)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]),
          "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]), "s"(c[5]), "s"(c[6])
        : "vcc");
  }
}
template <class F, int... I>
__device__ __forceinline__ void pos_static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void pos_static_for(F &&f) { pos_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// state <- MDS(state) + add[0..12) ; add = the next round's constants, WAVE-UNIFORM (or nullptr): the constants ride in the
// first multiply-add of the accumulators, so the add-round-constant layer costs nothing.
__device__ __forceinline__ void pos_mds_h(u32 lo[12], u32 hi[12], const u64 *__restrict__ add) {
  u32 nl[12], nh[12];
#pragma unroll
  for (int r = 0; r < 12; r++) {
    const u64 k = add ? add[r] : 0;
    u32 l[12], h[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { l[i] = lo[(i + r) % 12]; h[i] = hi[(i + r) % 12]; }
    u64 al, ah;
    if (r == 0) pos_mds_row<true>(l, h, (u64)(u32)k, k >> 32, al, ah);
    else pos_mds_row<false>(l, h, (u64)(u32)k, k >> 32, al, ah);
    pos_fold_h(al, ah, nl[r], nh[r]);
  }
#pragma unroll
  for (int r = 0; r < 12; r++) { lo[r] = nl[r]; hi[r] = nh[r]; }
}

// Three partial rounds (see the tables above): state = u after the constant layer of the first of them, on return u of the round
// after the third.  kc: the group's 14 constants.  next_w(i, ul, uh): element 0 after round i (1, 2) comes in as lazy halves and
// leaves as the S-box output that enters the following round - the permutation raises it to the 7th power, the PoseidonGate
// evaluator of the quotient (kernels_prover.hip) emits its difference to the gate's S-box wire and continues from the wire.
template <class Consts /* pointer to the 14 constants, any address space */, class NextW>
__device__ __forceinline__ void pos_partial3_core(u32 lo[12], u32 hi[12], Consts kc, NextW next_w) {
  constexpr PosPartialTables T = pos_partial_tables();
  u32 w0l = lo[0], w0h = hi[0], w1l, w1h, w2l, w2h;
  pos_sbox_h(w0l, w0h);
  {  // u1 = row 0 of the MDS layer on (w0, y): constants 25, 15, 41, 16, 2, ... all inline (pos_mds_row)
    const u64 c = kc[0];
    u32 l[12], h[12];
    l[0] = w0l; h[0] = w0h;
#pragma unroll
    for (int j = 1; j < 12; j++) { l[j] = lo[j]; h[j] = hi[j]; }
    u64 al, ah;
    pos_mds_row<true>(l, h, (u64)(u32)c, c >> 32, al, ah);
    pos_fold_h(al, ah, w1l, w1h);
    next_w(1, w1l, w1h);
  }
  // rows u2, u3 and y3: k + (the S-box output that arrives last) * (inline constant) first, then y[0..6) ; y[6..11) and the earlier S-box outputs
  auto row = [&](auto c0, u64 k, u32 xl0, u32 xh0, const u32 (&ca)[6], const u32 (&cb5)[5], u32 cw0, u32 cw1, bool has_w1, u32 &rl, u32 &rh) {
    constexpr u32 C0 = decltype(c0)::value;
    const u32 xa_l[6] = {lo[1], lo[2], lo[3], lo[4], lo[5], lo[6]}, xa_h[6] = {hi[1], hi[2], hi[3], hi[4], hi[5], hi[6]};
    u64 al = pos_chain_first7<C0>((u64)(u32)k, xl0, xa_l, ca), ah = pos_chain_first7<C0>(k >> 32, xh0, xa_h, ca);
    if (has_w1) {
      const u32 xb_l[7] = {lo[7], lo[8], lo[9], lo[10], lo[11], w0l, w1l}, xb_h[7] = {hi[7], hi[8], hi[9], hi[10], hi[11], w0h, w1h};
      const u32 cb[7] = {cb5[0], cb5[1], cb5[2], cb5[3], cb5[4], cw0, cw1};
      pos_chain_add<7>(al, xb_l, cb);
      pos_chain_add<7>(ah, xb_h, cb);
    } else {
      const u32 xb_l[6] = {lo[7], lo[8], lo[9], lo[10], lo[11], w0l}, xb_h[6] = {hi[7], hi[8], hi[9], hi[10], hi[11], w0h};
      const u32 cb[6] = {cb5[0], cb5[1], cb5[2], cb5[3], cb5[4], cw0};
      pos_chain_add<6>(al, xb_l, cb);
      pos_chain_add<6>(ah, xb_h, cb);
    }
    pos_fold_h(al, ah, rl, rh);
  };
  {
    const u32 ca[6] = {T.aA[0], T.aA[1], T.aA[2], T.aA[3], T.aA[4], T.aA[5]}, cb5[5] = {T.aA[6], T.aA[7], T.aA[8], T.aA[9], T.aA[10]};
    row(std::integral_constant<u32, T.m00>{}, kc[1], w1l, w1h, ca, cb5, T.ab, 0, false, w2l, w2h);
    next_w(2, w2l, w2h);
  }
  u32 nl[12], nh[12];
  {
    const u32 ca[6] = {T.aA2[0], T.aA2[1], T.aA2[2], T.aA2[3], T.aA2[4], T.aA2[5]}, cb5[5] = {T.aA2[6], T.aA2[7], T.aA2[8], T.aA2[9], T.aA2[10]};
    row(std::integral_constant<u32, T.m00>{}, kc[2], w2l, w2h, ca, cb5, T.aAb, T.ab, true, nl[0], nh[0]);
  }
  pos_static_for<11>([&](auto ii) {
    constexpr int i = decltype(ii)::value;
    const u32 ca[6] = {T.A3[i][0], T.A3[i][1], T.A3[i][2], T.A3[i][3], T.A3[i][4], T.A3[i][5]},
              cb5[5] = {T.A3[i][6], T.A3[i][7], T.A3[i][8], T.A3[i][9], T.A3[i][10]};
    row(std::integral_constant<u32, T.b[i]>{}, kc[3 + i], w2l, w2h, ca, cb5, T.A2b[i], T.Ab[i], true, nl[1 + i], nh[1 + i]);  // b[i] = M[1 + i][0] <= 41
  });
#pragma unroll
  for (int r = 0; r < 12; r++) { lo[r] = nl[r]; hi[r] = nh[r]; }
}
__device__ __forceinline__ void pos_partial3_h(u32 lo[12], u32 hi[12], const u64 *__restrict__ kc) {
  pos_partial3_core(lo, hi, kc, [&](int, u32 &ul, u32 &uh) { pos_sbox_h(ul, uh); });
}

// rc: POS_RC_WORDS words (the 360 round constants, then the group constants of the partial rounds)
__device__ __forceinline__ void pos_permute_gfx950(u64 s[12], const u64 *__restrict__ rc) {
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { u64 v = gl_add_nc(s[i], rc[i]); lo[i] = (u32)v; hi[i] = (u32)(v >> 32); }
  int round = 0;
#pragma unroll 1
  for (int r = 0; r < POS_FULL_HALF; r++, round++) {
#pragma unroll
    for (int i = 0; i < 12; i++) pos_sbox_h(lo[i], hi[i]);
    pos_mds_h(lo, hi, rc + (round + 1) * 12);
  }
#pragma unroll 1
  for (int g = 0; g < POS_GROUPS; g++, round += POS_GROUP) pos_partial3_h(lo, hi, rc + POS_ROUNDS * POS_W + POS_GROUP_CONSTS * g);
#pragma unroll 1
  for (; round < POS_FULL_HALF + POS_PARTIAL; round++) {
    pos_sbox_h(lo[0], hi[0]);
    pos_mds_h(lo, hi, rc + (round + 1) * 12);
  }
#pragma unroll 1
  for (int r = 0; r < POS_FULL_HALF - 1; r++, round++) {
#pragma unroll
    for (int i = 0; i < 12; i++) pos_sbox_h(lo[i], hi[i]);
    pos_mds_h(lo, hi, rc + (round + 1) * 12);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) pos_sbox_h(lo[i], hi[i]);
  pos_mds_h(lo, hi, nullptr);
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = gl_canon(((u64)hi[i] << 32) | lo[i]);
}

// ---- lane-cooperative permutation: a 16-lane group holds one state, lane j < 12 owns element j (lanes 12..15 ride
// along).  The S-box layer runs on all lanes at once and the circulant MDS layer is 11 wavefront shuffles
// (ds_bpermute) per half-word: ~4 k dependent instructions per permutation instead of ~20 k, so a permutation takes
// ~12 us instead of ~58 us.  Throughput per CU is ~3x lower than one permutation per lane (the partial rounds keep 11
// of 12 lanes idle), so this form is used where the work is latency bound: Merkle levels and FRI layers with at most
// POS_COOP_MAX_NODES nodes.  v: element j of the state (any u64); returns the canonical output element j.
__device__ __forceinline__ u64 pos_permute_coop(u64 v, u32 j, const u64 *__restrict__ rc /* LDS copy of the round constants */) {
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  const u32 lane = __lane_id(), jj = j < 12 ? j : 0, base = lane - j;
  u32 src[12];
#pragma unroll
  for (int i = 0; i < 12; i++) src[i] = (base + (jj + i >= 12 ? jj + i - 12 : jj + i)) << 2;
  v = gl_add_nc(v, rc[jj]);
  u32 lo = (u32)v, hi = (u32)(v >> 32);
  auto mds = [&](u64 add) {
    u64 al = (u32)add, ah = add >> 32;
    al += (u64)lo * C[0];
    ah += (u64)hi * C[0];
#pragma unroll
    for (int i = 1; i < 12; i++) {
      const u32 vl = (u32)__builtin_amdgcn_ds_bpermute((int)src[i], (int)lo), vh = (u32)__builtin_amdgcn_ds_bpermute((int)src[i], (int)hi);
      al += (u64)vl * C[i];
      ah += (u64)vh * C[i];
    }
    if (j == 0) { al += (u64)lo * 8u; ah += (u64)hi * 8u; }
    pos_fold_h(al, ah, lo, hi);
  };
  int round = 0;
#pragma unroll 1
  for (int r = 0; r < POS_FULL_HALF; r++, round++) {
    const u64 next = rc[(round + 1) * 12 + jj];
    pos_sbox_h(lo, hi);
    mds(next);
  }
#pragma unroll 1
  for (int r = 0; r < POS_PARTIAL; r++, round++) {
    const u64 next = rc[(round + 1) * 12 + jj];
    u32 s0 = lo, s1 = hi;
    pos_sbox_h(s0, s1);
    if (j == 0) { lo = s0; hi = s1; }
    mds(next);
  }
#pragma unroll 1
  for (int r = 0; r < POS_FULL_HALF; r++, round++) {
    const u64 next = round + 1 < POS_ROUNDS ? rc[(round + 1) * 12 + jj] : 0;
    pos_sbox_h(lo, hi);
    mds(next);
  }
  return gl_canon(((u64)hi << 32) | lo);
}
#elif defined(__HIPCC__)
__device__ u64 pos_permute_coop(u64 v, u32 j, const u64 *__restrict__ rc);  // host pass of hipcc: declaration only
#endif
constexpr u32 POS_COOP_MAX_NODES = 4096;

LCP2_HD void pos_permute(u64 s[12], const u64 *__restrict__ rc) {
#if defined(__HIP_DEVICE_COMPILE__)
  pos_permute_gfx950(s, rc);
#else
  pos_permute_portable(s, rc);
#endif
}

// ---- round-constant derivation (host side; uploaded to the device once) ----
// ChaCha8Rng::seed_from_u64(0) sampled with rand-0.8 uniform [0, p): SURVEY App. A.3.
inline void pos_derive_round_constants(u64 out[POS_ROUNDS * POS_W]) {
  auto rotl = [](u32 x, int r) { return (u32)((x << r) | (x >> (32 - r))); };
  u32 key[8];
  u64 st = 0;
  for (int i = 0; i < 8; i++) {
    st = st * 6364136223846793005ull + 11634580027462260723ull;
    u32 xs = (u32)(((st >> 18) ^ st) >> 27);
    int rot = (int)(st >> 59);
    key[i] = rot ? (u32)((xs >> rot) | (xs << (32 - rot))) : xs;
  }
  u32 blk[16];
  int pos = 16, got = 0;
  u64 ctr = 0;
  while (got < POS_ROUNDS * POS_W) {
    u32 w[2];
    for (int k = 0; k < 2; k++) {
      if (pos == 16) {
        u32 in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; i++) in[4 + i] = key[i];
        in[12] = (u32)ctr; in[13] = (u32)(ctr >> 32); in[14] = 0; in[15] = 0;
        ctr++;
        u32 x[16];
        for (int i = 0; i < 16; i++) x[i] = in[i];
        auto qr = [&](int a, int b, int c, int d) {
          x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 16);
          x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 12);
          x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 8);
          x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 7);
        };
        for (int r = 0; r < 4; r++) {
          qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
          qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) blk[i] = x[i] + in[i];
        pos = 0;
      }
      w[k] = blk[pos++];
    }
    u64 v = (u64)w[0] | ((u64)w[1] << 32);
    unsigned __int128 m = (unsigned __int128)v * GL_P;
    if ((u64)m <= 0xFFFFFFFF00000000ull) out[got++] = (u64)(m >> 64);
  }
}

}  // namespace lcp2
