// Multi-pass NTT over Goldilocks, LDS-staged with radix-8/16 register steps, written for gfx950.
//
// Replaces plonky2_field 0.1.1 fft.rs (`fft_with_options`, `ifft_with_options`),
// polynomial/mod.rs (`lde`, `coset_fft`, `coset_ifft`) as used by
// `PolynomialBatch::from_values` / `from_coeffs`, `compute_quotient_polys` and
// `fri_committed_trees` inside `data.prove(...)` (un-vendored plonky2 0.1.4,
// /root/reference/Cargo.lock:2347-2350,2425-2427; reference call site
// eth-lc-plonky2/src/main.rs:230).
//
// Decomposition (Cooley-Tukey index splitting, "four-step" generalised to any
// number of passes).  The log2(n) index bits are cut into groups from the top:
//     [ g_1 | g_2 | ... | g_m ]            g_m is the contiguous (low) group.
// A pass transforms one group for every value of the other bits.  A workgroup
// owns a SLAB of 2^L elements staged in LDS:
//     local index = ( batch | t : B bits | ls : S bits )
//     global index = ls | seg << S | t << g_lo | (wg_hi,batch) << (g_lo + B)
// so every global access is a run of 2^S consecutive elements (S = 4 -> one
// 128-byte line) and, in the last pass, fully contiguous.
//   forward  (DIF): natural in  -> bit-reversed out; passes run g_1 .. g_m;
//                   after the butterflies of group g the element is multiplied
//                   by w_{N_g}^(l * bitrev(t'))  (N_g = 2^(g_lo+B), l = bits below)
//   inverse  (DIT): bit-reversed in -> natural out; passes run g_m .. g_1 with
//                   the conjugate twiddle applied at load.
// The forward output order is exactly the Merkle leaf order of plonky2
// (`reverse_index_bits_in_place` after the LDE), so no transpose pass exists.
//
// Inside a slab the B stages of the group run as REGISTER STEPS of r <= 4 stages: a thread pulls the 2^r
// elements that differ in r consecutive index bits out of LDS, runs the 2^r-point transform in registers
// (the constant twiddles w_16^k are powers of two: shifts), applies the step twiddle w_{N'}^(m * bitrev(j)) (N' = size of
// the sub-transform the step starts, m = index bits below the step) and puts them back: 13 stages cost
// 4 LDS round trips and barriers instead of 13, and the index arithmetic is paid once per 2^r elements.
// LDS words are padded by one per 16 (lds_phys) so that the bottom step, where each thread owns 16
// consecutive elements, is bank-conflict free; every other step reads runs of consecutive words.
//
// The pass body is split into load / step / store phases that take an explicit
// thread id: the HIP kernel calls them with a barrier in between, the CPU
// emulation harness (tests/emu) calls them in loops.
#pragma once
#include "gl64.hpp"

namespace lcp2 {

constexpr u32 NTT_MAX_L = 13;       // slab = 2^13 elements = 64 KiB of LDS
constexpr u32 NTT_THREADS = 512;    // 2 workgroups per CU (2 x 68 KiB of LDS), 16 elements of the slab per thread
constexpr u32 NTT_MAX_STEPS = 8;
constexpr u32 NTT_MAX_STRIDED_B = 9;
constexpr u32 NTT_SEG_BITS = 4;     // 16 x 8 B = 128-byte runs in strided passes
constexpr u32 NTT_BATCH = 8;        // elements in flight per thread in the load and store phases (4: 3 % slower LDE; 16: spills)

struct TwoLevelTable {  // value(e) = lo[e & (2^h - 1)] * hi[e >> h];  h = NTT_DIRECT: one level, value(e) = lo[e]
  const u64 *lo;
  const u64 *hi;
  u32 h;
};
constexpr u32 NTT_DIRECT = 63;
constexpr u32 NTT_DIRECT_MAX_LG = 22;
// Coset scale of the prefetching kernel: computed from one table value per thread (FMODE 2) or read from the one-level table
// (FMODE 1).  Measured on MI355X (43 columns of 2^22, rate 8; profiles/r03_lde_ab.md): computed 16.46 ms and 2.4 GB fetched by the
// first pass, table 16.97 ms and 6.5 GB.
constexpr bool NTT_PF_COMPUTED_SCALE = true;  // one-level tables up to 2^22 entries (32 MiB) per table row

struct NttPassParams {
  const u64 *in;
  u64 *out;
  u64 in_col_stride, out_col_stride;  // elements, per blockIdx.y
  u64 in_z_stride, out_z_stride;      // elements, per blockIdx.z (coset); out uses bitrev(z, zbits)
  u32 zbits;
  u32 z_base;                         // coset index of blockIdx.z = 0 (coset-sharded LDE: one launch per leaf block)
  u32 out_block_base;                 // leaf block that sits at offset 0 of `out` (a rank holds blocks [base, base + count))
  u32 L, S, B, g_lo;                  // slab bits, run bits, group bits, bits below the group
  const u64 *group_tw;                // w_{2^B}^e (or its inverse), e < 2^B: step twiddles
  u32 xcd_group;                      // LDE first pass: the 2^zbits coset transforms of one slab run back to back on one XCD (kernels_ntt.hip)
  u32 nsteps;
  u32 step_plan;                      // stages per register step, 4 bits each, listed from the top bits of the group down
  TwoLevelTable tw;                   // w_{N_g}^e (or inverse); used when g_lo > 0
  u32 scale_mode;                     // 0 none, 1 scalar, 2 two-level table (per z)
  u64 scale_scalar;
  TwoLevelTable sc;                   // forward: applied at load; inverse: at store
  u64 sc_lo_z_stride, sc_hi_z_stride;
  u32 canonical_in;                   // the input is known to be canonical (it is the output of an earlier pass)
  const u64 *sc_step;                 // [z][16]: shift_z^(k * 2^(g_lo + 5)), the ratio between a thread's elements in the fused first step
                                      // of k_ntt_pass_pf (nullptr: the factors come from the one-level table)
};

LCP2_HD u32 lds_phys(u32 i) { return i + (i >> 4); }
LCP2_HD u32 ntt_lds_words(u32 L) { return (1u << L) + ((1u << L) >> 4) + 1; }

// The 16th roots of unity of the field are powers of two: 2 has order 192 (2^96 = -1) and plonky2's w_16 (the generator's
// power) is 2^156 = -2^60, so w_16^k = +-2^s with s < 96 for every k: the twiddle multiply of a register butterfly is a
// shift-reduce (gl_shl: 10-12 instructions against 21 for a general multiply) and the sign goes into the order of the
// subtraction.  tests/emu and the GPU parity tests compare the transforms with the oracle's textbook radix-2 NTT.
//   forward  w^k : k=1 -2^60  k=2 -2^24  k=3 2^84   k=4  2^48  k=5  2^12  k=6 -2^72  k=7 -2^36
//   inverse w^-k : k=1  2^36  k=2  2^72  k=3 -2^12  k=4 -2^48  k=5 -2^84  k=6  2^24  k=7  2^60
// The butterfly's canonical a + b and a - b.  Device: a - b is four instructions (the borrow selects -15, and v_mad_i64_i32 adds
// -15 * 0x11111111 = -(2^32 - 1): the difference + p mod 2^64; see gl64.hpp on the instruction classes) against hipcc's six, and
// a + b = a - (p - b) shares the form: ten instructions per butterfly instead of twelve.  (b = 0: p - b = p is not canonical; a - p
// borrows for every canonical a and the correction returns a.)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ u64 ntt_sub(u64 a, u64 b) {
  u32 d0, d1, sel;
  asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %2, 0, -15, vcc"
      : "=&v"(d0), "=&v"(d1), "=v"(sel) : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32)) : "vcc");
  u64 d = ((u64)d1 << 32) | d0;
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(d) : "v"(sel), "s"(0x11111111u) : "vcc");
  return d;
}
__device__ __forceinline__ u64 ntt_add(u64 a, u64 b) { return ntt_sub(a, GL_P - b); }
// canonical form in three instructions: x + 2^32 - 1 carries exactly when x >= p, and then x + (2^32 - 1) mod 2^64 = x - p.  Used by
// ntt_shl only: after the twiddle multiplies (gl_mul) and at the loads hipcc's four-instruction form stays - with this one there the
// contiguous pass of the LDE (k_ntt_pass_pf<false, 0, 4>, at its 128-register limit) spills 17 registers.
__device__ __forceinline__ u64 ntt_canon(u64 x) {
  u64 t;
  u32 c;
  asm("v_mad_u64_u32 %0, vcc, -1, 1, %2\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %1, 0, 1, vcc"
      : "=&v"(t), "=&v"(c) : "v"(x) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, -1, %0" : "+v"(x) : "v"(c) : "vcc");
  return x;
}
// x * 2^S for canonical x, canonical result: gl_shl (gl64.hpp) with the subtraction above, and for S <= 32 the 96-bit value
// (x >> (64 - S)) : (x << S) folded with two multiply-adds (hi (2^32 - 1) + lo, its carry times 2^32 - 1 again) and made canonical
// with a third (x + 2^32 - 1 carries exactly when x >= p): 8 instructions against hipcc's 12
template <unsigned S>
__device__ __forceinline__ u64 ntt_shl(u64 x) {
  static_assert(S > 0 && S < 96 && (S <= 32 || S % 32 != 0), "shift out of range");
  if constexpr (S <= 32) {
    u64 r = x << S;
    const u32 hi = (u32)(x >> (64 - S));
    u32 c;
    asm("v_mad_u64_u32 %0, vcc, %2, -1, %0\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %1, 0, 1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, %1, -1, %0"
        : "+v"(r), "=&v"(c) : "v"(hi) : "vcc");
    return ntt_canon(r);
  } else {
    constexpr unsigned t = S % 32;
    const u32 y0 = (u32)x << t, y1 = (u32)(x >> (32 - t)), y2 = (u32)(x >> (64 - t));
    if constexpr (S < 64) {
      const u32 s = y0 + y1;
      const u32 carry = s < y0;
      return ntt_sub(((u64)s << 32) | (u32)(0u - carry), (u64)y1 + y2);
    } else {
      return ntt_sub((u64)y0 << 32, ((u64)y2 << 32) + (u64)y1 + (u64)y0);
    }
  }
}
#else
LCP2_HD u64 ntt_canon(u64 x) { return gl_canon(x); }
LCP2_HD u64 ntt_sub(u64 a, u64 b) { return gl_sub(a, b); }
LCP2_HD u64 ntt_add(u64 a, u64 b) { return gl_add(a, b); }
template <unsigned S>
LCP2_HD u64 ntt_shl(u64 x) { return gl_shl<S>(x); }
#endif

// (a - b) * w_16^k
LCP2_HD u64 ntt_dif_twiddle(u64 a, u64 b, u32 k) {
  switch (k) {
    case 1: return ntt_shl<60>(ntt_sub(b, a));
    case 2: return ntt_shl<24>(ntt_sub(b, a));
    case 3: return ntt_shl<84>(ntt_sub(a, b));
    case 4: return ntt_shl<48>(ntt_sub(a, b));
    case 5: return ntt_shl<12>(ntt_sub(a, b));
    case 6: return ntt_shl<72>(ntt_sub(b, a));
    default: return ntt_shl<36>(ntt_sub(b, a));
  }
}
// (a + x w_16^-k, a - x w_16^-k)
LCP2_HD void ntt_dit_butterfly(u64 a, u64 x, u32 k, u64 &sum, u64 &diff) {
  u64 b;
  bool neg = false;
  switch (k) {
    case 1: b = ntt_shl<36>(x); break;
    case 2: b = ntt_shl<72>(x); break;
    case 3: b = ntt_shl<12>(x); neg = true; break;
    case 4: b = ntt_shl<48>(x); neg = true; break;
    case 5: b = ntt_shl<84>(x); neg = true; break;
    case 6: b = ntt_shl<24>(x); break;
    default: b = ntt_shl<60>(x); break;
  }
  sum = neg ? ntt_sub(a, b) : ntt_add(a, b);
  diff = neg ? ntt_add(a, b) : ntt_sub(a, b);
}

// 2^RB-point transforms on registers.  dif: natural in -> bit-reversed out; dit: bit-reversed in ->
// natural out; the constant twiddles w_16^(+-k) are the shifts above.
template <u32 RB>
LCP2_HD void ntt_reg_dif(u64 *x) {
  constexpr u32 R = 1u << RB;
#pragma unroll
  for (u32 s = 0; s < RB; s++) {
    const u32 half = R >> (s + 1);
#pragma unroll
    for (u32 q = 0; q < R / 2; q++) {
      const u32 i = q & (half - 1), i0 = ((q - i) << 1) | i, i1 = i0 + half;
      const u64 a = x[i0], b = x[i1];
      x[i0] = ntt_add(a, b);
      x[i1] = i ? ntt_dif_twiddle(a, b, i * (8 / half)) : ntt_sub(a, b);
    }
  }
}
template <u32 RB>
LCP2_HD void ntt_reg_dit(u64 *x) {
  constexpr u32 R = 1u << RB;
#pragma unroll
  for (u32 s = RB; s-- > 0;) {
    const u32 half = R >> (s + 1);
#pragma unroll
    for (u32 q = 0; q < R / 2; q++) {
      const u32 i = q & (half - 1), i0 = ((q - i) << 1) | i, i1 = i0 + half;
      const u64 a = x[i0];
      if (i) {
        ntt_dit_butterfly(a, x[i1], i * (8 / half), x[i0], x[i1]);
      } else {
        const u64 b = x[i1];
        x[i0] = ntt_add(a, b);
        x[i1] = ntt_sub(a, b);
      }
    }
  }
}

LCP2_HD u64 two_level(const TwoLevelTable &t, u64 e) {
  if (t.h == NTT_DIRECT) return t.lo[e];
  return gl_mul(t.lo[e & ((1ull << t.h) - 1)], t.hi[e >> t.h]);
}

struct NttPass {
  NttPassParams p;

  LCP2_HD u64 global_index(u32 wg, u32 local) const {
    const u32 S = p.S, B = p.B, g = p.g_lo;
    u64 ls = local & ((1u << S) - 1);
    u64 t = (local >> S) & ((1u << B) - 1);
    u64 bt = local >> (S + B);
    u32 nb = p.L - S - B;
    u64 seg = wg & ((1u << (g - S)) - 1);
    u64 hi = wg >> (g - S);
    return ls | (seg << S) | (t << g) | (((hi << nb) | bt) << (g + B));
  }
  LCP2_HD u64 low_bits(u32 wg, u32 local) const {  // l = value of the g_lo bits below the group
    u64 ls = local & ((1u << p.S) - 1);
    u64 seg = wg & ((1u << (p.g_lo - p.S)) - 1);
    return ls | (seg << p.S);
  }
  LCP2_HD u64 scale_at(u64 j, u32 z) const {
    if (p.scale_mode == 1) return p.scale_scalar;
    TwoLevelTable t = p.sc;
    t.lo += (u64)z * p.sc_lo_z_stride;
    t.hi += (u64)z * p.sc_hi_z_stride;
    return two_level(t, j);
  }
  LCP2_HD u64 group_twiddle(u32 wg, u32 local) const {
    u64 l = low_bits(wg, local);
    u32 tp = (local >> p.S) & ((1u << p.B) - 1);
    if (p.tw.h == NTT_DIRECT) return p.tw.lo[((u64)tp << p.g_lo) | l];  // [t'][l] = w^(l * bitrev(t')): consecutive lanes, consecutive words
    u64 k1 = bitrev32(tp, p.B);
    return two_level(p.tw, l * k1);
  }

  // The load and store phases move NTT_BATCH elements per thread at a time with the loads issued back to back before any of them
  // is used: with runtime trip counts the compiler otherwise serialises one HBM / LDS round trip per element.
  //
  // Thread tid owns the local indices i_k = tid + k * nthr.  When nthr is a multiple of 2^S (and of 16) the run offset of i_k
  // and its LDS padding phase do not depend on k, so the global index, the index into a one-level factor table (which is
  // indexed like the data) and the LDS word all advance by constants: one address computation per thread instead of a dozen
  // shifts and masks per element.  Slabs that the threads do not tile evenly (small transforms) take the element-wise path.
  LCP2_HD bool strided_walk(u32 nthr) const {
    return p.S + p.B == p.L && (nthr & ((1u << p.S) - 1)) == 0 && (nthr & 15) == 0 && ((1u << p.L) % (nthr * NTT_BATCH)) == 0;
  }
  template <bool INV>
  LCP2_HD void load(u64 *lds, u32 tid, u32 nthr, u32 wg, u32 col, u32 z) const {
    const u64 *src = p.in + (u64)col * p.in_col_stride + (u64)z * p.in_z_stride;
    const u32 n = 1u << p.L;
    // one-level factor tables are plain loads: issue them with the data (a two-level factor costs a multiply and waits)
    const bool factor = INV ? p.g_lo != 0 : p.scale_mode != 0;
    const bool early = factor && (INV ? p.tw.h == NTT_DIRECT : (p.scale_mode == 2 && p.sc.h == NTT_DIRECT));
    if (strided_walk(nthr)) {
      const u64 g0 = global_index(wg, tid), gs = (u64)(nthr >> p.S) << p.g_lo;
      const u32 ph0 = lds_phys(tid), ps = nthr + (nthr >> 4);
      const u64 *fac = nullptr;
      if (early) fac = INV ? p.tw.lo + (((u64)((tid >> p.S) & ((1u << p.B) - 1)) << p.g_lo) | low_bits(wg, tid)) : p.sc.lo + (u64)z * p.sc_lo_z_stride + g0;
      for (u32 k0 = 0; k0 < n / nthr; k0 += NTT_BATCH) {
        u64 v[NTT_BATCH], f[NTT_BATCH];
#pragma unroll
        for (u32 j = 0; j < NTT_BATCH; j++) {
          v[j] = src[g0 + (k0 + j) * gs];
          if (early) f[j] = fac[(k0 + j) * gs];
        }
#pragma unroll
        for (u32 j = 0; j < NTT_BATCH; j++) {
          const u32 i = tid + (k0 + j) * nthr;
          u64 x;  // the multiply takes any u64; without one the element is made canonical here
          if (factor) x = gl_mul(v[j], early ? f[j] : (INV ? group_twiddle(wg, i) : scale_at(g0 + (k0 + j) * gs, z)));
          else x = gl_canon(v[j]);
          lds[ph0 + (k0 + j) * ps] = x;
        }
      }
      return;
    }
    for (u32 i0 = tid; i0 < n; i0 += nthr * NTT_BATCH) {
      u64 g[NTT_BATCH], v[NTT_BATCH], f[NTT_BATCH];
#pragma unroll
      for (u32 j = 0; j < NTT_BATCH; j++) {
        u32 i = i0 + j * nthr;
        if (i >= n) i = 0;  // out-of-range lanes re-read element 0 (always valid)
        g[j] = global_index(wg, i);
        v[j] = src[g[j]];
        if (early) f[j] = INV ? group_twiddle(wg, i) : scale_at(g[j], z);
      }
#pragma unroll
      for (u32 j = 0; j < NTT_BATCH; j++) {
        u32 i = i0 + j * nthr;
        if (i >= n) continue;
        u64 x = gl_canon(v[j]);
        if (factor) x = gl_mul(x, early ? f[j] : (INV ? group_twiddle(wg, i) : scale_at(g[j], z)));
        lds[lds_phys(i)] = x;
      }
    }
  }

  // ---- The phases of the prefetching kernel (kernels_ntt.hip k_ntt_pass_pf): forward passes over 2^13-element slabs with 512
  // threads, STRIDED = (S 4, B 9: steps of 3 bits at local bits 10, 7, 4 - the first pass at 2^22; or SS = 7: S 7, B 6: steps at
  // local bits 10 and 7 - the first pass at 2^19, the size of the light-client step) or contiguous (S 0, B 13: 3 + 3 + 3 + 4 bits).
  // A thread's 16 prefetched elements (local indices tid + 512 k) are exactly two groups of the FIRST register step (k even /
  // k odd: local bits 10..12 = k >> 1), so that step runs on the prefetched registers and the slab reaches LDS already
  // transformed: one LDS round trip and one barrier less than load -> LDS -> step.  Likewise the LAST step of a strided pass
  // (bits 4..6: the 8 elements of a group are 8 runs apart in memory, the 64 lanes of a wave cover four 128-byte runs of each)
  // stores straight from registers, and the last step of a contiguous pass (16 consecutive elements per thread) is followed by
  // a transposition through the wave's OWN 1024-element region of the slab, which needs no workgroup barrier.
  LCP2_HD void prefetch(u32 tid, u32 nthr, u32 wg, u32 col, u32 z, u64 *v) const {
    const u64 *src = p.in + (u64)col * p.in_col_stride + (u64)z * p.in_z_stride;
    const u64 g0 = global_index(wg, tid), gs = (u64)(nthr >> p.S) << p.g_lo;
#pragma unroll
    for (u32 j = 0; j < 16; j++) v[j] = src[g0 + j * gs];
  }
  // FMODE 0: no factor (elements made canonical); 1: coset scale from the one-level table (one load and one multiply per
  // element); 2: coset scale computed: shift^(g0 + k gs) = shift^g0 (one table load per thread) x (shift^gs)^k (wave-uniform
  // scalars), with the thread's common factor carried through the linear transform into the step twiddles
  // The step twiddles of the lower steps are a strided subset of p.group_tw (w_{2^B}^(e << shift)): read from there they cost an
  // L2 round trip per step (a 64 KiB table walked with a stride of 8 or 64 entries does not live in the 32 KiB vector cache).
  // The workgroup keeps compact copies in the LDS left beside the slab (12 KiB per workgroup with two per CU):
  //   strided    [0, 512)   first step  w_512^e          [512, 576)    step at bit 7  w_64^e
  //   contiguous [0, 1024)  step at bit 7  w_1024^e      [1024, 1152)  step at bit 4  w_128^e   (the first step spans the whole table)
  static constexpr u32 PF_TW_WORDS_STRIDED = 576, PF_TW_WORDS_CONTIGUOUS = 1152;
  template <bool STRIDED, u32 SS = 4>
  LCP2_HD void pf_stage_twiddles(u64 *twl, u32 tid) const {
    if (STRIDED && SS == 7) {
      if (tid < 64) twl[tid] = p.group_tw[tid];  // w_64^e: the first step; the last one has no step twiddle
    } else if (STRIDED) {
      twl[tid] = p.group_tw[tid];
      if (tid < 64) twl[512 + tid] = p.group_tw[tid << 3];
    } else {
      twl[tid] = p.group_tw[tid << 3];
      twl[512 + tid] = p.group_tw[(512 + tid) << 3];
      if (tid < 128) twl[1024 + tid] = p.group_tw[tid << 6];
    }
  }
  // twt: the step twiddles w^(m * bitrev(j)) of the step, indexed by the exponent (p.group_tw itself for the top step, or the
  // workgroup's compact copy in LDS, pf_stage_twiddles)
  template <bool STRIDED, int FMODE, u32 SS = 4>
  LCP2_HD void pf_first_step(u64 *lds, u32 tid, u32 wg, u32 z, const u64 *v, const u64 *twt) const {
    constexpr u32 PS = 512 + 32;  // lds_phys(tid + 512 k) = lds_phys(tid) + k * PS
    const u32 ph0 = lds_phys(tid);
    const u64 gs = (u64)(512u >> p.S) << p.g_lo;
    const u64 *fac = FMODE ? p.sc.lo + (u64)z * p.sc_lo_z_stride + global_index(wg, tid) : nullptr;
    const u64 *stp = FMODE == 2 ? p.sc_step + 16 * (u64)z : nullptr;
    u64 f0 = 1;
    if (FMODE == 2) f0 = fac[0];
#pragma unroll
    for (u32 b = 0; b < 2; b++) {
      const u32 m = STRIDED ? (SS == 7 ? ((tid >> 7) | (b << 2)) : ((tid >> 4) | (b << 5))) : (tid | (b << 9));  // group bits below the step
      u64 x[8], tw[8], f[8];
#pragma unroll
      for (u32 j = 1; j < 8; j++) tw[j] = twt[m * bitrev32(j, 3)];
      if (FMODE == 1) {
#pragma unroll
        for (u32 j = 0; j < 8; j++) f[j] = fac[(2 * j + b) * gs];
#pragma unroll
        for (u32 j = 0; j < 8; j++) x[j] = gl_mul(v[2 * j + b], f[j]);
      } else if (FMODE == 2) {
        x[0] = gl_canon(v[b]);
#pragma unroll
        for (u32 j = 1; j < 8; j++) x[j] = gl_mul(v[2 * j + b], stp[2 * j]);
      } else {
#pragma unroll
        for (u32 j = 0; j < 8; j++) x[j] = STRIDED ? gl_canon(v[2 * j + b]) : v[2 * j + b];  // contiguous: p.canonical_in (ntt_pf_contiguous)
      }
      ntt_reg_dif<3>(x);
      if (FMODE == 2) {
        const u64 fb = b ? gl_mul(f0, stp[1]) : gl_canon(f0);
        x[0] = gl_mul(x[0], fb);
#pragma unroll
        for (u32 j = 1; j < 8; j++) x[j] = gl_mul(x[j], gl_mul_nc(tw[j], fb));
      } else {
#pragma unroll
        for (u32 j = 1; j < 8; j++) x[j] = gl_mul(x[j], tw[j]);
      }
#pragma unroll
      for (u32 j = 0; j < 8; j++) lds[ph0 + (2 * j + b) * PS] = x[j];
    }
  }
  // a middle step: 3 bits at local bit P (7 or 4): the two groups of a thread, the LDS reads and twiddle loads of the second one
  // issued before the transform of the first
  template <u32 P>
  LCP2_HD void pf_mid_step(u64 *lds, u32 tid, u32 kb_top, const u64 *twt) const {
    constexpr u32 PSTRIDE = (1u << P) + (1u << (P - 4));
    const u32 mbits = kb_top - 2;
    u64 x0[8], x1[8], tw0[8], tw1[8];
    const u32 b0 = ((tid >> P) << (P + 3)) | (tid & ((1u << P) - 1)), g1 = tid + 512, b1 = ((g1 >> P) << (P + 3)) | (g1 & ((1u << P) - 1));
    const u32 p0 = lds_phys(b0), p1 = lds_phys(b1);
    const u32 m0 = (b0 >> p.S) & ((1u << mbits) - 1), m1 = (b1 >> p.S) & ((1u << mbits) - 1);
#pragma unroll
    for (u32 j = 0; j < 8; j++) x0[j] = lds[p0 + j * PSTRIDE];
    if (mbits) {
#pragma unroll
      for (u32 j = 1; j < 8; j++) tw0[j] = twt[m0 * bitrev32(j, 3)];
    }
#pragma unroll
    for (u32 j = 0; j < 8; j++) x1[j] = lds[p1 + j * PSTRIDE];
    if (mbits) {
#pragma unroll
      for (u32 j = 1; j < 8; j++) tw1[j] = twt[m1 * bitrev32(j, 3)];
    }
    ntt_reg_dif<3>(x0);
    if (mbits) {
#pragma unroll
      for (u32 j = 1; j < 8; j++) x0[j] = gl_mul(x0[j], tw0[j]);
    }
#pragma unroll
    for (u32 j = 0; j < 8; j++) lds[p0 + j * PSTRIDE] = x0[j];
    ntt_reg_dif<3>(x1);
    if (mbits) {
#pragma unroll
      for (u32 j = 1; j < 8; j++) x1[j] = gl_mul(x1[j], tw1[j]);
    }
#pragma unroll
    for (u32 j = 0; j < 8; j++) lds[p1 + j * PSTRIDE] = x1[j];
  }
  // last step of a strided pass (local bits P..P+2 with P = S, no step twiddle) in two halves: the LDS reads, then - after the
  // barrier that lets the next slab in - transform, inter-group twiddle and the global stores from registers
  template <u32 P = 4>
  LCP2_HD void pf_last_strided_read(const u64 *lds, u32 tid, u64 *x) const {
    constexpr u32 PSTRIDE = (1u << P) + (1u << (P - 4));
#pragma unroll
    for (u32 q = 0; q < 2; q++) {
      const u32 g = tid + 512 * q, base = ((g >> P) << (P + 3)) | (g & ((1u << P) - 1)), pb = lds_phys(base);
#pragma unroll
      for (u32 j = 0; j < 8; j++) x[8 * q + j] = lds[pb + j * PSTRIDE];
    }
  }
  template <u32 P = 4>
  LCP2_HD void pf_last_strided_store(u32 tid, u32 wg, u32 col, u32 z, u64 *x) const {
    u64 *dst = p.out + (u64)col * p.out_col_stride + (u64)(bitrev32(z, p.zbits) - p.out_block_base) * p.out_z_stride;
    const bool direct = p.tw.h == NTT_DIRECT;
#pragma unroll
    for (u32 q = 0; q < 2; q++) {
      const u32 g = tid + 512 * q, base = ((g >> P) << (P + 3)) | (g & ((1u << P) - 1));
      const u64 g0 = global_index(wg, base), gs = (u64)1 << p.g_lo;  // element j of the group: local base | j << P, one run further
      u64 tw[8];
      if (direct) {
        const u64 *fac = p.tw.lo + (((u64)((base >> P) & ((1u << (13 - P)) - 1)) << p.g_lo) | low_bits(wg, base));
#pragma unroll
        for (u32 j = 0; j < 8; j++) tw[j] = fac[j * gs];
      } else {
#pragma unroll
        for (u32 j = 0; j < 8; j++) tw[j] = group_twiddle(wg, base | (j << P));
      }
      ntt_reg_dif<3>(x + 8 * q);
#pragma unroll
      for (u32 j = 0; j < 8; j++) dst[g0 + j * gs] = gl_mul(x[8 * q + j], tw[j]);
    }
  }
  // last step of a contiguous pass (bits 0..3: 16 consecutive elements per thread, no twiddle), written back into the wave's own
  // region, and the store of that region in runs of 512 bytes per wave instruction (reads only what the same wave wrote)
  LCP2_HD void pf_last_contiguous(u64 *lds, u32 tid) const {
    const u32 pb = lds_phys(tid << 4);
    u64 x[16];
#pragma unroll
    for (u32 j = 0; j < 16; j++) x[j] = lds[pb + j];
    ntt_reg_dif<4>(x);
#pragma unroll
    for (u32 j = 0; j < 16; j++) lds[pb + j] = x[j];
  }
  LCP2_HD void pf_store_wave_rows(const u64 *lds, u32 tid, u32 wg, u32 col, u32 z) const {
    u64 *dst = p.out + (u64)col * p.out_col_stride + (u64)(bitrev32(z, p.zbits) - p.out_block_base) * p.out_z_stride + ((u64)wg << 13);
    const u32 i0 = ((tid >> 6) << 10) | (tid & 63);  // element 64 k + lane of the wave's 1024
    u64 v[16];
#pragma unroll
    for (u32 k = 0; k < 16; k++) v[k] = lds[lds_phys(i0 + 64 * k)];
#pragma unroll
    for (u32 k = 0; k < 16; k++) dst[i0 + 64 * k] = v[k];
  }

  // Register step over the RB group bits whose top one is kb_top (group-relative): see the header comment.
  // PC: the local bit position P of the step's lowest bit when it is known at compile time (-1: runtime).  For P >= 4 the padded
  // LDS word of element j is word(base) + j * (2^P + 2^(P-4)), for the bottom step (P = 0, 16 elements) word(base) + j: with
  // PC given these are immediate offsets of the ds instructions instead of three address instructions per access.
  template <bool INV, u32 RB, int PC>
  LCP2_HD void step_r(u64 *lds, u32 tid, u32 nthr, u32 kb_top) const {
    constexpr u32 R = 1u << RB;
    const u32 mbits = kb_top + 1 - RB;                      // group bits below the step
    const u32 P = PC >= 0 ? (u32)PC : p.S + mbits;          // local bit position of the step's lowest bit
    const u32 ngroups = 1u << (p.L - RB);
    const u32 tw_shift = p.B - kb_top - 1;                  // w_{N'} = w_{2^B}^(2^tw_shift)
    const bool linear = P >= 4 || (P == 0 && RB == 4);
    const u32 pstride = P >= 4 ? (1u << P) + (1u << (P - 4)) : 1u;
    for (u32 g = tid; g < ngroups; g += nthr) {
      const u32 base = ((g >> P) << (P + RB)) | (g & ((1u << P) - 1));
      const u32 m = (base >> p.S) & ((1u << mbits) - 1);
      const u32 pb = lds_phys(base);
      u64 x[R], tw[R];
#pragma unroll
      for (u32 j = 0; j < R; j++) x[j] = lds[linear ? pb + j * pstride : lds_phys(base | (j << P))];
      if (mbits) {
#pragma unroll
        for (u32 j = 1; j < R; j++) tw[j] = p.group_tw[(u64)(m * bitrev32(j, RB)) << tw_shift];
      }
      if (!INV) {
        ntt_reg_dif<RB>(x);
        if (mbits) {
#pragma unroll
          for (u32 j = 1; j < R; j++) x[j] = gl_mul(x[j], tw[j]);
        }
      } else {
        if (mbits) {
#pragma unroll
          for (u32 j = 1; j < R; j++) x[j] = gl_mul(x[j], tw[j]);
        }
        ntt_reg_dit<RB>(x);
      }
#pragma unroll
      for (u32 j = 0; j < R; j++) lds[linear ? pb + j * pstride : lds_phys(base | (j << P))] = x[j];
    }
  }
  // step si of the pass in execution order (forward: from the top bits down; inverse: from the bottom up)
  template <bool INV>
  LCP2_HD void step(u64 *lds, u32 tid, u32 nthr, u32 si) const {
    const u32 idx = INV ? p.nsteps - 1 - si : si;
    u32 above = 0;
    for (u32 i = 0; i < idx; i++) above += (p.step_plan >> (4 * i)) & 15;
    const u32 kb_top = p.B - 1 - above;
    const u32 rb = (p.step_plan >> (4 * idx)) & 15, P = p.S + kb_top + 1 - rb;
    // the steps of the 2^13-element slabs of a large transform (3 + 3 + 3 + 4 contiguous, 3 + 3 + 3 strided over 16-element runs)
    if (rb == 3 && P == 4) return step_r<INV, 3, 4>(lds, tid, nthr, kb_top);
    if (rb == 3 && P == 7) return step_r<INV, 3, 7>(lds, tid, nthr, kb_top);
    if (rb == 3 && P == 10) return step_r<INV, 3, 10>(lds, tid, nthr, kb_top);
    if (rb == 4 && P == 0) return step_r<INV, 4, 0>(lds, tid, nthr, kb_top);
    switch (rb) {
      case 1: step_r<INV, 1, -1>(lds, tid, nthr, kb_top); break;
      case 2: step_r<INV, 2, -1>(lds, tid, nthr, kb_top); break;
      case 3: step_r<INV, 3, -1>(lds, tid, nthr, kb_top); break;
      default: step_r<INV, 4, -1>(lds, tid, nthr, kb_top); break;
    }
  }

  template <bool INV>
  LCP2_HD void store(const u64 *lds, u32 tid, u32 nthr, u32 wg, u32 col, u32 z) const {
    u64 *dst = p.out + (u64)col * p.out_col_stride + (u64)(bitrev32(z, p.zbits) - p.out_block_base) * p.out_z_stride;
    const u32 n = 1u << p.L;
    const bool scaled = INV ? p.scale_mode != 0 : p.g_lo != 0;
    if (strided_walk(nthr)) {
      const u64 g0 = global_index(wg, tid), gs = (u64)(nthr >> p.S) << p.g_lo;
      const u32 ph0 = lds_phys(tid), ps = nthr + (nthr >> 4);
      const bool direct = scaled && (INV ? (p.scale_mode == 2 && p.sc.h == NTT_DIRECT) : p.tw.h == NTT_DIRECT);
      const u64 *fac = nullptr;
      if (direct) fac = INV ? p.sc.lo + (u64)z * p.sc_lo_z_stride + g0 : p.tw.lo + (((u64)((tid >> p.S) & ((1u << p.B) - 1)) << p.g_lo) | low_bits(wg, tid));
      for (u32 k0 = 0; k0 < n / nthr; k0 += NTT_BATCH) {
        u64 v[NTT_BATCH], tw[NTT_BATCH];
#pragma unroll
        for (u32 j = 0; j < NTT_BATCH; j++) {
          v[j] = lds[ph0 + (k0 + j) * ps];
          tw[j] = 1;
          if (direct) tw[j] = fac[(k0 + j) * gs];
          else if (scaled) tw[j] = INV ? scale_at(g0 + (k0 + j) * gs, z) : group_twiddle(wg, tid + (k0 + j) * nthr);
        }
#pragma unroll
        for (u32 j = 0; j < NTT_BATCH; j++) dst[g0 + (k0 + j) * gs] = scaled ? gl_mul(v[j], tw[j]) : v[j];
      }
      return;
    }
    for (u32 i0 = tid; i0 < n; i0 += nthr * NTT_BATCH) {
      u64 v[NTT_BATCH], tw[NTT_BATCH];
#pragma unroll
      for (u32 j = 0; j < NTT_BATCH; j++) {
        u32 i = i0 + j * nthr;
        if (i >= n) i = 0;
        v[j] = lds[lds_phys(i)];
        tw[j] = 1;
        if (!INV) {
          if (p.g_lo) tw[j] = group_twiddle(wg, i);
        } else {
          if (p.scale_mode) tw[j] = scale_at(global_index(wg, i), z);
        }
      }
#pragma unroll
      for (u32 j = 0; j < NTT_BATCH; j++) {
        u32 i = i0 + j * nthr;
        if (i >= n) continue;
        dst[global_index(wg, i)] = scaled ? gl_mul(v[j], tw[j]) : v[j];
      }
    }
  }
};

// ---- which launches take the prefetching kernel (shared by the library and the emulation harness) ----
// The slab shapes of a large forward transform, factors (if any) from a one-level table, and enough slabs that the shorter
// grid still fills the chip.  (Forward passes only: the inverse instantiations need more VGPRs than four waves per SIMD leave.)
// the strided shapes: 4 (S 4, B 9), 7 (S 7, B 6), 0 = neither
inline u32 ntt_pf_strided_s(const NttPassParams &p) {
  if (p.L == 13 && p.S == 4 && p.B == 9 && p.nsteps == 3 && p.step_plan == 0x333) return 4;
  if (p.L == 13 && p.S == 7 && p.B == 6 && p.nsteps == 2 && p.step_plan == 0x33) return 7;
  return 0;
}
inline bool ntt_pf_strided(const NttPassParams &p) { return ntt_pf_strided_s(p) != 0; }
inline bool ntt_pf_contiguous(const NttPassParams &p) {
  return p.L == 13 && p.S == 0 && p.B == 13 && p.nsteps == 4 && p.step_plan == 0x4333 && p.g_lo == 0 && p.scale_mode == 0 && p.canonical_in;
}
// slabs per workgroup of the prefetching form, 0 = the launch takes the plain kernel.  Only the first load of a workgroup is
// exposed, so as many as still leave 2048 workgroups (4 rounds of the 512 resident ones) and divide the slab count into a
// multiple of 8 workgroups (the XCD mapping).
inline u32 ntt_pf_slabs_per_wg(const NttPassParams &p, bool inverse, u64 total) {
  if (inverse || total >= (1ull << 31)) return 0;
  const bool strided = ntt_pf_strided(p) && p.g_lo != 0 && (p.scale_mode == 0 || (p.scale_mode == 2 && p.sc.h == NTT_DIRECT));
  if (!strided && !ntt_pf_contiguous(p)) return 0;
  for (u32 k = 16; k >= 2; k >>= 1)
    if (total % (k * 8) == 0 && total / k >= 2048) return k;
  return 0;
}

// ---- host-side pass planning (shared by the library and the emulation harness) ----
struct NttGroup {
  u32 B, g_lo, L, S;
};

// Cuts lg bits into groups, listed from the TOP group down to the contiguous one.
inline int ntt_plan(u32 lg, NttGroup out[8]) {
  int n = 0;
  if (lg <= NTT_MAX_L) {
    out[n++] = NttGroup{lg, 0, lg, 0};
    return n;
  }
  u32 upper = lg - NTT_MAX_L;
  u32 npass = (upper + NTT_MAX_STRIDED_B - 1) / NTT_MAX_STRIDED_B;
  u32 pos = lg;
  for (u32 i = 0; i < npass; i++) {
    u32 B = upper / npass + (i < upper % npass ? 1 : 0);
    pos -= B;
    u32 S = NTT_MAX_L - B;
    if (S > pos) S = pos;
    out[n++] = NttGroup{B, pos, B + S, S};
  }
  out[n++] = NttGroup{NTT_MAX_L, 0, NTT_MAX_L, 0};
  return n;
}

// Cuts the B stages of a group into register steps (top-down order).  The bottom step of a contiguous group
// (S = 0) takes 4 bits (the padded layout makes it conflict free), everything above it is cut into 3s with 4s
// absorbing the remainder, so that no step starts at local bit 1..3.
inline u32 ntt_step_plan(u32 B, u32 S, u32 &plan) {
  unsigned char up[NTT_MAX_STEPS];  // bottom-up
  u32 n = 0, left = B;
  if (S == 0 && left >= 4) { up[n++] = 4; left -= 4; }
  while (left) {
    u32 r;
    if (left <= 4) r = left;
    else if (left % 3 == 0) r = 3;
    else if (left == 5) r = 3;
    else r = 4;
    up[n++] = (unsigned char)r;
    left -= r;
  }
  plan = 0;
  for (u32 i = 0; i < n; i++) plan |= (u32)up[n - 1 - i] << (4 * i);
  return n;
}

// Bit-reversal permutation of 2^lg elements through a 64 x 64 LDS tile
// (rows padded to 65): both the read and the write are 512-byte runs.
struct BitrevTile {
  const u64 *in;
  u64 *out;
  u64 in_col_stride, out_col_stride;
  u32 lg;
  unsigned long long *noncanonical;  // nullable: set to 1 when an input value is >= p (the caller's buffer is then not usable as is
                                     // where canonical values are assumed: the witness check of lcp2_prove)
  LCP2_HD void load(u64 *lds, u32 tid, u32 nthr, u32 wg, u32 col) const {
    const u64 *src = in + (u64)col * in_col_stride;
    bool seen = false;
    for (u32 i = tid; i < 4096; i += nthr) {
      u32 h = i >> 6, l = i & 63;
      u64 g = ((u64)h << (lg - 6)) | ((u64)wg << 6) | l;
      const u64 v = src[g];
      seen = seen || v >= GL_P;
      lds[h * 65 + l] = gl_canon(v);
    }
    if (seen && noncanonical) *noncanonical = 1;  // every writer stores the same value
  }
  LCP2_HD void store(const u64 *lds, u32 tid, u32 nthr, u32 wg, u32 col) const {
    u64 *dst = out + (u64)col * out_col_stride;
    u32 mid = bitrev32(wg, lg - 12);
    for (u32 i = tid; i < 4096; i += nthr) {
      u32 a = i >> 6, b = i & 63;  // output row a (= rev6(l)), output column b (= rev6(h))
      u32 l = bitrev32(a, 6), h = bitrev32(b, 6);
      u64 g = ((u64)a << (lg - 6)) | ((u64)mid << 6) | b;
      dst[g] = lds[h * 65 + l];
    }
  }
};

}  // namespace lcp2
