// Multi-pass radix-2 NTT over Goldilocks, LDS-staged, written for gfx950.
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
// The pass body is split into load / stage / store phases that take an explicit
// thread id: the HIP kernel calls them with a barrier in between, the CPU
// emulation harness (tests/emu) calls them in loops.
#pragma once
#include "gl64.hpp"

namespace lcp2 {

constexpr u32 NTT_MAX_L = 13;       // slab = 2^13 elements = 64 KiB of LDS
constexpr u32 NTT_THREADS = 1024;   // 2 workgroups x 16 waves per CU share the 160 KiB LDS: the stages are barrier/latency bound, so
                                    // occupancy matters more than work per thread (measured LDE: 256 thr 174 ms, 512 thr 132 ms, 1024 thr 123 ms)
constexpr u32 NTT_MAX_STRIDED_B = 9;
constexpr u32 NTT_SEG_BITS = 4;     // 16 x 8 B = 128-byte runs in strided passes
constexpr u32 NTT_BATCH = 4;        // elements (butterflies) in flight per thread

struct TwoLevelTable {  // value(e) = lo[e & (2^h - 1)] * hi[e >> h]
  const u64 *lo;
  const u64 *hi;
  u32 h;
};

struct NttPassParams {
  const u64 *in;
  u64 *out;
  u64 in_col_stride, out_col_stride;  // elements, per blockIdx.y
  u64 in_z_stride, out_z_stride;      // elements, per blockIdx.z (coset); out uses bitrev(z, zbits)
  u32 zbits;
  u32 z_base;                         // coset index of blockIdx.z = 0 (coset-sharded LDE: one launch per leaf block)
  u32 out_block_base;                 // leaf block that sits at offset 0 of `out` (a rank holds blocks [base, base + count))
  u32 L, S, B, g_lo;                  // slab bits, run bits, group bits, bits below the group
  const u64 *stage_tw;                // w_{2^B}^j (or its inverse), j < 2^(B-1)
  TwoLevelTable tw;                   // w_{N_g}^e (or inverse); used when g_lo > 0
  u32 scale_mode;                     // 0 none, 1 scalar, 2 two-level table (per z)
  u64 scale_scalar;
  TwoLevelTable sc;                   // forward: applied at load; inverse: at store
  u64 sc_lo_z_stride, sc_hi_z_stride;
};

LCP2_HD u64 two_level(const TwoLevelTable &t, u64 e) {
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
    u64 k1 = bitrev32(tp, p.B);
    return two_level(p.tw, l * k1);
  }

  // All three phases move NTT_BATCH elements per thread at a time with the loads issued back to back before any
  // of them is used: with runtime trip counts the compiler otherwise serialises one HBM / LDS round trip per element.
  template <bool INV>
  LCP2_HD void load(u64 *lds, u32 tid, u32 nthr, u32 wg, u32 col, u32 z) const {
    const u64 *src = p.in + (u64)col * p.in_col_stride + (u64)z * p.in_z_stride;
    const u32 n = 1u << p.L;
    for (u32 i0 = tid; i0 < n; i0 += nthr * NTT_BATCH) {
      u64 g[NTT_BATCH], v[NTT_BATCH];
#pragma unroll
      for (u32 j = 0; j < NTT_BATCH; j++) {
        u32 i = i0 + j * nthr;
        g[j] = global_index(wg, i < n ? i : 0);  // out-of-range lanes re-read element 0 (always valid)
        v[j] = src[g[j]];
      }
#pragma unroll
      for (u32 j = 0; j < NTT_BATCH; j++) {
        u32 i = i0 + j * nthr;
        if (i >= n) continue;
        u64 x = gl_canon(v[j]);
        if (!INV) {
          if (p.scale_mode) x = gl_mul(x, scale_at(g[j], z));
        } else {
          if (p.g_lo) x = gl_mul(x, group_twiddle(wg, i));
        }
        lds[i] = x;
      }
    }
  }

  // one radix-2 stage on local bit b (S <= b < S + B)
  template <bool INV>
  LCP2_HD void stage(u64 *lds, u32 tid, u32 nthr, u32 b) const {
    const u32 half = 1u << (p.L - 1);
    const u32 kb = b - p.S;  // bit position inside the group
    for (u32 q0 = tid; q0 < half; q0 += nthr * NTT_BATCH) {
      u32 i0[NTT_BATCH];
      u64 a[NTT_BATCH], c[NTT_BATCH], w[NTT_BATCH];
#pragma unroll
      for (u32 j = 0; j < NTT_BATCH; j++) {
        u32 q = q0 + j * nthr;
        if (q >= half) q = 0;
        i0[j] = ((q >> b) << (b + 1)) | (q & ((1u << b) - 1));
        u32 k = (i0[j] >> p.S) & ((1u << kb) - 1);
        w[j] = p.stage_tw[(u64)k << (p.B - 1 - kb)];
        a[j] = lds[i0[j]];
        c[j] = lds[i0[j] | (1u << b)];
      }
#pragma unroll
      for (u32 j = 0; j < NTT_BATCH; j++) {
        if (q0 + j * nthr >= half) continue;
        const u32 i1 = i0[j] | (1u << b);
        if (!INV) {
          lds[i0[j]] = gl_add(a[j], c[j]);
          lds[i1] = gl_mul(gl_sub(a[j], c[j]), w[j]);
        } else {
          u64 t = gl_mul(c[j], w[j]);
          lds[i0[j]] = gl_add(a[j], t);
          lds[i1] = gl_sub(a[j], t);
        }
      }
    }
  }

  template <bool INV>
  LCP2_HD void store(const u64 *lds, u32 tid, u32 nthr, u32 wg, u32 col, u32 z) const {
    u64 *dst = p.out + (u64)col * p.out_col_stride + (u64)(bitrev32(z, p.zbits) - p.out_block_base) * p.out_z_stride;
    const u32 n = 1u << p.L;
    for (u32 i0 = tid; i0 < n; i0 += nthr * NTT_BATCH) {
      u64 v[NTT_BATCH], tw[NTT_BATCH];
#pragma unroll
      for (u32 j = 0; j < NTT_BATCH; j++) {
        u32 i = i0 + j * nthr;
        if (i >= n) i = 0;
        v[j] = lds[i];
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
        const bool scaled = INV ? p.scale_mode != 0 : p.g_lo != 0;
        dst[global_index(wg, i)] = scaled ? gl_mul(v[j], tw[j]) : v[j];
      }
    }
  }
};

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

// Bit-reversal permutation of 2^lg elements through a 64 x 64 LDS tile
// (rows padded to 65): both the read and the write are 512-byte runs.
struct BitrevTile {
  const u64 *in;
  u64 *out;
  u64 in_col_stride, out_col_stride;
  u32 lg;
  LCP2_HD void load(u64 *lds, u32 tid, u32 nthr, u32 wg, u32 col) const {
    const u64 *src = in + (u64)col * in_col_stride;
    for (u32 i = tid; i < 4096; i += nthr) {
      u32 h = i >> 6, l = i & 63;
      u64 g = ((u64)h << (lg - 6)) | ((u64)wg << 6) | l;
      lds[h * 65 + l] = gl_canon(src[g]);
    }
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
