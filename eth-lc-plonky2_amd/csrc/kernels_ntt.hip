// NTT pass kernels for gfx950 (rows a2, a3, a4 of SURVEY.md section 8): see ntt.hpp for the
// decomposition.  One workgroup = one 2^L-element slab in LDS (up to 68 KiB with padding, so two
// workgroups share a CU's 160 KiB), NTT_THREADS threads, one barrier per register step (3-4 stages).
// Global traffic is one read and one write of the slab per pass, in runs of >= 128 bytes.
#include <cstdlib>
#include "internal.hpp"

namespace lcp2 {

template <bool INV>
__global__ __launch_bounds__(NTT_THREADS, 4) void k_ntt_pass(NttPassParams p) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  NttPass pass{p};
  const u32 tid = threadIdx.x;
  u32 wg = blockIdx.x, col = blockIdx.y, z = blockIdx.z + p.z_base;
  if (p.xcd_group) {
    // XCD-aware mapping (speed only): blocks b and b + 8 share an XCD and its L2.  The 2^zbits coset transforms of an LDE read
    // the same coefficient slab, so the j-th block of XCD label x takes coset j % nz of slab (j / nz) * 8 + x: the slab is
    // fetched from HBM once and the other cosets hit L2, instead of nz HBM reads spread over the launch.
    const u32 lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), x = lin & 7, j = lin >> 3, nz = gridDim.z;
    if (p.xcd_group == 2) {  // column before slab (see k_ntt_pass_pf)
      z = j % nz + p.z_base; col = (j / nz) % gridDim.y; wg = (j / (nz * gridDim.y)) * 8 + x;
    } else {
      const u32 sc = (j / nz) * 8 + x;
      z = j % nz + p.z_base;
      wg = sc % gridDim.x;
      col = sc / gridDim.x;
    }
  }
  pass.template load<INV>(lds, tid, NTT_THREADS, wg, col, z);
  __syncthreads();
  for (u32 si = 0; si < p.nsteps; si++) {
    pass.template step<INV>(lds, tid, NTT_THREADS, si);
    __syncthreads();
  }
  pass.template store<INV>(lds, tid, NTT_THREADS, wg, col, z);
}

// Workgroup barrier that orders LDS accesses only: the global loads of a prefetch and the stores of the previous slab stay in
// flight across it (__syncthreads() would drain the stores: its release fence waits vmcnt(0)).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// The passes of a large transform (2^13-element slabs: 3 + 3 + 3 strided bits over 16-element runs, or 3 + 3 + 3 + 4 contiguous
// bits) with SEVERAL slabs per workgroup and the next slab's elements prefetched into registers while the current one is in its
// register steps: with two workgroups per CU (the slab fills the LDS) nothing else hides the HBM latency of the load phase,
// which costs 18 % of the LDE when exposed.  FACTORS: the load multiplies by a one-level factor table (the coset scale of an
// LDE's first pass, the inter-group twiddle of an inverse strided pass).
template <bool INV, bool STRIDED, bool FACTORS>
__global__ __launch_bounds__(NTT_THREADS, 4) void k_ntt_pass_pf(NttPassParams p, u32 gx, u32 gy, u32 gz, u32 slabs_per_wg) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  NttPass pass{p};
  const u32 tid = threadIdx.x;
  auto coords = [&](u32 it, u32 &wg, u32 &col, u32 &z) {  // the slab of iteration `it`: the block mapping of k_ntt_pass
    const u32 lin = blockIdx.x + it * gridDim.x;
    if (p.xcd_group == 2) {
      // per XCD (label x = lin & 7) the slabs run coset-fastest, then column, then slab index: the 64 workgroups resident on an
      // XCD at a time are 8 cosets x 8 columns of ONE slab index, so the coefficient slab of a column is fetched once for its
      // 8 cosets, and the coset-scale rows and the inter-group twiddle row of that slab index - which depend on (z, wg) and on wg
      // only, not on the column - are fetched once for all the columns of the launch instead of once per column
      const u32 x = lin & 7, j = lin >> 3;
      z = j % gz + p.z_base; col = (j / gz) % gy; wg = (j / (gz * gy)) * 8 + x;
    } else if (p.xcd_group) {
      const u32 x = lin & 7, j = lin >> 3, sc = (j / gz) * 8 + x;
      z = j % gz + p.z_base; wg = sc % gx; col = sc / gx;
    } else {
      wg = lin % gx; col = (lin / gx) % gy; z = lin / (gx * gy) + p.z_base;
    }
  };
  u64 v[16];
  u32 wg, col, z;
  coords(0, wg, col, z);
  pass.prefetch(tid, NTT_THREADS, wg, col, z, v);
  for (u32 it = 0; it < slabs_per_wg; it++) {
    pass.template commit<INV, FACTORS>(lds, tid, NTT_THREADS, wg, z, v);
    lds_barrier();
    u32 nwg = wg, ncol = col, nz = z;
    if (it + 1 < slabs_per_wg) {
      coords(it + 1, nwg, ncol, nz);
      pass.prefetch(tid, NTT_THREADS, nwg, ncol, nz, v);
    }
    if (STRIDED) {  // S = 4, B = 9: steps of 3 bits at local bits 10, 7, 4
      if (!INV) {
        pass.template step_r<false, 3, 10>(lds, tid, NTT_THREADS, 8); lds_barrier();
        pass.template step_r<false, 3, 7>(lds, tid, NTT_THREADS, 5); lds_barrier();
        pass.template step_r<false, 3, 4>(lds, tid, NTT_THREADS, 2); lds_barrier();
      } else {
        pass.template step_r<true, 3, 4>(lds, tid, NTT_THREADS, 2); lds_barrier();
        pass.template step_r<true, 3, 7>(lds, tid, NTT_THREADS, 5); lds_barrier();
        pass.template step_r<true, 3, 10>(lds, tid, NTT_THREADS, 8); lds_barrier();
      }
    } else {        // S = 0, B = 13: 3 + 3 + 3 bits at 10, 7, 4 and the 4 bottom bits
      if (!INV) {
        pass.template step_r<false, 3, 10>(lds, tid, NTT_THREADS, 12); lds_barrier();
        pass.template step_r<false, 3, 7>(lds, tid, NTT_THREADS, 9); lds_barrier();
        pass.template step_r<false, 3, 4>(lds, tid, NTT_THREADS, 6); lds_barrier();
        pass.template step_r<false, 4, 0>(lds, tid, NTT_THREADS, 3); lds_barrier();
      } else {
        pass.template step_r<true, 4, 0>(lds, tid, NTT_THREADS, 3); lds_barrier();
        pass.template step_r<true, 3, 4>(lds, tid, NTT_THREADS, 6); lds_barrier();
        pass.template step_r<true, 3, 7>(lds, tid, NTT_THREADS, 9); lds_barrier();
        pass.template step_r<true, 3, 10>(lds, tid, NTT_THREADS, 12); lds_barrier();
      }
    }
    pass.template store<INV>(lds, tid, NTT_THREADS, wg, col, z);
    lds_barrier();  // the slab has been read out: the next commit may overwrite it
    wg = nwg; col = ncol; z = nz;
  }
}

__global__ __launch_bounds__(256) void k_bitrev_tile(BitrevTile b) {
  __shared__ u64 lds[64 * 65];
  b.load(lds, threadIdx.x, 256, blockIdx.x, blockIdx.y);
  __syncthreads();
  b.store(lds, threadIdx.x, 256, blockIdx.x, blockIdx.y);
}

__global__ void k_bitrev_small(const u64 *__restrict__ in, u64 is, u64 *__restrict__ out, u64 os, u32 lg) {
  u32 col = blockIdx.y;
  u32 n = 1u << lg;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    out[(u64)col * os + bitrev32(i, lg)] = gl_canon(in[(u64)col * is + i]);
}


void launch_ntt_pass(hipStream_t s, bool inverse, const NttPassParams &p, u32 wgs, u32 cols, u32 nz) {
  size_t lds_bytes = (size_t)8 * ntt_lds_words(p.L);
  // the prefetching form: the slab shapes of a large transform, factors (if any) from one-level tables, and enough slabs that
  // the shorter grid still fills the chip
  const u64 total = (u64)wgs * cols * nz;
  const bool strided = p.L == 13 && p.S == 4 && p.B == 9 && p.nsteps == 3 && p.step_plan == 0x333;
  const bool contiguous = p.L == 13 && p.S == 0 && p.B == 13 && p.nsteps == 4 && p.step_plan == 0x4333 && p.g_lo == 0;
  const bool factor = inverse ? p.g_lo != 0 : p.scale_mode != 0;
  const bool direct = inverse ? p.tw.h == NTT_DIRECT : (p.scale_mode == 2 && p.sc.h == NTT_DIRECT);
  // (forward passes only: the inverse instantiations need 30-odd more VGPRs than the 128 of four waves per SIMD and lose to the
  // plain kernel once they spill)
  static const int pf_mask = getenv("LCP2_NTT_PF") ? atoi(getenv("LCP2_NTT_PF")) : 3;  // debugging aid: bit 0 strided, bit 1 contiguous form
  if (!inverse && ((strided && (pf_mask & 1)) || (contiguous && (pf_mask & 2))) && (!factor || (direct && strided)) && total < (1ull << 31)) {
    // slabs per workgroup: only the first load of a workgroup is exposed, so as many as still leave 2048 workgroups (4 rounds
    // of the 512 resident ones) and divide the slab count into a multiple of 8 workgroups (the XCD mapping)
    u32 spw = 0;
    for (u32 k = 16; k >= 2 && !spw; k >>= 1)
      if (total % (k * 8) == 0 && total / k >= 2048) spw = k;
    if (spw) {
      const dim3 grid((u32)(total / spw));
#define LCP2_PF(INV, STR, FAC) hipLaunchKernelGGL((k_ntt_pass_pf<INV, STR, FAC>), grid, dim3(NTT_THREADS), lds_bytes, s, p, wgs, cols, nz, spw)
      if (strided) { if (factor) LCP2_PF(false, true, true); else LCP2_PF(false, true, false); }
      else LCP2_PF(false, false, false);
#undef LCP2_PF
      return;
    }
  }
  dim3 grid(wgs, cols, nz);
  if (inverse) hipLaunchKernelGGL(k_ntt_pass<true>, grid, dim3(NTT_THREADS), lds_bytes, s, p);
  else hipLaunchKernelGGL(k_ntt_pass<false>, grid, dim3(NTT_THREADS), lds_bytes, s, p);
}
void launch_bitrev_tile(hipStream_t s, const BitrevTile &b, u32 wgs, u32 cols) {
  hipLaunchKernelGGL(k_bitrev_tile, dim3(wgs, cols), dim3(256), 0, s, b);
}
void launch_bitrev_small(hipStream_t s, const u64 *in, u64 is, u64 *out, u64 os, u32 lg, u32 cols) {
  u32 n = 1u << lg;
  u32 blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_bitrev_small, dim3(blocks, cols), dim3(256), 0, s, in, is, out, os, lg);
}

}  // namespace lcp2
