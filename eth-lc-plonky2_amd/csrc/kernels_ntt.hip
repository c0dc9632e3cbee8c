// NTT pass kernels for gfx950 (rows a2, a3, a4 of SURVEY.md section 8): see ntt.hpp for the
// decomposition.  One workgroup = one 2^L-element slab in LDS (up to 68 KiB with padding, so two
// workgroups share a CU's 160 KiB), NTT_THREADS threads, one barrier per register step (3-4 stages).
// Global traffic is one read and one write of the slab per pass, in runs of >= 128 bytes.
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

// The forward passes of a large transform (2^13-element slabs: 3 + 3 + 3 strided bits over 16-element runs, or 3 + 3 + 3 + 4
// contiguous bits) with SEVERAL slabs per workgroup and the next slab's elements prefetched into registers while the current one
// is in its register steps: with two workgroups per CU (the slab fills the LDS) nothing else hides the HBM latency of the load
// phase.  The first step runs on the prefetched registers and the last one stores from registers (strided) or through the wave's
// own LDS region (contiguous): see ntt.hpp.  Barriers per slab: 3 (strided) / 4 (contiguous), LDS round trips 2 / 4.
// FMODE: coset scale of an LDE's first pass (0 none, 1 one-level table, 2 computed from one table value per thread).
// SS: the strided shape (4: S 4, B 9, three steps; 7: S 7, B 6, two steps - the first step is followed by the last one directly).
template <bool STRIDED, int FMODE, u32 SS = 4>
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
      // only, not on the column - are shared by the columns in flight instead of fetched once per column
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
  u64 *twl = lds + ntt_lds_words(13);  // compact step twiddles behind the slab (ntt.hpp pf_stage_twiddles); the first barrier publishes them
  pass.template pf_stage_twiddles<STRIDED, SS>(twl, tid);
  if (STRIDED) lds_barrier();  // the first step of a strided pass reads them already
  // Everything a phase derives from the thread id alone - 64-bit twiddle addresses above all, two dozen per step - is loop invariant,
  // and hipcc hoists all of it out of the slab loop and then spills it (100 VGPRs of scratch, reloaded for every slab).  The thread
  // id is therefore made opaque once per phase: the addresses are recomputed (one or two instructions each) where they are used.
  auto fresh_tid = [&]() { u32 t = tid; asm volatile("" : "+v"(t)); return t; };
  for (u32 it = 0; it < slabs_per_wg; it++) {
    pass.template pf_first_step<STRIDED, FMODE, SS>(lds, fresh_tid(), wg, z, v, STRIDED ? twl : p.group_tw);
    lds_barrier();
    u32 nwg = wg, ncol = col, nz = z;
    if (it + 1 < slabs_per_wg) {
      coords(it + 1, nwg, ncol, nz);
      pass.prefetch(fresh_tid(), NTT_THREADS, nwg, ncol, nz, v);
    }
    if (STRIDED && SS == 7) {  // S = 7, B = 6: the remaining step at local bit 7
      u64 x[16];
      const u32 t = fresh_tid();
      pass.template pf_last_strided_read<7>(lds, t, x);
      lds_barrier();  // the slab has been read out: the next first step may overwrite it
      pass.template pf_last_strided_store<7>(t, wg, col, z, x);
    } else if (STRIDED) {  // S = 4, B = 9: the remaining steps at local bits 7 and 4
      pass.template pf_mid_step<7>(lds, fresh_tid(), 5, twl + 512);
      lds_barrier();
      u64 x[16];
      const u32 t = fresh_tid();
      pass.template pf_last_strided_read<4>(lds, t, x);
      lds_barrier();  // the slab has been read out: the next first step may overwrite it
      pass.template pf_last_strided_store<4>(t, wg, col, z, x);
    } else {        // S = 0, B = 13: 3 + 3 bits at 7 and 4, then the 4 bottom bits
      pass.template pf_mid_step<7>(lds, fresh_tid(), 9, twl);
      lds_barrier();
      pass.template pf_mid_step<4>(lds, fresh_tid(), 6, twl + 1024);
      lds_barrier();
      const u32 t = fresh_tid();
      pass.pf_last_contiguous(lds, t);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-local: the rows below were written by this wave (LDS is in order per wave)
      pass.pf_store_wave_rows(lds, t, wg, col, z);
      lds_barrier();  // the slab has been read out: the next first step may overwrite it
    }
    wg = nwg; col = ncol; z = nz;
  }
}

__global__ __launch_bounds__(256) void k_bitrev_tile(BitrevTile b) {
  __shared__ u64 lds[64 * 65];
  b.load(lds, threadIdx.x, 256, blockIdx.x, blockIdx.y);
  __syncthreads();
  b.store(lds, threadIdx.x, 256, blockIdx.x, blockIdx.y);
}

__global__ void k_bitrev_small(const u64 *__restrict__ in, u64 is, u64 *__restrict__ out, u64 os, u32 lg, unsigned long long *noncanonical) {
  u32 col = blockIdx.y;
  u32 n = 1u << lg;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const u64 v = in[(u64)col * is + i];
    if (noncanonical && v >= GL_P) *noncanonical = 1;
    out[(u64)col * os + bitrev32(i, lg)] = gl_canon(v);
  }
}
// out = canonical copy of in; the flag as above (witness buffers that do not pass through the bit-reversal of an iNTT)
__global__ void k_canon_copy(const u64 *in, u64 *out /* may be in */, u64 count, unsigned long long *noncanonical) {
  bool seen = false;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (u64)gridDim.x * blockDim.x) {
    const u64 v = in[i];
    seen = seen || v >= GL_P;
    if (out) out[i] = gl_canon(v);
  }
  if (seen && noncanonical) *noncanonical = 1;
}


void launch_ntt_pass(hipStream_t s, bool inverse, const NttPassParams &p, u32 wgs, u32 cols, u32 nz) {
  size_t lds_bytes = (size_t)8 * ntt_lds_words(p.L);
  const u64 total = (u64)wgs * cols * nz;
  const u32 spw = ntt_pf_slabs_per_wg(p, inverse, total);
  if (spw) {
    const dim3 grid((u32)(total / spw));
    lds_bytes += 8 * (ntt_pf_strided(p) ? NttPass::PF_TW_WORDS_STRIDED : NttPass::PF_TW_WORDS_CONTIGUOUS);  // 2 x 77 KiB still share a CU
#define LCP2_PF(STR, FM, SS) hipLaunchKernelGGL((k_ntt_pass_pf<STR, FM, SS>), grid, dim3(NTT_THREADS), lds_bytes, s, p, wgs, cols, nz, spw)
    if (ntt_pf_strided_s(p) == 7) {
      if (!p.scale_mode) LCP2_PF(true, 0, 7);
      else if (p.sc_step) LCP2_PF(true, 2, 7);
      else LCP2_PF(true, 1, 7);
    } else if (ntt_pf_strided(p)) {
      if (!p.scale_mode) LCP2_PF(true, 0, 4);
      else if (p.sc_step) LCP2_PF(true, 2, 4);
      else LCP2_PF(true, 1, 4);
    } else {
      LCP2_PF(false, 0, 4);
    }
#undef LCP2_PF
    return;
  }
  dim3 grid(wgs, cols, nz);
  if (inverse) hipLaunchKernelGGL(k_ntt_pass<true>, grid, dim3(NTT_THREADS), lds_bytes, s, p);
  else hipLaunchKernelGGL(k_ntt_pass<false>, grid, dim3(NTT_THREADS), lds_bytes, s, p);
}
void launch_bitrev_tile(hipStream_t s, const BitrevTile &b, u32 wgs, u32 cols) {
  hipLaunchKernelGGL(k_bitrev_tile, dim3(wgs, cols), dim3(256), 0, s, b);
}
void launch_bitrev_small(hipStream_t s, const u64 *in, u64 is, u64 *out, u64 os, u32 lg, u32 cols, unsigned long long *noncanonical) {
  u32 n = 1u << lg;
  u32 blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_bitrev_small, dim3(blocks, cols), dim3(256), 0, s, in, is, out, os, lg, noncanonical);
}
// `height` runs of `width` words, the pitches (in words) apart: row blocks out of / into whole columns
__global__ __launch_bounds__(256) void k_copy_2d(u64 *__restrict__ dst, u64 dst_pitch, const u64 *__restrict__ src, u64 src_pitch, u64 width) {
  const u64 h = blockIdx.y;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < width; i += (u64)gridDim.x * blockDim.x) dst[h * dst_pitch + i] = src[h * src_pitch + i];
}
void launch_copy_2d(hipStream_t s, u64 *dst, u64 dst_pitch, const u64 *src, u64 src_pitch, u64 width, u32 height) {
  if (!width || !height) return;
  const u64 want = (width + 255) / 256;
  hipLaunchKernelGGL(k_copy_2d, dim3((unsigned)(want < 4096 ? want : 4096), height), dim3(256), 0, s, dst, dst_pitch, src, src_pitch, width);
}

void launch_canon_copy(hipStream_t s, const u64 *in, u64 *out, u64 count, unsigned long long *noncanonical) {
  const u64 want = (count + 255) / 256;
  hipLaunchKernelGGL(k_canon_copy, dim3((unsigned)(want < 8192 ? (want ? want : 1) : 8192)), dim3(256), 0, s, in, out, count, noncanonical);
}

}  // namespace lcp2
