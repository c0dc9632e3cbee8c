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
    const u32 sc = (j / nz) * 8 + x;
    z = j % nz + p.z_base;
    wg = sc % gridDim.x;
    col = sc / gridDim.x;
  }
  pass.template load<INV>(lds, tid, NTT_THREADS, wg, col, z);
  __syncthreads();
  for (u32 si = 0; si < p.nsteps; si++) {
    pass.template step<INV>(lds, tid, NTT_THREADS, si);
    __syncthreads();
  }
  pass.template store<INV>(lds, tid, NTT_THREADS, wg, col, z);
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
