// Micro-benchmark of the 64 x 64 -> 64 Goldilocks multiply (csrc/gl64.hpp gl_mul_halves), four forms, results compared:
//   current   the carry-chain form of rounds 2-4: four v_mad_u64_u32, the 128-bit product summed by add-with-carry, the reduction on carry
//             chains (17 instructions)
//   chained   the partial products ride in the 64-bit addend of v_mad_u64_u32 (p = a0 b0; t = a0 b1 + p_hi; u = a1 b0 + t (carry c);
//             v = a1 b1 + (u_hi + c 2^32)): the three add-with-carry of the product become register moves
//   chained2  + both conditional corrections of the reduction as multiply-adds of a selected factor (selects in VOP2, constants in VGPRs)
//   chained3  + selects in VOP3 with inline constants, no constant VGPRs: what gl_mul_halves is now (12 instructions + 2 moves)
// 12 independent chains of dependent multiplies per lane (x <- x * y, the shape of a full round's S-box layer).
//     hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mulchain tools/ubench/mulchain.hip && tools/ubench/mulchain [waves per SIMD: 1 2 3 4 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../eth-lc-plonky2_amd/csrc/gl64.hpp"
using namespace lcp2;
#define ITER 512
#define CH 12

#if defined(__HIP_DEVICE_COMPILE__)
// the carry-chain form of rounds 2-4 (17 instructions), kept here as the baseline
__device__ __forceinline__ void mul_carry_chain(u32 a0, u32 a1, u32 b0, u32 b1, u32 &r0, u32 &r1, u32 k1, u32 km1) {
  u64 p = (u64)a0 * b0, m = (u64)a0 * b1, h = (u64)a1 * b1;
  u32 c;
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e32 %1, 0, %4, vcc"
      : "+v"(m), "=v"(c) : "v"(a1), "v"(b0), "v"(k1) : "vcc");
  const u32 p0 = (u32)p, p1 = (u32)(p >> 32), m0 = (u32)m, m1 = (u32)(m >> 32), h0 = (u32)h, h1 = (u32)(h >> 32);
  u32 lo1, hi0, hi1;  // 128-bit product = (hi1:hi0:lo1:p0)
  asm("v_add_co_u32 %0, vcc, %3, %4\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %1, vcc, %5, %6, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %2, vcc, %7, %8, vcc"
      : "=&v"(lo1), "=&v"(hi0), "=&v"(hi1) : "v"(p1), "v"(m0), "v"(h0), "v"(m1), "v"(h1), "v"(c) : "vcc");
  u32 t0, t1, e;  // t = lo - hi1 ; on borrow t -= 2^32 - 1
  asm("v_sub_co_u32 %0, vcc, %3, %4\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e32 %2, 0, %6, vcc\n\t"
      "v_sub_co_u32 %0, vcc, %0, %2\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
      : "=&v"(t0), "=&v"(t1), "=&v"(e) : "v"(p0), "v"(hi1), "v"(lo1), "v"(km1) : "vcc");
  u64 t = ((u64)t1 << 32) | t0, r;  // r = hi0 * (2^32 - 1) + t ; on carry r += 2^32 - 1
  u32 e2;
  asm("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e32 %1, 0, %4, vcc"
      : "=&v"(r), "=v"(e2) : "v"(hi0), "v"(t), "v"(km1) : "vcc");
  u32 q0 = (u32)r, q1 = (u32)(r >> 32);
  asm("v_add_co_u32 %0, vcc, %2, %4\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %1, vcc, 0, %3, vcc"
      : "=&v"(r0), "=&v"(r1) : "v"(q0), "v"(q1), "v"(e2) : "vcc");
}

__device__ __forceinline__ void mul_chained(u32 a0, u32 a1, u32 b0, u32 b1, u32 &r0, u32 &r1, u32 k1, u32 km1) {
  const u64 p = (u64)a0 * b0;
  const u64 t = (u64)a0 * b1 + (p >> 32);  // < 2^64: (2^32 - 1)^2 + 2^32 - 1
  u64 u;
  u32 c;
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e32 %1, 0, %5, vcc"
      : "=&v"(u), "=&v"(c) : "v"(a1), "v"(b0), "v"(t), "v"(k1) : "vcc");
  const u64 v = (u64)a1 * b1 + (((u64)c << 32) | (u32)(u >> 32));  // < 2^64
  const u32 p0 = (u32)p, lo1 = (u32)u, hi0 = (u32)v, hi1 = (u32)(v >> 32);
  u32 t0, t1, e;  // t = lo - hi1 ; on borrow t -= 2^32 - 1
  asm("v_sub_co_u32 %0, vcc, %3, %4\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e32 %2, 0, %6, vcc\n\t"
      "v_sub_co_u32 %0, vcc, %0, %2\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
      : "=&v"(t0), "=&v"(t1), "=&v"(e) : "v"(p0), "v"(hi1), "v"(lo1), "v"(km1) : "vcc");
  u64 tt = ((u64)t1 << 32) | t0, r;
  u32 e2;
  asm("v_mad_u64_u32 %0, vcc, %2, -1, %3\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e32 %1, 0, %4, vcc"
      : "=&v"(r), "=v"(e2) : "v"(hi0), "v"(tt), "v"(km1) : "vcc");
  u32 q0 = (u32)r, q1 = (u32)(r >> 32);
  asm("v_add_co_u32 %0, vcc, %2, %4\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %1, vcc, 0, %3, vcc"
      : "=&v"(r0), "=&v"(r1) : "v"(q0), "v"(q1), "v"(e2) : "vcc");
}
// variant 2: chained product, and the two conditional corrections of the reduction as multiply-adds:
//   t -= (2^32 - 1) on borrow   ==  t + sel * 65537 with sel = borrow ? -65535 : 0   (v_mad_i64_i32: -65535 * 65537 = -(2^32 - 1))
//   r += (2^32 - 1) on carry    ==  r + c * (2^32 - 1)  with c = carry ? 1 : 0        (v_mad_u64_u32)
__device__ __forceinline__ void mul_chained2(u32 a0, u32 a1, u32 b0, u32 b1, u32 &r0, u32 &r1, u32 k1, u32 km1, u32 kn, u32 sk) {
  const u64 p = (u64)a0 * b0;
  const u64 t = (u64)a0 * b1 + (p >> 32);
  u64 u;
  u32 c;
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e32 %1, 0, %5, vcc"
      : "=&v"(u), "=&v"(c) : "v"(a1), "v"(b0), "v"(t), "v"(k1) : "vcc");
  const u64 v = (u64)a1 * b1 + (((u64)c << 32) | (u32)(u >> 32));
  const u32 p0 = (u32)p, lo1 = (u32)u, hi0 = (u32)v, hi1 = (u32)(v >> 32);
  u32 t0, t1, sel;
  asm("v_sub_co_u32 %0, vcc, %3, %4\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e32 %2, 0, %6, vcc"
      : "=&v"(t0), "=&v"(t1), "=&v"(sel) : "v"(p0), "v"(hi1), "v"(lo1), "v"(kn) : "vcc");
  u64 tt = ((u64)t1 << 32) | t0, r;
  u32 c2;
  asm("v_mad_i64_i32 %0, vcc, %2, %3, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %4, -1, %0\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e32 %1, 0, %5, vcc"
      : "+v"(tt), "=&v"(c2) : "v"(sel), "s"(sk), "v"(hi0), "v"(k1) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, -1, %2" : "=v"(r) : "v"(c2), "v"(tt) : "vcc");
  r0 = (u32)r; r1 = (u32)(r >> 32);
}
// variant 3: variant 2 with the selects in the VOP3 encoding (inline constants 1 and -15, 0x11111111 in an SGPR): no constant VGPRs
__device__ __forceinline__ void mul_chained3(u32 a0, u32 a1, u32 b0, u32 b1, u32 &r0, u32 &r1, u32 sk) {
  const u64 p = (u64)a0 * b0;
  const u64 t = (u64)a0 * b1 + (p >> 32);
  u64 u;
  u32 c;
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %1, 0, 1, vcc"
      : "=&v"(u), "=&v"(c) : "v"(a1), "v"(b0), "v"(t) : "vcc");
  const u64 v = (u64)a1 * b1 + (((u64)c << 32) | (u32)(u >> 32));
  const u32 p0 = (u32)p, lo1 = (u32)u, hi0 = (u32)v, hi1 = (u32)(v >> 32);
  u32 t0, t1, sel;
  asm("v_sub_co_u32 %0, vcc, %3, %4\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %2, 0, -15, vcc"
      : "=&v"(t0), "=&v"(t1), "=&v"(sel) : "v"(p0), "v"(hi1), "v"(lo1) : "vcc");
  u64 tt = ((u64)t1 << 32) | t0, r;
  u32 c2;
  asm("v_mad_i64_i32 %0, vcc, %2, %3, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %4, -1, %0\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %1, 0, 1, vcc"
      : "+v"(tt), "=&v"(c2) : "v"(sel), "s"(sk), "v"(hi0) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, -1, %2" : "=v"(r) : "v"(c2), "v"(tt) : "vcc");
  r0 = (u32)r; r1 = (u32)(r >> 32);
}
#endif


template <int V>
__global__ __launch_bounds__(256) void k_chain(u64 *out, u64 seed) {
  extern __shared__ u64 lds_pad[];  // only there to limit the waves per SIMD
  if (seed == 1) lds_pad[threadIdx.x] = seed;
#if defined(__HIP_DEVICE_COMPILE__)
  u32 k1, km1;
  asm volatile("v_mov_b32 %0, 1\n\tv_mov_b32 %1, -1" : "=v"(k1), "=v"(km1));
  u32 kn, sk;
  asm volatile("v_mov_b32 %0, 0xffff0001\n\ts_mov_b32 %1, 0x10001" : "=v"(kn), "=s"(sk));
  u32 sk3;
  asm volatile("s_mov_b32 %0, 0x11111111" : "=s"(sk3));
  u32 x0[CH], x1[CH], y0[CH], y1[CH];
  for (int i = 0; i < CH; i++) {
    const u64 a = seed * (2 * i + 3) + threadIdx.x * 0x9E3779B97F4A7C15ull + blockIdx.x, b = seed * (2 * i + 5) ^ (a >> 7);
    x0[i] = (u32)a; x1[i] = (u32)(a >> 32); y0[i] = (u32)b; y1[i] = (u32)(b >> 32);
  }
#pragma unroll 1
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < CH; i++) {
      if (V == 0) mul_carry_chain(x0[i], x1[i], y0[i], y1[i], x0[i], x1[i], k1, km1);
      else if (V == 1) mul_chained(x0[i], x1[i], y0[i], y1[i], x0[i], x1[i], k1, km1);
      else if (V == 2) mul_chained2(x0[i], x1[i], y0[i], y1[i], x0[i], x1[i], k1, km1, kn, sk);
      else mul_chained3(x0[i], x1[i], y0[i], y1[i], x0[i], x1[i], sk3);
    }
  }
  u64 s = 0;
  for (int i = 0; i < CH; i++) s ^= gl_canon(((u64)x1[i] << 32) | x0[i]) * (i + 1);
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
#endif
}

template <int V>
static void launch(int blocks, int threads, size_t lds, u64 *d) {
  hipFuncSetAttribute(reinterpret_cast<const void *>(&k_chain<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_chain<V>, dim3(blocks), dim3(threads), lds, 0, d, 12345ull);
}

int main(int argc, char **argv) {
  const int blocks = 256 * 8, threads = 256;
  const int waves = argc > 1 ? atoi(argv[1]) : 8;  // waves per SIMD, set through the LDS a block asks for
  const size_t lds = waves >= 8 ? 0 : (size_t)(160 * 1024 / waves) & ~(size_t)255;
  printf("%d waves per SIMD (%zu bytes of LDS per 4-wave block)\n", waves, lds);
  u64 *d[4];
  std::vector<u64> h[4];
  for (int v = 0; v < 4; v++) { hipMalloc(&d[v], (size_t)blocks * threads * 8); h[v].resize((size_t)blocks * threads); }
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; rep++)
    for (int v = 0; v < 4; v++) {
      hipEventRecord(a);
      if (v == 0) launch<0>(blocks, threads, lds, d[v]);
      else if (v == 1) launch<1>(blocks, threads, lds, d[v]);
      else if (v == 2) launch<2>(blocks, threads, lds, d[v]);
      else launch<3>(blocks, threads, lds, d[v]);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      const double mults = (double)blocks * threads * CH * ITER, waves = mults / 64;
      if (rep == 2) printf("%-8s %8.3f ms  %.2f cycles per wave-multiply per SIMD @2.4GHz\n", v == 0 ? "current" : v == 1 ? "chained" : v == 2 ? "chained2" : "chained3", ms, ms * 1e-3 * 2.4e9 * 1024 / waves);
    }
  for (int v = 0; v < 4; v++) hipMemcpy(h[v].data(), d[v], h[v].size() * 8, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < h[0].size(); i++) bad += (h[0][i] != h[1][i]) + (h[0][i] != h[2][i]) + (h[0][i] != h[3][i]);
  printf("results %s (%zu of %zu differ)\n", bad ? "DIFFER" : "equal", bad, h[0].size());
  return bad != 0;
}
