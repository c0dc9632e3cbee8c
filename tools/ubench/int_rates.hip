// Micro-benchmark: issue rate of the integer VALU instructions the field arithmetic is built from (gfx950).
// Each kernel runs ITER x 16 independent instances of one instruction per lane; reports cycles per
// wave-instruction per SIMD assuming 2.4 GHz (the ratio between rows is what matters).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 2048
#define DEF_KERNEL(NAME, ASM, CONSTR_EXTRA)                                                             \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {                          \
    unsigned a[16];                                                                                     \
    unsigned b = seed + threadIdx.x, c = seed * 3 + 1;                                                  \
    for (int i = 0; i < 16; i++) a[i] = seed + i + threadIdx.x;                                         \
    for (int it = 0; it < ITER; it++) {                                                                 \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c));   \
    }                                                                                                   \
    unsigned s = 0;                                                                                     \
    for (int i = 0; i < 16; i++) s += a[i];                                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                     \
  }
DEF_KERNEL(k_add_u32, "v_add_u32 %0, %0, %1", )
DEF_KERNEL(k_add3_u32, "v_add3_u32 %0, %0, %1, %2", )
DEF_KERNEL(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1", )
DEF_KERNEL(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1", )
DEF_KERNEL(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1", )
DEF_KERNEL(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2", )
DEF_KERNEL(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1", )
DEF_KERNEL(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1", )
DEF_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2", )
DEF_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", )
DEF_KERNEL(k_mad_i32_i24, "v_mad_i32_i24 %0, %0, %1, %2", )
DEF_KERNEL(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %2", )
DEF_KERNEL(k_perm_b32, "v_perm_b32 %0, %0, %1, %2", )
DEF_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 11", )
DEF_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2", )
DEF_KERNEL(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1", )
DEF_KERNEL(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2", )
DEF_KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1", )

#define DEF_KERNEL64(NAME, ASM)                                                                         \
  __global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed) {                          \
    unsigned long long a[16];                                                                           \
    unsigned b = seed + threadIdx.x, c = seed * 3 + 1;                                                  \
    unsigned long long d = seed * 7ull + threadIdx.x;                                                   \
    for (int i = 0; i < 16; i++) a[i] = seed + i + threadIdx.x;                                         \
    for (int it = 0; it < ITER; it++) {                                                                 \
      _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "v"(d)); \
    }                                                                                                   \
    unsigned long long s = 0;                                                                           \
    for (int i = 0; i < 16; i++) s += a[i];                                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)s ^ (unsigned)(s >> 32);                     \
  }
DEF_KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
DEF_KERNEL64(k_mad_u64_u32_inl, "v_mad_u64_u32 %0, vcc, %1, 41, %0")          // multiplier as an inline constant (the Poseidon MDS form)
DEF_KERNEL64(k_mad_u64_u32_sdst, "v_mad_u64_u32 %0, s[10:11], %1, %2, %0")    // carry-out to an SGPR pair instead of vcc
DEF_KERNEL(k_add_co_u32, "v_add_co_u32 %0, vcc, %0, %1", )
DEF_KERNEL(k_addc_co_u32, "v_addc_co_u32 %0, vcc, %0, %1, vcc", )
DEF_KERNEL(k_cndmask_e32, "v_cndmask_b32_e32 %0, %0, %1, vcc", )
DEF_KERNEL(k_mov_b32, "v_mov_b32 %0, %1", )

DEF_KERNEL(k_dot2_u32_u16, "v_dot2_u32_u16 %0, %0, %1, %2", )
DEF_KERNEL(k_mul_lo_u32_inl, "v_mul_lo_u32 %0, %0, 41", )
DEF_KERNEL(k_s_nop1, "s_nop 1", )
DEF_KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 2, %3")
DEF_KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, 5, %0")
DEF_KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %3")
DEF_KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %3, %3")
DEF_KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %3, %3")

template <class K> void run(const char *name, K k, unsigned *d_out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  double wave_instr = (double)blocks * 4 * ITER * 16;            // wave-instructions issued
  double per_simd = wave_instr / (256.0 * 4);                     // per SIMD
  double cyc = ms * 1e-3 * 2.4e9 / per_simd;
  printf("%-20s %8.3f ms  %6.2f cycles/wave-instr/SIMD @2.4GHz\n", name, ms, cyc);
}
int main() {
  unsigned *d_out;
  hipMalloc(&d_out, 256 * 8 * 256 * 4);
#define RUN(K) run(#K, K, d_out)
  RUN(k_add_u32); RUN(k_add3_u32); RUN(k_lshl_add_u32); RUN(k_and_or); RUN(k_cndmask); RUN(k_alignbit); RUN(k_perm_b32);
  RUN(k_mul_lo_u32); RUN(k_mul_hi_u32); RUN(k_mul_u32_u24); RUN(k_mul_hi_u32_u24); RUN(k_mad_u32_u24); RUN(k_mad_i32_i24);
  RUN(k_dot4_u32_u8); RUN(k_pk_mul_lo_u16); RUN(k_pk_mad_u16); RUN(k_pk_add_u16);
  RUN(k_mad_u64_u32); RUN(k_mad_u64_u32_inl); RUN(k_mad_u64_u32_sdst); RUN(k_lshl_add_u64); RUN(k_lshlrev_b64);
  RUN(k_add_co_u32); RUN(k_addc_co_u32); RUN(k_cndmask_e32); RUN(k_mov_b32); RUN(k_dot2_u32_u16); RUN(k_mul_lo_u32_inl); RUN(k_s_nop1);
  RUN(k_fma_f32); RUN(k_pk_fma_f32); RUN(k_mul_f64); RUN(k_fma_f64);
  return 0;
}
