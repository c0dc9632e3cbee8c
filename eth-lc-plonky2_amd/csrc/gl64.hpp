// Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension for gfx950.
//
// Replaces plonky2_field 0.1.1 `GoldilocksField` / `QuadraticExtension`
// (un-vendored dependency of the reference, /root/reference/Cargo.lock:2425-2427;
// reached via `F = <C as GenericConfig<D>>::F`, eth-lc-plonky2/src/main.rs:74-76).
//
// Invariant: every value held between operations is CANONICAL (< p).  Inputs
// that come from outside the library go through canon() once at load.
// The header is plain C++ when not compiled by hipcc so that the kernel index
// logic can be exercised by the CPU emulation harness in tests/emu/.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LCP2_HD __host__ __device__ __forceinline__
#else
#define LCP2_HD inline
#endif

namespace lcp2 {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u64 GL_P = 0xFFFFFFFF00000001ull;
constexpr u64 GL_EPS = 0xFFFFFFFFull;  // 2^64 mod p
constexpr u64 GL_GENERATOR = 7;        // multiplicative generator = coset shift
constexpr u64 GL_ROOT_2_32 = 1753635133440165772ull;
constexpr u64 GL_W = 7;                // F_p[X]/(X^2 - 7)

LCP2_HD u64 gl_canon(u64 x) { return x >= GL_P ? x - GL_P : x; }

LCP2_HD u64 gl_add(u64 a, u64 b) {
  u64 s = a + b;
  // a,b < p: at most one wrap; subtracting p modulo 2^64 is exact in both cases
  return (s < a || s >= GL_P) ? s - GL_P : s;
}
LCP2_HD u64 gl_sub(u64 a, u64 b) {
  u64 d = a - b;
  return a < b ? d + GL_P : d;
}
LCP2_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
LCP2_HD u64 gl_dbl(u64 a) { return gl_add(a, a); }

LCP2_HD void gl_mul_wide(u64 a, u64 b, u64 &lo, u64 &hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  lo = a * b;
  hi = __umul64hi(a, b);
#else
  unsigned __int128 m = (unsigned __int128)a * b;
  lo = (u64)m;
  hi = (u64)(m >> 64);
#endif
}

// x = lo + 2^64*hi ; 2^64 = 2^32-1, 2^96 = -1 (mod p).  Accepts any 128-bit x.
LCP2_HD u64 gl_reduce128(u64 lo, u64 hi) {
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GL_EPS;
  u64 t1 = (hi_lo << 32) - hi_lo;  // hi_lo * (2^32 - 1)
  u64 r = t0 + t1;
  if (r < t0) r += GL_EPS;
  return gl_canon(r);
}
// ---- lazy forms: results are any u64 congruent to the field element (used inside Poseidon, where only the
// final state is canonicalised).  Inputs of gl_mul_nc / gl_sqr_nc may be any u64.
LCP2_HD u64 gl_reduce128_nc(u64 lo, u64 hi) {
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GL_EPS;
  u64 t1 = (hi_lo << 32) - hi_lo;
  u64 r = t0 + t1;
  if (r < t0) r += GL_EPS;
  return r;
}
#if defined(__HIP_DEVICE_COMPILE__)
// gfx950: (r1:r0) = (a1:a0) * (b1:b0) mod p, lazy result, any 32-bit halves in.  Measured (tools/ubench/int_rates, mulchain): every
// integer VALU op that is VOP3-encoded or touches vcc (v_mad_u64_u32, v_add_co / v_addc_co / v_sub_co, v_cndmask in either
// encoding) costs ~4.3 cycles per wave-instruction; only plain 32-bit VOP1/VOP2 ops (v_mov, v_add_u32) run at ~2.5.  So the
// multiply is written with as FEW instructions of the first kind as the arithmetic allows, which means multiply-adds wherever a
// carry chain would otherwise run:
//   product    p = a0 b0 ;  t = a0 b1 + p_hi ;  u = a1 b0 + t (carry c) ;  v = a1 b1 + (u_hi + c 2^32)      -> (v : u_lo : p_lo)
//              the partial products ride in the 64-bit addend of v_mad_u64_u32: 4 multiply-adds, one select for c, and the
//              register moves that make (p_hi, 0) and (u_hi, c) even-aligned pairs - no add-with-carry at all;
//   reduction  x = lo + hi0 (2^32 - 1) - hi1:   t = lo - hi1, on borrow t -= 2^32 - 1, then r = hi0 (2^32 - 1) + t, on carry
//              r += 2^32 - 1.  Both corrections are multiply-adds of a selected factor:
//                t + sel * 0x11111111 with sel = borrow ? -15 : 0   (v_mad_i64_i32: -15 * 0x11111111 = -(2^32 - 1), the 64-bit
//                                                                   sum wraps like the subtraction does)
//                r + c2 * (2^32 - 1)  with c2  = carry  ? 1 : 0     (v_mad_u64_u32; r < 2^64 - 2^33 + 1 after a wrap, no second carry)
// 12 such instructions and 2 moves (56.5 cycles per wave-multiply against 76.6 for the carry-chain form of rounds 2-4, which had 17;
// hipcc's own lowering of the portable form has 27).  All selects use inline constants, so no constant VGPRs are held.
// hipcc pads nothing inside an asm string, so the wait states it emits itself for the same pairs (a VALU write of vcc -> any VALU
// read of it, carry-in of v_addc / v_subb and v_cndmask in either encoding alike: 2 wait states) are written out as s_nop 1.
// tools/check_hazards.py disassembles the built library and fails the build if any such pair, hand-written or
// compiler-generated, has fewer.
__device__ __forceinline__ void gl_mul_halves(u32 a0, u32 a1, u32 b0, u32 b1, u32 &r0, u32 &r1) {
  const u64 p = (u64)a0 * b0;
  const u64 t = (u64)a0 * b1 + (p >> 32);  // < 2^64: (2^32 - 1)^2 + 2^32 - 1
  u64 u;
  u32 c;
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %1, 0, 1, vcc"
      : "=&v"(u), "=&v"(c) : "v"(a1), "v"(b0), "v"(t) : "vcc");
  const u64 v = (u64)a1 * b1 + (((u64)c << 32) | (u32)(u >> 32));  // < 2^64: the 128-bit product is (v : u_lo : p_lo)
  const u32 p0 = (u32)p, lo1 = (u32)u, hi0 = (u32)v, hi1 = (u32)(v >> 32);
  u32 t0, t1, sel;  // t = lo - hi1 ; sel = -15 on borrow
  asm("v_sub_co_u32 %0, vcc, %3, %4\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %2, 0, -15, vcc"
      : "=&v"(t0), "=&v"(t1), "=&v"(sel) : "v"(p0), "v"(hi1), "v"(lo1) : "vcc");
  u64 r = ((u64)t1 << 32) | t0;
  u32 c2;  // r = t - (borrow ? 2^32 - 1 : 0) + hi0 (2^32 - 1), c2 = its carry
  asm("v_mad_i64_i32 %0, vcc, %2, %3, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %4, -1, %0\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %1, 0, 1, vcc"
      : "+v"(r), "=&v"(c2) : "v"(sel), "s"(0x11111111u), "v"(hi0) : "vcc");
  u64 q;
  asm("v_mad_u64_u32 %0, vcc, %1, -1, %2" : "=v"(q) : "v"(c2), "v"(r) : "vcc");
  r0 = (u32)q;
  r1 = (u32)(q >> 32);
}
#endif

LCP2_HD u64 gl_mul_nc(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 r0, r1;
  gl_mul_halves((u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32), r0, r1);
  return ((u64)r1 << 32) | r0;
#else
  u64 lo, hi;
  gl_mul_wide(a, b, lo, hi);
  return gl_reduce128_nc(lo, hi);
#endif
}
// a^2 with three 32x32 products instead of four
LCP2_HD u64 gl_sqr_nc(u64 a) {
  u64 a0 = (u32)a, a1 = a >> 32;
  u64 p00 = a0 * a0, p01 = a0 * a1, p11 = a1 * a1;
  u64 lo = p00 + (p01 << 33);
  u64 hi = p11 + (p01 >> 31) + (lo < p00 ? 1 : 0);
  return gl_reduce128_nc(lo, hi);
}
// a: any u64, b: canonical
LCP2_HD u64 gl_add_nc(u64 a, u64 b) {
  u64 s = a + b;
  return s < a ? s + GL_EPS : s;
}

// a: any u64, b: canonical -> a - b, lazy (a - b + 2^64 = a - b + eps (mod p) on borrow, and then >= 2^32: no second borrow)
LCP2_HD u64 gl_sub_nc(u64 a, u64 b) {
  u64 d = a - b;
  return a < b ? d - GL_EPS : d;
}

LCP2_HD u64 gl_mul(u64 a, u64 b) { return gl_canon(gl_mul_nc(a, b)); }
LCP2_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }

LCP2_HD u64 gl_pow(u64 b, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_sqr(b);
    e >>= 1;
  }
  return r;
}
LCP2_HD u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }
LCP2_HD u64 gl_root_of_unity(unsigned k) {
  u64 r = GL_ROOT_2_32;
  for (unsigned i = k; i < 32; i++) r = gl_sqr(r);
  return r;
}

// x * 2^S mod p.  The 16th roots of unity of the field are powers of two (2 has order 192, 2^96 = -1; plonky2's w_16 = 2^156 = -2^60),
// which is what the register butterflies of the NTT multiply by.
//   0 < S <= 32: the product is the 128-bit value (x >> (64 - S)) : (x << S) and the compiler drops the hi_hi part of the reduction
//                (10 instructions);
//   32 < S < 96, S % 32 != 0, x CANONICAL: with y = x << (S % 32) = (y2 : y1 : y0) in 32-bit words,
//        S / 32 = 1:  y * 2^32 = y2 2^96 + y1 2^64 + y0 2^32 = (y0 + y1) 2^32 - (y1 + y2)
//        S / 32 = 2:  y * 2^64 = y2 2^128 + y1 2^96 + y0 2^64 = y0 2^32 - (y2 2^32 + y1 + y0)
//      both are one canonical subtraction X - Z of canonical values (12 instructions against 19 for the generic shift-reduce
//      and 21 for a general multiply by the constant).
template <unsigned S>
LCP2_HD u64 gl_shl(u64 x) {
  static_assert(S > 0 && S < 96 && (S <= 32 || S % 32 != 0), "shift out of range");
  if constexpr (S <= 32) {
    return gl_reduce128(x << S, x >> (64 - S));
  } else {
    constexpr unsigned t = S % 32;
    const u32 y0 = (u32)x << t, y1 = (u32)(x >> (32 - t)), y2 = (u32)(x >> (64 - t));
    if constexpr (S < 64) {
      const u32 s = y0 + y1;
      const u32 carry = s < y0;  // (y0 + y1) 2^32 = s 2^32 + carry (2^32 - 1): canonical, s = 2^32 - 1 excludes a carry
      return gl_sub(((u64)s << 32) | (u32)(0u - carry), (u64)y1 + y2);
    } else {
      return gl_sub((u64)y0 << 32, ((u64)y2 << 32) + (u64)y1 + (u64)y0);
    }
  }
}

// the lazy form of a short shift (any u64 in, any u64 congruent to x 2^S out)
template <unsigned S>
LCP2_HD u64 gl_shl_nc(u64 x) {
  static_assert(S > 0 && S <= 32, "shift out of range");
  return gl_reduce128_nc(x << S, x >> (64 - S));
}
// x * c for a 32-bit constant c: two 32 x 32 multiply-adds give the 96-bit product, one fold reduces it (x: any u64)
LCP2_HD u64 gl_mul_u32_nc(u64 x, u32 c) {
  const u64 p0 = (u64)(u32)x * c, p1 = (x >> 32) * c + (p0 >> 32);  // p1 < 2^64: (2^32 - 1)^2 + 2^32 - 1
  return gl_reduce128_nc((p1 << 32) | (u32)p0, p1 >> 32);
}
LCP2_HD u64 gl_mul_u32(u64 x, u32 c) { return gl_canon(gl_mul_u32_nc(x, c)); }

struct gl2 {
  u64 c0, c1;
};
LCP2_HD gl2 gl2_make(u64 a, u64 b) { gl2 r; r.c0 = a; r.c1 = b; return r; }
LCP2_HD gl2 gl2_add(gl2 a, gl2 b) { return gl2_make(gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)); }
LCP2_HD gl2 gl2_sub(gl2 a, gl2 b) { return gl2_make(gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)); }
LCP2_HD gl2 gl2_mul(gl2 a, gl2 b) {
  u64 c0 = gl_add(gl_mul(a.c0, b.c0), gl_mul(GL_W, gl_mul(a.c1, b.c1)));
  u64 c1 = gl_add(gl_mul(a.c0, b.c1), gl_mul(a.c1, b.c0));
  return gl2_make(c0, c1);
}
LCP2_HD gl2 gl2_scale(gl2 a, u64 s) { return gl2_make(gl_mul(a.c0, s), gl_mul(a.c1, s)); }
LCP2_HD gl2 gl2_add_base(gl2 a, u64 b) { return gl2_make(gl_add(a.c0, b), a.c1); }
LCP2_HD gl2 gl2_sub_base(gl2 a, u64 b) { return gl2_make(gl_sub(a.c0, b), a.c1); }
LCP2_HD bool gl2_eq(gl2 a, gl2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }
LCP2_HD gl2 gl2_inv(gl2 a) {
  u64 n = gl_sub(gl_sqr(a.c0), gl_mul(GL_W, gl_sqr(a.c1)));
  u64 ni = gl_inv(n);
  return gl2_make(gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni));
}
LCP2_HD gl2 gl2_pow(gl2 b, u64 e) {
  gl2 r = gl2_make(1, 0);
  while (e) {
    if (e & 1) r = gl2_mul(r, b);
    b = gl2_mul(b, b);
    e >>= 1;
  }
  return r;
}

LCP2_HD u32 bitrev32(u32 x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
  u32 r = 0;
  for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
#endif
}

}  // namespace lcp2
