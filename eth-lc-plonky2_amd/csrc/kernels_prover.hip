// Prover kernels K5 - K9 for gfx950 (rows a7 - a13 of SURVEY.md section 8).
//
//  K5  k_perm_chunks / scan / k_perm_finalize   wires_permutation_partial_products_and_zs  (plonk/prover.rs)
//  K6  k_quotient                                compute_quotient_polys + eval_vanishing_poly_base_batch
//                                                (+ every gate's eval_unfiltered_base via the gate-program interpreter)
//  K7  k_eval_polys / k_compose / k_divide_*     OpeningSet::new, PolynomialBatch::prove_openings (fri/oracle.rs)
//  K8  k_fri_fold (+ NTT, hash kernels)          fri_committed_trees (fri/prover.rs)
//  K9  k_pow_search                              fri_proof_of_work, deterministic minimum witness
//
// Every kernel indexes the LDE matrices in their storage (= Merkle leaf) order, so all column reads are
// coalesced 512-byte runs per wave; the only gathers are the two Z(g x) values per point in K6.
#include "internal.hpp"
#include "poseidon.hpp"
#include "prover_kernels.hpp"

namespace lcp2 {

// ------------------------------------------------------------------ generic exclusive scan (field add / mul)
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 4, SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;

template <bool MUL> __device__ __forceinline__ u64 scan_op(u64 a, u64 b) { return MUL ? gl_mul(a, b) : gl_add(a, b); }
template <bool MUL> __device__ __forceinline__ u64 scan_id() { return MUL ? 1 : 0; }

// out[i] = op over logical predecessors of i (exclusive); logical index = reverse ? n-1-i : i; batch = blockIdx.y
template <bool MUL>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_block(const u64 *__restrict__ in, u64 *__restrict__ out, u64 *__restrict__ block_tot,
                                                             u64 n, int reverse, u64 batch_stride, u64 nblocks) {
  __shared__ u64 sh[SCAN_THREADS];
  const u64 *src = in + blockIdx.y * batch_stride;
  u64 *dst = out + blockIdx.y * batch_stride;
  u64 base = (u64)blockIdx.x * SCAN_BLOCK + (u64)threadIdx.x * SCAN_ITEMS;
  u64 v[SCAN_ITEMS];
  u64 run = scan_id<MUL>();
#pragma unroll
  for (int e = 0; e < SCAN_ITEMS; e++) {
    u64 li = base + e;
    u64 x = scan_id<MUL>();
    if (li < n) x = src[reverse ? n - 1 - li : li];
    v[e] = run;  // exclusive inside the thread
    run = scan_op<MUL>(run, x);
  }
  sh[threadIdx.x] = run;
  __syncthreads();
  // Hillis-Steele inclusive scan of the 256 thread totals
  for (int off = 1; off < SCAN_THREADS; off <<= 1) {
    u64 t = sh[threadIdx.x];
    u64 o = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : scan_id<MUL>();
    __syncthreads();
    sh[threadIdx.x] = scan_op<MUL>(o, t);
    __syncthreads();
  }
  u64 prefix = threadIdx.x ? sh[threadIdx.x - 1] : scan_id<MUL>();
#pragma unroll
  for (int e = 0; e < SCAN_ITEMS; e++) {
    u64 li = base + e;
    if (li < n) dst[reverse ? n - 1 - li : li] = scan_op<MUL>(prefix, v[e]);
  }
  if (threadIdx.x == SCAN_THREADS - 1) block_tot[blockIdx.y * nblocks + blockIdx.x] = sh[SCAN_THREADS - 1];
}
// exclusive scan of the block totals, one workgroup per batch (sequential over 256-wide tiles)
template <bool MUL>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_totals(u64 *__restrict__ block_tot, u64 nblocks) {
  __shared__ u64 sh[SCAN_THREADS];
  u64 *t = block_tot + blockIdx.x * nblocks;
  u64 carry = scan_id<MUL>();
  for (u64 base = 0; base < nblocks; base += SCAN_THREADS) {
    u64 i = base + threadIdx.x;
    u64 x = i < nblocks ? t[i] : scan_id<MUL>();
    sh[threadIdx.x] = x;
    __syncthreads();
    for (int off = 1; off < SCAN_THREADS; off <<= 1) {
      u64 cur = sh[threadIdx.x];
      u64 o = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : scan_id<MUL>();
      __syncthreads();
      sh[threadIdx.x] = scan_op<MUL>(o, cur);
      __syncthreads();
    }
    u64 excl = threadIdx.x ? sh[threadIdx.x - 1] : scan_id<MUL>();
    u64 total = sh[SCAN_THREADS - 1];
    if (i < nblocks) t[i] = scan_op<MUL>(carry, excl);
    carry = scan_op<MUL>(carry, total);
    __syncthreads();
  }
}
template <bool MUL>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(u64 *__restrict__ out, const u64 *__restrict__ block_tot, u64 n, int reverse,
                                                             u64 batch_stride, u64 nblocks) {
  u64 *dst = out + blockIdx.y * batch_stride;
  u64 pre = block_tot[blockIdx.y * nblocks + blockIdx.x];
  u64 base = (u64)blockIdx.x * SCAN_BLOCK;
  for (u32 e = threadIdx.x; e < SCAN_BLOCK; e += SCAN_THREADS) {
    u64 li = base + e;
    if (li < n) { u64 ph = reverse ? n - 1 - li : li; dst[ph] = scan_op<MUL>(pre, dst[ph]); }
  }
}

void launch_scan(hipStream_t s, bool mul, const u64 *in, u64 *out, u64 *block_tot, u64 n, bool reverse, u32 batches, u64 batch_stride) {
  u64 nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  dim3 grid((unsigned)nblocks, batches);
  if (mul) {
    hipLaunchKernelGGL(k_scan_block<true>, grid, dim3(SCAN_THREADS), 0, s, in, out, block_tot, n, (int)reverse, batch_stride, nblocks);
    hipLaunchKernelGGL(k_scan_totals<true>, dim3(batches), dim3(SCAN_THREADS), 0, s, block_tot, nblocks);
    hipLaunchKernelGGL(k_scan_apply<true>, grid, dim3(SCAN_THREADS), 0, s, out, block_tot, n, (int)reverse, batch_stride, nblocks);
  } else {
    hipLaunchKernelGGL(k_scan_block<false>, grid, dim3(SCAN_THREADS), 0, s, in, out, block_tot, n, (int)reverse, batch_stride, nblocks);
    hipLaunchKernelGGL(k_scan_totals<false>, dim3(batches), dim3(SCAN_THREADS), 0, s, block_tot, nblocks);
    hipLaunchKernelGGL(k_scan_apply<false>, grid, dim3(SCAN_THREADS), 0, s, out, block_tot, n, (int)reverse, batch_stride, nblocks);
  }
}
u64 scan_scratch_words(u64 n, u32 batches) { return ((n + SCAN_BLOCK - 1) / SCAN_BLOCK) * batches; }

// ------------------------------------------------------------------ K5: permutation argument on H
// One thread per (row, challenge): the NCHUNK quotient-chunk products  prod_j num_j / den_j  and their product.
__global__ __launch_bounds__(256) void k_perm_chunks(PermArgs a) {
  u32 rb = blockIdx.x, ch = blockIdx.y;
  if (gridDim.x % 8 == 0) {
    // XCD-aware mapping (speed only; blocks b and b + 8 share an XCD's L2): the challenges of one row block run back to back on
    // one XCD, so the wires and sigmas they both read come from HBM once
    const u32 lin = blockIdx.x + gridDim.x * blockIdx.y, x = lin & 7, j = lin >> 3;
    ch = j % gridDim.y;
    rb = (j / gridDim.y) * 8 + x;
  }
  u64 row = (u64)rb * blockDim.x + threadIdx.x;
  if (row >= a.n) return;
  const u64 beta = a.betas[ch], gamma = a.gammas[ch];
  u64 x = two_level(a.subgroup, row);
  u64 bx = gl_mul(beta, x);
  u64 pn[PERM_MAX_CHUNKS], pd[PERM_MAX_CHUNKS];
#pragma unroll
  for (u32 k = 0; k < PERM_MAX_CHUNKS; k++) {
    pn[k] = 1; pd[k] = 1;
    if (k < a.nchunks) {
      for (u32 j = k * a.chunk; j < a.num_routed && j < (k + 1) * a.chunk; j++) {
        u64 w = gl_canon(a.wires[(u64)j * a.n + row]);
        u64 wg = gl_add(w, gamma);
        pn[k] = gl_mul(pn[k], gl_add(wg, gl_mul(bx, a.k_is[j])));
        pd[k] = gl_mul(pd[k], gl_add(wg, gl_mul(beta, a.sigmas[(u64)j * a.n + row])));
      }
    }
  }
  // Montgomery batch inversion of the chunk denominators
  u64 pre[PERM_MAX_CHUNKS];
  u64 acc = 1;
#pragma unroll
  for (u32 k = 0; k < PERM_MAX_CHUNKS; k++) { pre[k] = acc; acc = gl_mul(acc, pd[k]); }
  u64 inv = gl_inv(acc);
  u64 tot = 1;
#pragma unroll
  for (int k = PERM_MAX_CHUNKS - 1; k >= 0; k--) {
    u64 dinv = gl_mul(inv, pre[k]);
    inv = gl_mul(inv, pd[k]);
    pn[k] = gl_mul(pn[k], dinv);  // quotient chunk product
  }
#pragma unroll
  for (u32 k = 0; k < PERM_MAX_CHUNKS; k++)
    if (k < a.nchunks) { a.chunk_q[((u64)ch * a.nchunks + k) * a.n + row] = pn[k]; tot = gl_mul(tot, pn[k]); }
  a.row_tot[(u64)ch * a.n + row] = tot;
}
// Z (exclusive prefix product of the row totals) is in zs[ch]; partial products pp_k = Z * q_0 .. q_k
__global__ __launch_bounds__(256) void k_perm_finalize(PermArgs a) {
  u64 row = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 ch = blockIdx.y;
  if (row >= a.n) return;
  u64 acc = a.zs_out[(u64)ch * a.n + row];
  u32 npp = a.nchunks - 1;
  for (u32 k = 0; k < npp; k++) {
    acc = gl_mul(acc, a.chunk_q[((u64)ch * a.nchunks + k) * a.n + row]);
    a.zs_out[((u64)a.num_challenges + (u64)ch * npp + k) * a.n + row] = acc;
  }
}
void launch_perm_chunks(hipStream_t s, const PermArgs &a) {
  hipLaunchKernelGGL(k_perm_chunks, dim3((unsigned)((a.n + 255) / 256), a.num_challenges), dim3(256), 0, s, a);
}
void launch_perm_finalize(hipStream_t s, const PermArgs &a) {
  hipLaunchKernelGGL(k_perm_finalize, dim3((unsigned)((a.n + 255) / 256), a.num_challenges), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------ K6: quotient polynomial values on the LDE coset
// Gate-program interpreter: registers live in LDS (reg r of thread t at lds[r * T + t]: conflict free), decode is
// wave-uniform (scalar unit), operands come from the wires / constants LDE columns at this thread's point.
__device__ __forceinline__ u64 q_operand(const QuotientArgs &a, u32 kind, u32 idx, const u64 *lds, u32 T, u32 tid, u64 i) {
  switch (kind) {
    case 0: return lds[idx * T + tid];
    case 1: return a.wires[(u64)idx * a.stride + i];
    case 2: return a.consts[(u64)(a.num_selectors + idx) * a.stride + i];
    case 3: return a.imm[idx];
    default: return a.pis[idx];
  }
}

__global__ __launch_bounds__(QUOTIENT_THREADS) void k_quotient(QuotientArgs a) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 T = QUOTIENT_THREADS, tid = threadIdx.x;
  const u64 i = (u64)blockIdx.x * T + tid;  // local storage (leaf) index; global index = a.leaf0 + i
  if (i >= a.count) return;                  // no barrier is used below
  const u64 ig = a.leaf0 + i;
  const u32 CH = a.num_challenges;
  u64 res[QUOTIENT_MAX_CH];
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) res[c] = 0;

  // ---- gate constraints: sum_g filter_g * sum_i alpha^i c_{g,i}
  for (u32 g = 0; g < a.num_gates; g++) {
    const GateDev G = a.gates[g];
    u64 acc[QUOTIENT_MAX_CH];
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) acc[c] = 0;
    // the instruction words are wave-uniform scalar loads: fetch one instruction ahead so that the scalar-cache round
    // trip overlaps the arithmetic of the current instruction (the code array is padded by one instruction)
    const uint2 *code2 = (const uint2 *)a.code;
    uint2 nxt = code2[G.code_offset];
    for (u32 pc = G.code_offset; pc < G.code_offset + G.code_len; pc++) {
      const u32 w0 = nxt.x, w1 = nxt.y;
      nxt = code2[pc + 1];
      const u32 op = w0 & 0xF, dst = (w0 >> 8) & 0xFF, ka = (w0 >> 16) & 0xF, kb = (w0 >> 20) & 0xF, ia = w1 & 0xFFFF, ib = w1 >> 16;
      u64 x = q_operand(a, ka, ia, lds, T, tid, i);
      if (op == LCP2_OP_EMIT || op == LCP2_OP_EMITBOOL) {
        if (op == LCP2_OP_EMITBOOL) x = gl_sub(gl_mul(x, x), x);
#pragma unroll
        for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
          if (c < CH) acc[c] = gl_add(gl_mul(acc[c], a.alphas[c]), x);
        continue;
      }
      u64 y = q_operand(a, kb, ib, lds, T, tid, i);
      u64 r;
      switch (op) {  // uniform across the wave: the code stream is the same for every point
        case LCP2_OP_ADD: r = gl_add(x, y); break;
        case LCP2_OP_SUB: r = gl_sub(x, y); break;
        case LCP2_OP_MUL: r = gl_mul(x, y); break;
        case LCP2_OP_XOR: { const u64 xy = gl_mul(x, y); r = gl_sub(gl_sub(gl_add(x, y), xy), xy); break; }
        case LCP2_OP_DBLADD: r = gl_add(gl_add(x, x), y); break;
        default: r = gl_add(lds[dst * T + tid], gl_mul(x, y)); break;  // LCP2_OP_MULADD
      }
      lds[dst * T + tid] = r;
    }
    u64 s = a.consts[(u64)G.selector_index * a.stride + i];
    u64 f = 1;
    for (u32 j = G.group_start; j < G.group_end; j++)
      if (j != G.selector_value) f = gl_mul(f, gl_sub((u64)j, s));
    if (a.num_selectors > 1) f = gl_mul(f, gl_sub(0xFFFFFFFFull, s));
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
      if (c < CH) res[c] = gl_add(res[c], gl_mul(f, acc[c]));
  }

  // ---- permutation argument terms, folded in front of the gate constraints:
  //   terms = [ L0 (Z_c - 1) ]_c ++ [ prev * prod num - next * prod den ]_{c,k} ; res <- sum_t alpha^t terms_t + alpha^nt * gates
  const u32 lgN = a.lgN;
  const u64 jnat = bitrev32((u32)ig, lgN);
  const u64 x = two_level(a.points, jnat);  // 7 * w_N^bitrev(ig)
  const u64 inext = bitrev32((u32)((jnat + (1u << a.rate_bits)) & (a.N - 1)), lgN) - a.leaf0;  // same coset = same leaf block
  const u32 npp = a.nchunks - 1;
  // Horner from the last term down to the first, for every alpha
  for (int c2 = (int)CH - 1; c2 >= 0; c2--) {
    const u64 beta = a.betas[c2], gamma = a.gammas[c2];
    const u64 bx = gl_mul(beta, x);
    for (int k = (int)a.nchunks - 1; k >= 0; k--) {
      u64 pn = 1, pd = 1;
      for (u32 j = k * a.chunk; j < a.num_routed && j < (k + 1) * a.chunk; j++) {
        u64 wg = gl_add(a.wires[(u64)j * a.stride + i], gamma);
        pn = gl_mul(pn, gl_add(wg, gl_mul(bx, a.k_is[j])));
        pd = gl_mul(pd, gl_add(wg, gl_mul(beta, a.consts[(u64)(a.num_constants + j) * a.stride + i])));
      }
      u64 prev = k == 0 ? a.zs[(u64)c2 * a.stride + i] : a.zs[((u64)CH + (u64)c2 * npp + (k - 1)) * a.stride + i];
      u64 next = (u32)k < npp ? a.zs[((u64)CH + (u64)c2 * npp + k) * a.stride + i] : a.zs[(u64)c2 * a.stride + inext];
      u64 term = gl_sub(gl_mul(prev, pn), gl_mul(next, pd));
#pragma unroll
      for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
        if (c < CH) res[c] = gl_add(gl_mul(res[c], a.alphas[c]), term);
    }
  }
  const u64 l0 = a.l0[ig];
  for (int c2 = (int)CH - 1; c2 >= 0; c2--) {
    u64 term = gl_mul(l0, gl_sub(a.zs[(u64)c2 * a.stride + i], 1));
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
      if (c < CH) res[c] = gl_add(gl_mul(res[c], a.alphas[c]), term);
  }
  const u64 zhi = a.zh_inv[ig >> (lgN - a.rate_bits)];
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
    if (c < CH) a.out[(u64)c * a.N + ig] = gl_mul(res[c], zhi);
}
void launch_quotient(hipStream_t s, const QuotientArgs &a) {
  size_t lds = (size_t)a.num_regs * QUOTIENT_THREADS * sizeof(u64);
  hipLaunchKernelGGL(k_quotient, dim3((unsigned)((a.count + QUOTIENT_THREADS - 1) / QUOTIENT_THREADS)), dim3(QUOTIENT_THREADS), lds, s, a);
}

// ------------------------------------------------------------------ K7a: evaluate coefficient polynomials at an extension point
// grid (chunks, polys); a chunk is EVAL_CHUNK coefficients; thread t owns coefficients t, t+256, ...
// partial[poly][chunk] = z^(chunk*EVAL_CHUNK) * sum_t z^t * Horner_m(c[t + 256 m]; z^256)
__global__ __launch_bounds__(256) void k_eval_polys(EvalArgs a) {
  __shared__ u64 sh0[256], sh1[256];
  const u32 t = threadIdx.x, chunk = blockIdx.x, poly = blockIdx.y;
  const u64 *c = a.coeffs + (u64)poly * a.col_stride + (u64)chunk * a.chunk_len;
  gl2 acc = gl2_make(0, 0);
  const gl2 z256 = gl2_make(a.zstep[0], a.zstep[1]);
  for (int m = (int)a.items - 1; m >= 0; m--) {
    u32 idx = t + 256u * (u32)m;
    acc = gl2_mul(acc, z256);
    if (idx < a.chunk_len) acc = gl2_add_base(acc, c[idx]);
  }
  acc = gl2_mul(acc, gl2_make(a.zpow_t[2 * t], a.zpow_t[2 * t + 1]));
  sh0[t] = acc.c0; sh1[t] = acc.c1;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < (u32)off) { sh0[t] = gl_add(sh0[t], sh0[t + off]); sh1[t] = gl_add(sh1[t], sh1[t + off]); }
    __syncthreads();
  }
  if (t == 0) {
    gl2 r = gl2_mul(gl2_make(sh0[0], sh1[0]), gl2_make(a.zpow_chunk[2 * chunk], a.zpow_chunk[2 * chunk + 1]));
    a.partial[2 * ((u64)poly * a.nchunks + chunk)] = r.c0;
    a.partial[2 * ((u64)poly * a.nchunks + chunk) + 1] = r.c1;
  }
}
__global__ void k_eval_reduce(const u64 *__restrict__ partial, u32 nchunks, u32 npolys, u64 *__restrict__ out) {
  u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npolys) return;
  gl2 s = gl2_make(0, 0);
  for (u32 k = 0; k < nchunks; k++) s = gl2_add(s, gl2_make(partial[2 * ((u64)p * nchunks + k)], partial[2 * ((u64)p * nchunks + k) + 1]));
  out[2 * p] = s.c0; out[2 * p + 1] = s.c1;
}
void launch_eval_polys(hipStream_t s, const EvalArgs &a, u32 npolys, u64 *out) {
  hipLaunchKernelGGL(k_eval_polys, dim3(a.nchunks, npolys), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_eval_reduce, dim3((npolys + 63) / 64), dim3(64), 0, s, a.partial, a.nchunks, npolys, out);
}

// ------------------------------------------------------------------ K7b: composition polynomial and division by (X - z)
// t0_i = (sum_j alpha^j f_j[i]) * zeta^i over every committed polynomial, t1_i = (sum_{j<CH} alpha^j Z_j[i]) * (g zeta)^i
// planes: [t0.c0, t0.c1, t1.c0, t1.c1][n]
__global__ __launch_bounds__(256) void k_compose(ComposeArgs a) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  gl2 f0 = gl2_make(0, 0);
  u32 j = 0;
  for (u32 o = 0; o < 4; o++) {
    const u64 *cf = a.coeffs[o];
    for (u32 c = 0; c < a.ncols[o]; c++, j++) {
      u64 v = cf[(u64)c * a.n + i];
      f0.c0 = gl_add(f0.c0, gl_mul(a.alpha_pows[2 * j], v));
      f0.c1 = gl_add(f0.c1, gl_mul(a.alpha_pows[2 * j + 1], v));
    }
  }
  gl2 f1 = gl2_make(0, 0);
  for (u32 c = 0; c < a.num_challenges; c++) {
    u64 v = a.coeffs[2][(u64)c * a.n + i];
    f1.c0 = gl_add(f1.c0, gl_mul(a.alpha_pows[2 * c], v));
    f1.c1 = gl_add(f1.c1, gl_mul(a.alpha_pows[2 * c + 1], v));
  }
  gl2 z0 = gl2_mul(gl2_make(a.z0_lo[2 * (i & a.zmask)], a.z0_lo[2 * (i & a.zmask) + 1]), gl2_make(a.z0_hi[2 * (i >> a.zh)], a.z0_hi[2 * (i >> a.zh) + 1]));
  gl2 z1 = gl2_mul(gl2_make(a.z1_lo[2 * (i & a.zmask)], a.z1_lo[2 * (i & a.zmask) + 1]), gl2_make(a.z1_hi[2 * (i >> a.zh)], a.z1_hi[2 * (i >> a.zh) + 1]));
  gl2 t0 = gl2_mul(f0, z0), t1 = gl2_mul(f1, z1);
  a.planes[i] = t0.c0; a.planes[a.n + i] = t0.c1; a.planes[2 * a.n + i] = t1.c0; a.planes[3 * a.n + i] = t1.c1;
}
// planes now hold the exclusive suffix sums S_{i+1}; final_i = alpha^CH * zeta^-(i+1) S0_{i+1} + (g zeta)^-(i+1) S1_{i+1}
__global__ __launch_bounds__(256) void k_divide_finalize(ComposeArgs a, u64 *__restrict__ out0, u64 *__restrict__ out1) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  u64 e = i + 1;  // <= n; the inverse tables cover n+1 exponents through the hi table
  gl2 zi0 = gl2_mul(gl2_make(a.zi0_lo[2 * (e & a.zmask)], a.zi0_lo[2 * (e & a.zmask) + 1]), gl2_make(a.zi0_hi[2 * (e >> a.zh)], a.zi0_hi[2 * (e >> a.zh) + 1]));
  gl2 zi1 = gl2_mul(gl2_make(a.zi1_lo[2 * (e & a.zmask)], a.zi1_lo[2 * (e & a.zmask) + 1]), gl2_make(a.zi1_hi[2 * (e >> a.zh)], a.zi1_hi[2 * (e >> a.zh) + 1]));
  gl2 s0 = gl2_make(a.planes[i], a.planes[a.n + i]), s1 = gl2_make(a.planes[2 * a.n + i], a.planes[3 * a.n + i]);
  gl2 q0 = gl2_mul(gl2_mul(s0, zi0), gl2_make(a.alpha_shift[0], a.alpha_shift[1]));
  gl2 q1 = gl2_mul(s1, zi1);
  gl2 r = gl2_add(q0, q1);
  out0[i] = r.c0; out1[i] = r.c1;
}
void launch_compose(hipStream_t s, const ComposeArgs &a) {
  hipLaunchKernelGGL(k_compose, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
}
void launch_divide_finalize(hipStream_t s, const ComposeArgs &a, u64 *out0, u64 *out1) {
  hipLaunchKernelGGL(k_divide_finalize, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a, out0, out1);
}

// ------------------------------------------------------------------ K8: FRI fold  new[k] = sum_j beta^j c[arity k + j]
__global__ __launch_bounds__(256) void k_fri_fold(const u64 *__restrict__ c0, const u64 *__restrict__ c1, u64 *__restrict__ o0,
                                                   u64 *__restrict__ o1, u64 nout, u32 arity, u64 b0, u64 b1) {
  u64 k = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nout) return;
  const gl2 beta = gl2_make(b0, b1);
  gl2 acc = gl2_make(0, 0);
  for (int j = (int)arity - 1; j >= 0; j--) acc = gl2_add(gl2_mul(acc, beta), gl2_make(c0[k * arity + j], c1[k * arity + j]));
  o0[k] = acc.c0; o1[k] = acc.c1;
}
void launch_fri_fold(hipStream_t s, const u64 *c0, const u64 *c1, u64 *o0, u64 *o1, u64 nout, u32 arity, u64 b0, u64 b1) {
  hipLaunchKernelGGL(k_fri_fold, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, s, c0, c1, o0, o1, nout, arity, b0, b1);
}
// FRI query: out[q][2j + e] = plane_e[leaf * arity + j]
__global__ void k_gather_ext_leaves(const u64 *__restrict__ p0, const u64 *__restrict__ p1, u32 arity, const u64 *__restrict__ leaf_idx,
                                    u32 k, u64 *__restrict__ out) {
  u32 q = blockIdx.x;
  if (q >= k) return;
  u64 leaf = leaf_idx[q];
  for (u32 t = threadIdx.x; t < 2 * arity; t += blockDim.x) out[(u64)q * 2 * arity + t] = (t & 1 ? p1 : p0)[leaf * arity + (t >> 1)];
}
void launch_gather_ext_leaves(hipStream_t s, const u64 *p0, const u64 *p1, u32 arity, const u64 *leaf_idx, u32 k, u64 *out) {
  if (!k) return;
  hipLaunchKernelGGL(k_gather_ext_leaves, dim3(k), dim3(64), 0, s, p0, p1, arity, leaf_idx, k, out);
}

// ------------------------------------------------------------------ K9: proof of work (minimum witness)
__global__ __launch_bounds__(256) void k_pow_search(PowArgs a) {
  u64 w = a.start + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = a.state[i];
#pragma unroll
  for (int i = 0; i < 12; i++)
    if (i == (int)a.pos) s[i] = w;
  pos_permute(s, a.rc);
  if (w < GL_P && (s[7] >> (64 - a.bits)) == 0) atomicMin((unsigned long long *)a.result, (unsigned long long)w);
}
void launch_pow_search(hipStream_t s, const PowArgs &a, u64 count) {
  hipLaunchKernelGGL(k_pow_search, dim3((unsigned)(count / 256)), dim3(256), 0, s, a);
}

__global__ void k_fill(u64 *p, u64 n, u64 v) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void launch_fill(hipStream_t s, u64 *p, u64 n, u64 v) {
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}

}  // namespace lcp2
