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
#include "kernels_gates.hpp"

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
  u64 x = two_level(a.subgroup, a.row0 + row);
  u64 bx = gl_mul(beta, x);
  u64 pn[PERM_MAX_CHUNKS], pd[PERM_MAX_CHUNKS];
#pragma unroll
  for (u32 k = 0; k < PERM_MAX_CHUNKS; k++) {
    pn[k] = 1; pd[k] = 1;
    if (k < a.nchunks) {
      for (u32 j = k * a.chunk; j < a.num_routed && j < (k + 1) * a.chunk; j++) {
        u64 w = gl_canon(a.wires[(u64)j * a.wires_stride + row]);
        u64 wg = gl_add(w, gamma);
        // lazy through the products (any u64 congruent to the element); the batch inversion below multiplies canonically
        pn[k] = gl_mul_nc(pn[k], gl_add_nc(gl_mul_nc(bx, a.k_is[j]), wg));
        pd[k] = gl_mul_nc(pd[k], gl_add_nc(gl_mul_nc(beta, a.sigmas[(u64)j * a.sigma_stride + row]), wg));
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
  if (a.prefix) {
    acc = gl_mul(acc, a.prefix[ch]);
    a.zs_out[(u64)ch * a.n + row] = acc;
  }
  u32 npp = a.nchunks - 1;
  for (u32 k = 0; k < npp; k++) {
    acc = gl_mul(acc, a.chunk_q[((u64)ch * a.nchunks + k) * a.n + row]);
    a.zs_out[((u64)a.num_challenges + (u64)ch * npp + k) * a.n + row] = acc;
  }
}
// Coset-sharded proof: the quotient chunks out of the per-coset interpolants.  in[c][b * n + l] is coefficient l of the polynomial
// of degree < n that agrees with challenge c's quotient on leaf block b (the coset of shift s_b); the quotient is
// sum_k x^(k n) Q_k(x) and x^n = s_b^n on that coset, so the interpolants are a size-R transform of the chunks Q_k, coefficient by
// coefficient, and out[c][k * n + l] = sum_b m[k][b] in[c][b * n + l] with the inverse matrix m (prover.hip).
__global__ __launch_bounds__(256) void k_quotient_combine(const u64 *__restrict__ in, u64 *__restrict__ out, const u64 *__restrict__ m, u64 n, u32 R, u64 plane) {
  const u64 l = (u64)blockIdx.x * 256 + threadIdx.x;
  if (l >= n) return;
  const u64 base = (u64)blockIdx.y * plane + l;
  u64 r[8];
#pragma unroll
  for (u32 b = 0; b < 8; b++) r[b] = b < R ? in[base + (u64)b * n] : 0;
  for (u32 k = 0; k < R; k++) {
    u64 acc = 0;
#pragma unroll
    for (u32 b = 0; b < 8; b++)
      if (b < R) acc = gl_add(acc, gl_mul(m[k * R + b], r[b]));
    out[base + (u64)k * n] = acc;
  }
}
void launch_quotient_combine(hipStream_t s, const u64 *in, u64 *out, const u64 *m, u64 n, u32 R, u64 plane, u32 num_challenges) {
  hipLaunchKernelGGL(k_quotient_combine, dim3((unsigned)((n + 255) / 256), num_challenges), dim3(256), 0, s, in, out, m, n, R, plane);
}

void launch_perm_chunks(hipStream_t s, const PermArgs &a) {
  hipLaunchKernelGGL(k_perm_chunks, dim3((unsigned)((a.n + 255) / 256), a.num_challenges), dim3(256), 0, s, a);
}
void launch_perm_finalize(hipStream_t s, const PermArgs &a) {
  hipLaunchKernelGGL(k_perm_finalize, dim3((unsigned)((a.n + 255) / 256), a.num_challenges), dim3(256), 0, s, a);
}

// Host: the staged form of the programs.  Instructions are scanned in order; when one needs a WIRE / CONST operand that is not
// in the current window, a new window opens: the distinct column operands of the instructions ahead are collected (in order of
// first use) until QUOTIENT_STAGE of them are found, one LDG fetches them, and operands are rewritten to their slots.
void stage_gate_programs(const std::vector<uint32_t> &code, std::vector<GateDev> &gates, u32 num_wires, u32 num_selectors,
                         std::vector<uint32_t> &out, std::vector<uint32_t> &lists) {
  out.clear(); lists.clear();
  auto column_of = [&](u32 kind, u32 idx) { return kind == 1 ? idx : num_wires + num_selectors + idx; };
  auto nsrc_of = [](u32 op) { return (op == LCP2_OP_EMIT || op == LCP2_OP_EMITBOOL || op == LCP2_OP_SBOX) ? 1u : op == LCP2_OP_PMDS ? 0u : 2u; };
  for (GateDev &G : gates) {
    const u32 first = G.code_offset, last = G.code_offset + G.code_len, new_first = (u32)out.size() / 2;
    std::vector<u32> window;  // columns staged by the last LDG
    for (u32 pc = first; pc < last; pc++) {
      u32 w0 = code[2 * pc], w1 = code[2 * pc + 1];
      const u32 op = w0 & 0xF;
      u32 kk[2] = {(w0 >> 16) & 0xF, (w0 >> 20) & 0xF}, ii[2] = {w1 & 0xFFFF, w1 >> 16};
      const u32 nsrc = nsrc_of(op);
      auto slot_of = [&](u32 col) { for (u32 s = 0; s < window.size(); s++) if (window[s] == col) return (int)s; return -1; };
      bool missing = false;
      for (u32 k = 0; k < nsrc; k++)
        if ((kk[k] == 1 || kk[k] == 2) && slot_of(column_of(kk[k], ii[k])) < 0) missing = true;
      if (missing) {  // open a new window from here
        window.clear();
        for (u32 q = pc; q < last && window.size() < QUOTIENT_STAGE; q++) {
          const u32 v0 = code[2 * q], v1 = code[2 * q + 1], o = v0 & 0xF;
          const u32 k2[2] = {(v0 >> 16) & 0xF, (v0 >> 20) & 0xF}, i2[2] = {v1 & 0xFFFF, v1 >> 16};
          std::vector<u32> need;
          for (u32 k = 0; k < nsrc_of(o); k++)
            if (k2[k] == 1 || k2[k] == 2) {
              const u32 col = column_of(k2[k], i2[k]);
              bool have = false;
              for (u32 c : window) have = have || c == col;
              for (u32 c : need) have = have || c == col;
              if (!have) need.push_back(col);
            }
          if (window.size() + need.size() > QUOTIENT_STAGE) break;  // an instruction's operands never straddle two windows
          window.insert(window.end(), need.begin(), need.end());
        }
        out.push_back(QOP_LDG | (u32)window.size() << 8);
        out.push_back((u32)lists.size());
        lists.insert(lists.end(), window.begin(), window.end());
      }
      for (u32 k = 0; k < nsrc; k++)
        if (kk[k] == 1 || kk[k] == 2) { ii[k] = (u32)slot_of(column_of(kk[k], ii[k])); kk[k] = QKIND_STAGE; }
      if (op != LCP2_OP_PMDS) {
        w0 = (w0 & 0xFFFF) | kk[0] << 16 | kk[1] << 20;
        w1 = (nsrc >= 1 ? ii[0] : (w1 & 0xFFFF)) | (nsrc >= 2 ? ii[1] : (w1 >> 16)) << 16;
      }
      out.push_back(w0); out.push_back(w1);
    }
    G.code_offset = new_first;
    G.code_len = (u32)out.size() / 2 - new_first;
  }
  lists.resize(lists.size() + QUOTIENT_STAGE, 0);  // an LDG always reads entry 0 of its list: keep the tail readable
}

// ---- native PoseidonGate (LCP2_GATE_NATIVE_POSEIDON): plonky2 gates/poseidon.rs::eval_unfiltered_base with the state in
// VGPRs (32-bit halves, lazily reduced, exactly the permutation of the hash kernels) instead of LDS registers and one
// interpreted instruction at a time.  Wires: input 0..12, output 12..24, swap 24, delta 25..29, S-box inputs of full rounds
// 1..3 at 29 + 12 (r - 1) + i, of the partial rounds at 65 + r, of full rounds 4..7 at 87 + 12 r + i.  The constraints come out
// first to last; acc is the Horner chain with 1 / alpha (rescaled by the caller), as for every EMIT_FORWARD gate.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void q_poseidon_native(const QuotientArgs &a, u64 i, u64 *lds, u32 T, u32 tid, QEmit &emit) {
  const_as<u64> rc = konst(a.rc);
  // the constraints come first to last: constraint j has the weight alpha^j (QTerms; no 1 / alpha, no rescaling)
  QEmit &E = emit;
  E.begin_terms();
  auto emit_term = [&](u64 x) { E.term(a, E.emitted++, x); };
  const u64 *W = a.wires + i;
  const u64 st = a.stride;
  u64 in[12], dl[4], swap;
#pragma unroll
  for (int j = 0; j < 12; j++) in[j] = W[(u64)j * st];
  swap = W[24 * st];
#pragma unroll
  for (int j = 0; j < 4; j++) dl[j] = W[(u64)(25 + j) * st];
  emit_term(gl_mul_nc(swap, gl_sub(swap, 1)));
#pragma unroll
  for (int j = 0; j < 4; j++) emit_term(gl_sub_nc(gl_mul_nc(swap, gl_sub(in[j + 4], in[j])), dl[j]));
  u32 lo[12], hi[12];
#pragma unroll
  for (int j = 0; j < 12; j++) {
    u64 v = j < 4 ? gl_add(in[j], dl[j]) : j < 8 ? gl_sub(in[j], dl[j - 4]) : in[j];
    v = gl_add_nc(v, rc[j]);
    lo[j] = (u32)v; hi[j] = (u32)(v >> 32);
  }
  auto constrain12 = [&](u32 first_wire) {  // state - sbox_in for the 12 lanes, state <- sbox_in
    u64 w[12];
#pragma unroll
    for (int j = 0; j < 12; j++) w[j] = W[(u64)(first_wire + j) * st];
#pragma unroll
    for (int j = 0; j < 12; j++) {
      emit_term(gl_sub_nc(((u64)hi[j] << 32) | lo[j], w[j]));
      lo[j] = (u32)w[j]; hi[j] = (u32)(w[j] >> 32);
    }
  };
  u32 round = 0;
#pragma unroll 1
  for (u32 r = 0; r < POS_FULL_HALF; r++, round++) {
    if (r) constrain12(29 + 12 * (r - 1));
#pragma unroll
    for (int j = 0; j < 12; j++) pos_sbox_h(lo[j], hi[j]);
    u64 nxt[12];
#pragma unroll
    for (int j = 0; j < 12; j++) nxt[j] = rc[(round + 1) * 12 + j];
    pos_mds_h(lo, hi, nxt);
  }
  // partial rounds, three at a time as in the hash kernels (poseidon.hpp pos_partial3_core): element 0 after every round is
  // emitted against the gate's S-box wire and the round continues from the wire.  The 22 S-box wires are fetched in groups of
  // QUOTIENT_STAGE through the staging slots (LDS; this kernel has no interpreter registers).
  static_assert(QUOTIENT_STAGE % POS_GROUP == 0, "a group of partial rounds must not straddle two staging batches");
  auto stage_sbox_wires = [&](u32 r) {
    u64 pw[QUOTIENT_STAGE];
#pragma unroll
    for (u32 j = 0; j < QUOTIENT_STAGE; j++) pw[j] = W[(u64)(65 + min(r + j, (u32)POS_PARTIAL - 1)) * st];
#pragma unroll
    for (u32 j = 0; j < QUOTIENT_STAGE; j++) lds[j * T + tid] = pw[j];
  };
  auto constrain0 = [&](u32 r, u32 &ul, u32 &uh) {  // element 0 - S-box wire of partial round r; element 0 <- the wire
    const u64 w = lds[(r % QUOTIENT_STAGE) * T + tid];
    emit_term(gl_sub_nc(((u64)uh << 32) | ul, w));
    ul = (u32)w; uh = (u32)(w >> 32);
  };
  u32 r = 0;
#pragma unroll 1
  for (u32 g = 0; g < POS_GROUPS; g++, r += POS_GROUP, round += POS_GROUP) {
    if (r % QUOTIENT_STAGE == 0) stage_sbox_wires(r);
    constrain0(r, lo[0], hi[0]);
    pos_partial3_core(lo, hi, &rc[POS_ROUNDS * POS_W + POS_GROUP_CONSTS * g], [&](int i, u32 &ul, u32 &uh) {
      constrain0(r + i, ul, uh);
      pos_sbox_h(ul, uh);
    });
  }
#pragma unroll 1
  for (; r < POS_PARTIAL; r++, round++) {  // the round the groups leave over
    if (r % QUOTIENT_STAGE == 0) stage_sbox_wires(r);
    constrain0(r, lo[0], hi[0]);
    pos_sbox_h(lo[0], hi[0]);
    u64 nxt[12];
#pragma unroll
    for (int j = 0; j < 12; j++) nxt[j] = rc[(round + 1) * 12 + j];
    pos_mds_h(lo, hi, nxt);
  }
#pragma unroll 1
  for (u32 r = 0; r < POS_FULL_HALF; r++, round++) {
    constrain12(87 + 12 * r);
#pragma unroll
    for (int j = 0; j < 12; j++) pos_sbox_h(lo[j], hi[j]);
    u64 nxt[12];
#pragma unroll
    for (int j = 0; j < 12; j++) nxt[j] = round + 1 < POS_ROUNDS ? rc[(round + 1) * 12 + j] : 0;
    pos_mds_h(lo, hi, nxt);
  }
  u64 out[12];
#pragma unroll
  for (int j = 0; j < 12; j++) out[j] = W[(u64)(12 + j) * st];
#pragma unroll
  for (int j = 0; j < 12; j++) emit_term(gl_sub_nc(((u64)hi[j] << 32) | lo[j], out[j]));
  E.finish_terms();
}
#endif

// ---- native ArithmeticGate { num_ops } (gates/arithmetic_base.rs): output - (c0 * m0 * m1 + c1 * addend) per operation, wires
// 4k .. 4k+3.  Constraints are folded last to first (plain Horner with alpha), four operations = 16 wire loads per batch.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void q_arithmetic_native(const QuotientArgs &a, u64 i, u32 num_ops, QEmit &emit) {
  const u64 *W = a.wires + i;
  const u64 st = a.stride;
  const u64 c0 = a.consts[(u64)a.num_selectors * st + i], c1 = a.consts[(u64)(a.num_selectors + 1) * st + i];
  emit.begin_terms();  // operation k is constraint k: weight alpha^k
  for (int top = (int)num_ops; top > 0; top -= 4) {  // operations [top - 4, top), clamped at 0
    const int first = top >= 4 ? top - 4 : 0;
    u64 w[16];
#pragma unroll
    for (int j = 0; j < 16; j++) w[j] = W[(u64)min(4 * first + j, 4 * (int)num_ops - 1) * st];  // a short last batch re-reads its last wire
#pragma unroll
    for (int k = 3; k >= 0; k--) {
      if (first + k < top) {
        const u64 comp = gl_add(gl_mul(gl_mul(w[4 * k], w[4 * k + 1]), c0), gl_mul(w[4 * k + 2], c1));
        emit.term(a, (u32)(first + k), gl_sub(w[4 * k + 3], comp));
      }
    }
  }
  emit.finish_terms();
}
// ---- native BaseSumGate<2> { num_limbs } (gates/base_sum.rs): constraints [sum_i 2^i limb_i - wire_0, limb_i^2 - limb_i ...],
// folded last to first: the limb constraints from the top limb down (the recomposition is the same walk), then the sum.
__device__ __forceinline__ void q_base_sum2_native(const QuotientArgs &a, u64 i, u32 num_limbs, QEmit &emit) {
  const u64 *W = a.wires + i;
  const u64 st = a.stride;
  u64 sum = 0;
  emit.begin_terms();  // constraint 0 is the sum, constraint 1 + l the booleanity of limb l
  for (int top = (int)num_limbs; top > 0; top -= 16) {  // limbs [top - 16, top) = wires [top - 15, top]
    u64 w[16];
#pragma unroll
    for (int j = 0; j < 16; j++) { const int l = top - 1 - j; w[j] = W[(u64)(1 + (l > 0 ? l : 0)) * st]; }
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (top - 1 - j >= 0) {
        emit.term(a, (u32)(top - j), gl_mul_nc(w[j], gl_sub(w[j], 1)));  // limb top - 1 - j
        sum = gl_add(gl_add(sum, sum), w[j]);
      }
    }
  }
  emit.term(a, 0, gl_sub(sum, W[0]));
  emit.finish_terms();
}
#endif

// val[c] <- filter_g(point) * sum_i alpha_c^i constraint_{g,i}(point) for gate g at the point whose operands sit at index i.
// NATIVE = 0 interprets the gate's program, LCP2_GATE_NATIVE_* runs the native evaluator of that plonky2 gate (the generated
// straight-line evaluators have kernels of their own: kernels_gates.hpp).
template <u32 NATIVE>
__device__ __forceinline__ void q_gate_value(const QuotientArgs &a, u32 g, const GateDev &G, u64 i, u64 *lds, u32 T, u32 tid, u64 val[QUOTIENT_MAX_CH]) {
#if defined(__HIP_DEVICE_COMPILE__)
  QEmit emit;
  q_emit_begin(a, G, emit);
  if (NATIVE == LCP2_GATE_NATIVE_POSEIDON) q_poseidon_native(a, i, lds, T, tid, emit);
  else if (NATIVE == LCP2_GATE_NATIVE_ARITHMETIC) q_arithmetic_native(a, i, G.num_constraints, emit);
  else if (NATIVE == LCP2_GATE_NATIVE_BASE_SUM2) q_base_sum2_native(a, i, G.num_constraints - 1, emit);
  else q_interpret(a, G, i, lds, T, tid, emit);
  q_gate_finish(a, g, G, i, emit, val);
#endif
}


// K6 is one launch per gate type plus the permutation pass: every kernel carries only the registers its gate needs (the native
// PoseidonGate wants ~120 VGPRs, the permutation pass ~90, an interpreted gate ~80), so none drags the others' occupancy down,
// and an interpreted gate's LDS registers are not allocated beside a native one.  A gate kernel adds filter * constraints into
// out[c][point] (the first one of a proof stores); the extra traffic is 32 bytes per point and launch, ~1.5 % of what K6 reads.
// CHECK = true is the same evaluation over the rows of H (lcp2_prove's LCP2_E_UNSAT): on a row only its own gate has a
// non-zero filter, so a wave skips a gate that none of its rows holds, and a non-zero value is a violated constraint.
template <u32 NATIVE, bool CHECK>
__global__ __launch_bounds__(QUOTIENT_THREADS, 2) void k_q_gate(QuotientArgs a, u32 g, u32 accumulate, unsigned long long *flag) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 T = QUOTIENT_THREADS, tid = threadIdx.x;
  const u64 i0 = (u64)blockIdx.x * T + tid;
  if (!CHECK && i0 >= a.count) return;            // no barrier is used below
  const u64 i = i0 < a.count ? i0 : a.count - 1;  // CHECK: the tail re-checks the last row so that every lane votes
  const GateDev G = q_load_gate(a, g);
  if (CHECK) {
    const u64 sel = a.consts[(u64)G.selector_index * a.stride + i];
    if (!__any(sel == G.selector_value)) return;
  }
  u64 val[QUOTIENT_MAX_CH];
  q_gate_value<NATIVE>(a, g, G, i, lds, T, tid, val);
  if (CHECK) {
    bool bad = false;
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
      if (c < a.num_challenges && val[c] != 0) bad = true;
    if (bad) atomicMin(flag, (unsigned long long)i + 1);
  } else {
    const u64 ig = a.leaf0 + i;
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
      if (c < a.num_challenges) a.out[(u64)c * a.N + ig] = accumulate ? gl_add(a.out[(u64)c * a.N + ig], val[c]) : val[c];
  }
}

// x * 7 for any u64 x, lazy result: the 67-bit product from two multiply-adds, folded with two more (its high word times 2^64 mod p,
// and that sum's carry): 5 instructions of the 4.3-cycle kind against 12 for a general multiply
__device__ __forceinline__ u64 q_mul7_nc(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const u64 p0 = (u64)(u32)x * 7u, p1 = (u64)(u32)(x >> 32) * 7u + (p0 >> 32);
  u64 lo = (p1 << 32) | (u32)p0;
  const u32 hi = (u32)(p1 >> 32);
  u32 c;
  asm("v_mad_u64_u32 %0, vcc, %2, -1, %0\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %1, 0, 1, vcc\n\t"
      "v_mad_u64_u32 %0, vcc, %1, -1, %0"
      : "+v"(lo), "=&v"(c) : "v"(hi) : "vcc");
  return lo;
#else
  return gl_mul_u32_nc(x, 7u);
#endif
}

// permutation argument + division by Z_H: out[c][point] holds the sum of the gate terms on entry, the quotient value on exit
__global__ __launch_bounds__(QUOTIENT_THREADS, 2) void k_q_perm(QuotientArgs a, u32 have_gates) {
  const u32 T = QUOTIENT_THREADS, tid = threadIdx.x;
  const u64 i = (u64)blockIdx.x * T + tid;  // local storage (leaf) index; global index = a.leaf0 + i
  if (i >= a.count) return;
  const u64 ig = a.leaf0 + i;
  const u32 CH = a.num_challenges;
  u64 res[QUOTIENT_MAX_CH];
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) res[c] = (c < CH && have_gates) ? a.out[(u64)c * a.N + ig] : 0;

  // ---- permutation argument terms, folded in front of the gate constraints:
  //   terms = [ L0 (Z_c - 1) ]_c ++ [ prev * prod num - next * prod den ]_{c,k} ;  out = sum_t alpha^t terms_t + alpha^nt * gates
  // The chunks are walked once for all challenges: the 8 wires and 8 sigmas of a batch and the next partial products are 18
  // loads issued back to back (one HBM round trip per batch instead of one per column) and every column is read once.  The
  // terms of challenge c2 then arrive first to last, so their block sum is a Horner chain with 1 / alpha, weighted afterwards
  // by the power of alpha at which the block starts (alpha_pow: host table; alpha = 0 leaves term 0 alone, handled below).
  const u32 lgN = a.lgN;
  const u64 jnat = bitrev32((u32)ig, lgN);
  const u64 x = two_level(a.points, jnat);  // 7 * w_N^bitrev(ig)
  const u64 inext = bitrev32((u32)((jnat + (1u << a.rate_bits)) & (a.N - 1)), lgN) - a.leaf0;  // same coset = same leaf block
  const u32 npp = a.nchunks - 1;
  u64 beta[QUOTIENT_MAX_CH], gamma[QUOTIENT_MAX_CH], bx[QUOTIENT_MAX_CH], prev[QUOTIENT_MAX_CH], z0[QUOTIENT_MAX_CH];
  u64 bxk[QUOTIENT_MAX_CH];  // beta x k_j of the next wire when k_j = 7^j (a.kis_pow7): lazy, carried with q_mul7_nc
  u64 hh[QUOTIENT_MAX_CH][QUOTIENT_MAX_CH];  // [c2][alpha challenge c]
  u64 ainv[QUOTIENT_MAX_CH];
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) {
    beta[c] = c < CH ? konst(a.betas)[c] : 0; gamma[c] = c < CH ? konst(a.gammas)[c] : 0;
    ainv[c] = c < CH ? konst(a.alpha_inv)[c] : 0;
    bx[c] = gl_mul(beta[c], x);
    bxk[c] = bx[c];
    z0[c] = c < CH ? a.zs[(u64)c * a.stride + i] : 0;
    prev[c] = z0[c];
#pragma unroll
    for (u32 d = 0; d < QUOTIENT_MAX_CH; d++) hh[c][d] = 0;
  }
  for (u32 k = 0; k < a.nchunks; k++) {
    u64 pn[QUOTIENT_MAX_CH], pd[QUOTIENT_MAX_CH], nx[QUOTIENT_MAX_CH];
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) {
      pn[c] = 1; pd[c] = 1;
      nx[c] = c < CH ? (k < npp ? a.zs[((u64)CH + (u64)c * npp + k) * a.stride + i] : a.zs[(u64)c * a.stride + inext]) : 0;
    }
    const u32 jend = min((k + 1) * a.chunk, a.num_routed);
    for (u32 j0 = k * a.chunk; j0 < jend; j0 += 8) {
      u64 w[8], sg[8];
#pragma unroll
      for (u32 jj = 0; jj < 8; jj++) {  // lanes past the end of the chunk repeat its last column
        const u32 j = min(j0 + jj, jend - 1);
        w[jj] = a.wires[(u64)j * a.stride + i];
        sg[jj] = a.consts[(u64)(a.num_constants + j) * a.stride + i];
      }
#pragma unroll
      for (u32 jj = 0; jj < 8; jj++) {
        if (j0 + jj < jend) {
          const u64 kj = konst(a.k_is)[j0 + jj];
#pragma unroll
          for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
            if (c < CH) {
              // lazy values (any u64 congruent to the element) through the products: gl_mul_nc takes them, and the chunk
              // products only meet canonical arithmetic in the gl_mul of the term below
              const u64 wg = gl_add(w[jj], gamma[c]);
              const u64 bk = a.kis_pow7 ? bxk[c] : gl_mul_nc(bx[c], kj);
              if (a.kis_pow7) bxk[c] = q_mul7_nc(bxk[c]);
              pn[c] = gl_mul_nc(pn[c], gl_add_nc(bk, wg));
              pd[c] = gl_mul_nc(pd[c], gl_add_nc(gl_mul_nc(beta[c], sg[jj]), wg));
            }
        }
      }
    }
#pragma unroll
    for (u32 c2 = 0; c2 < QUOTIENT_MAX_CH; c2++)
      if (c2 < CH) {
        const u64 term = gl_sub(gl_mul(prev[c2], pn[c2]), gl_mul(nx[c2], pd[c2]));
        prev[c2] = nx[c2];
#pragma unroll
        for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
          if (c < CH) hh[c2][c] = gl_add(gl_mul(hh[c2][c], ainv[c]), term);
      }
  }
  const u64 l0 = a.l0[ig];
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
    if (c < CH) {
      const_as<u64> pw = konst(a.alpha_pow) + c * QUOTIENT_ALPHA_POWS;
      u64 l0t[QUOTIENT_MAX_CH];
#pragma unroll
      for (u32 c2 = 0; c2 < QUOTIENT_MAX_CH; c2++) l0t[c2] = c2 < CH ? gl_mul(l0, gl_sub(z0[c2], 1)) : 0;
      if (ainv[c] == 0) { res[c] = l0t[0]; continue; }  // alpha = 0: only the term of weight alpha^0 survives
      u64 r = gl_mul(res[c], pw[CH + CH * a.nchunks]);
#pragma unroll
      for (u32 c2 = 0; c2 < QUOTIENT_MAX_CH; c2++)
        if (c2 < CH) {
          r = gl_add(r, gl_mul(hh[c2][c], pw[CH + c2 * a.nchunks + a.nchunks - 1]));
          r = gl_add(r, gl_mul(l0t[c2], pw[c2]));
        }
      res[c] = r;
    }
  const u64 zhi = a.zh_inv[ig >> (lgN - a.rate_bits)];
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
    if (c < CH) a.out[(u64)c * a.N + ig] = gl_mul(res[c], zhi);
}

// filter_g(point): the selector polynomial of gate g's group with the factor of g's own value left out
__device__ __forceinline__ u64 q_filter(const QuotientArgs &a, const GateDev &G, u64 i) {
  const u64 s = a.consts[(u64)G.selector_index * a.stride + i];
  u64 f = 1;
  for (u32 j = G.group_start; j < G.group_end; j++)
    if (j != G.selector_value) f = gl_mul(f, gl_sub((u64)j, s));
  if (a.num_selectors > 1) f = gl_mul(f, gl_sub(0xFFFFFFFFull, s));
  return f;
}
// ArithmeticGate and BaseSumGate<2> of one circuit in one walk over the wires (both read the routed wires from index 0 up: the
// wires are loaded once, 16 at a time from the top, and feed both evaluators; same constraint order as q_arithmetic_native and
// q_base_sum2_native, so the sums are the same field elements)
__device__ __forceinline__ void q_arith_base_pair(const QuotientArgs &a, u64 i, const GateDev &GA, const GateDev &GB, u64 valA[QUOTIENT_MAX_CH], u64 valB[QUOTIENT_MAX_CH]) {
#if defined(__HIP_DEVICE_COMPILE__)
  const u32 CH = a.num_challenges;
  QEmit eA, eB;
  eA.CH = eB.CH = CH; eA.emitted = eB.emitted = 0;
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) { eA.acc[c] = eB.acc[c] = 0; eA.step[c] = eB.step[c] = c < CH ? konst(a.alphas)[c] : 0; }
  eA.begin_terms(); eB.begin_terms();
  const int num_ops = (int)GA.num_constraints, num_limbs = (int)GB.num_constraints - 1;
  const u64 *W = a.wires + i;
  const u64 st = a.stride;
  const u64 c0 = a.consts[(u64)a.num_selectors * st + i], c1 = a.consts[(u64)(a.num_selectors + 1) * st + i];
  const int top_wire = max(4 * num_ops, num_limbs + 1), last_wire = (int)a.num_wires - 1;
  u64 sum = 0, w0 = 0;
  for (int hi = (top_wire + 15) & ~15; hi > 0; hi -= 16) {  // wires [hi - 16, hi)
    u64 w[16];
#pragma unroll
    for (int j = 0; j < 16; j++) w[j] = W[(u64)min(hi - 16 + j, last_wire) * st];
#pragma unroll
    for (int kk = 3; kk >= 0; kk--) {
      if ((hi - 16) / 4 + kk < num_ops) {
        const u64 comp = gl_add(gl_mul(gl_mul(w[4 * kk], w[4 * kk + 1]), c0), gl_mul(w[4 * kk + 2], c1));
        eA.term(a, (u32)((hi - 16) / 4 + kk), gl_sub(w[4 * kk + 3], comp));
      }
    }
#pragma unroll
    for (int j = 15; j >= 0; j--) {
      const int limb = hi - 16 + j - 1;  // wire 0 is the sum, limb l sits on wire l + 1
      if (limb >= 0 && limb < num_limbs) {
        eB.term(a, (u32)(1 + limb), gl_mul_nc(w[j], gl_sub(w[j], 1)));
        sum = gl_add(gl_add(sum, sum), w[j]);
      }
    }
    if (hi == 16) w0 = w[0];
  }
  eB.term(a, 0, gl_sub(sum, w0));
  eA.finish_terms(); eB.finish_terms();
  const u64 fA = q_filter(a, GA, i), fB = q_filter(a, GB, i);
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
    if (c < CH) { valA[c] = gl_mul(fA, eA.acc[c]); valB[c] = gl_mul(fB, eB.acc[c]); }
#endif
}

// The light gates of a circuit in ONE launch: interpreted gates (Constant, PublicInput and whatever else has no native form) and
// the native ArithmeticGate / BaseSumGate, evaluated one after the other by the same thread with a single update of `out`.  Alone
// each of them is a latency-bound kernel of a few hundred to a few thousand instructions per point (one instruction per 5 - 8
// cycles against 3.7 for the heavy gates); together their loads overlap and three read-modify-write passes over `out` go away.
struct LightGates { u32 count; u32 g[8]; };
template <bool CHECK>
__global__ __launch_bounds__(QUOTIENT_THREADS, 2) void k_q_light(QuotientArgs a, LightGates L, u32 accumulate, unsigned long long *flag) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 T = QUOTIENT_THREADS, tid = threadIdx.x;
  const u64 i0 = (u64)blockIdx.x * T + tid;
  if (!CHECK && i0 >= a.count) return;            // no barrier is used below
  const u64 i = i0 < a.count ? i0 : a.count - 1;
  u64 sum[QUOTIENT_MAX_CH];
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) sum[c] = 0;
  // an ArithmeticGate and a BaseSumGate in the list share one walk over the wires
  u32 ka = ~0u, kb = ~0u;
  if (a.use_native)
    for (u32 k = 0; k < L.count; k++) {
      const u32 kind = konst((const u32 *)a.gates)[(size_t)L.g[k] * (sizeof(GateDev) / 4) + 7] & LCP2_GATE_NATIVE_MASK;
      if (kind == LCP2_GATE_NATIVE_ARITHMETIC && ka == ~0u) ka = k;
      if (kind == LCP2_GATE_NATIVE_BASE_SUM2 && kb == ~0u) kb = k;
    }
  const bool pair = ka != ~0u && kb != ~0u;
  if (pair) {
    const GateDev GA = q_load_gate(a, L.g[ka]), GB = q_load_gate(a, L.g[kb]);
    bool run = true;
    if (CHECK) {
      const u64 sa = a.consts[(u64)GA.selector_index * a.stride + i], sb = a.consts[(u64)GB.selector_index * a.stride + i];
      run = __any(sa == GA.selector_value || sb == GB.selector_value);
    }
    if (run) {
      u64 va[QUOTIENT_MAX_CH], vb[QUOTIENT_MAX_CH];
      q_arith_base_pair(a, i, GA, GB, va, vb);
#pragma unroll
      for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
        if (c < a.num_challenges) sum[c] = gl_add(va[c], vb[c]);
    }
  }
  for (u32 k = 0; k < L.count; k++) {
    if (pair && (k == ka || k == kb)) continue;
    const u32 g = L.g[k];
    const GateDev G = q_load_gate(a, g);
    if (CHECK) {
      const u64 sel = a.consts[(u64)G.selector_index * a.stride + i];
      if (!__any(sel == G.selector_value)) continue;  // wave-uniform
    }
    u64 val[QUOTIENT_MAX_CH];
    switch (a.use_native ? (G.flags & LCP2_GATE_NATIVE_MASK) : 0) {  // wave-uniform
      case LCP2_GATE_NATIVE_ARITHMETIC: q_gate_value<LCP2_GATE_NATIVE_ARITHMETIC>(a, g, G, i, lds, T, tid, val); break;
      case LCP2_GATE_NATIVE_BASE_SUM2: q_gate_value<LCP2_GATE_NATIVE_BASE_SUM2>(a, g, G, i, lds, T, tid, val); break;
      default: q_gate_value<0>(a, g, G, i, lds, T, tid, val); break;
    }
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
      if (c < a.num_challenges) sum[c] = gl_add(sum[c], val[c]);
  }
  if (CHECK) {
    // on a row of H only the row's own gate has a non-zero filter, so the sum is that gate's value
    bool bad = false;
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
      if (c < a.num_challenges && sum[c] != 0) bad = true;
    if (bad) atomicMin(flag, (unsigned long long)i + 1);
  } else {
    const u64 ig = a.leaf0 + i;
#pragma unroll
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
      if (c < a.num_challenges) a.out[(u64)c * a.N + ig] = accumulate ? gl_add(a.out[(u64)c * a.N + ig], sum[c]) : sum[c];
  }
}

// build()-time check of a native evaluator against the program it claims to be (random points in a.wires / a.consts)
template <u32 NATIVE>
__global__ __launch_bounds__(QUOTIENT_THREADS, 2) void k_native_check(QuotientArgs a, u32 g, unsigned long long *flag) {
  extern __shared__ __attribute__((aligned(16))) u64 lds[];
  const u32 T = QUOTIENT_THREADS, tid = threadIdx.x;
  const u64 i = (u64)blockIdx.x * T + tid;
  if (i >= a.count) return;
  const GateDev G = q_load_gate(a, g);
  u64 r0[QUOTIENT_MAX_CH], r1[QUOTIENT_MAX_CH];
  q_gate_value<0>(a, g, G, i, lds, T, tid, r0);
  q_gate_value<NATIVE>(a, g, G, i, lds, T, tid, r1);
  bool bad = false;
#pragma unroll
  for (u32 c = 0; c < QUOTIENT_MAX_CH; c++)
    if (c < a.num_challenges && r0[c] != r1[c]) bad = true;
  if (bad) atomicMin(flag, (unsigned long long)i + 1);
}

// a generated evaluator lives in one of the kernels_gates_*.hip units: ask them in turn
static bool launch_generated(hipStream_t s, const QuotientArgs &a, u32 k, u32 g, u32 accumulate, unsigned long long *flag, u32 mode) {
  return launch_generated_sha(s, a, k, g, accumulate, flag, mode) || launch_generated_u32a(s, a, k, g, accumulate, flag, mode) ||
         launch_generated_u32b(s, a, k, g, accumulate, flag, mode) || launch_generated_reca(s, a, k, g, accumulate, flag, mode) ||
         launch_generated_recb(s, a, k, g, accumulate, flag, mode);
}
namespace {
template <bool CHECK>
void launch_gate(hipStream_t s, const QuotientArgs &a, const GateDev &G, u32 g, u32 accumulate, unsigned long long *flag) {
  const dim3 grid((unsigned)((a.count + QUOTIENT_THREADS - 1) / QUOTIENT_THREADS)), block(QUOTIENT_THREADS);
  const size_t stage = (size_t)QUOTIENT_STAGE * QUOTIENT_THREADS * sizeof(u64), interp = (size_t)(a.num_regs + QUOTIENT_STAGE) * QUOTIENT_THREADS * sizeof(u64);
  const u32 kind = a.use_native ? (G.flags & LCP2_GATE_NATIVE_MASK) : 0;
  if (kind & 0x8000u) {  // validate_programs has bounded the index
    launch_generated(s, a, (kind >> 8) & 0x7Fu, g, accumulate, flag, CHECK ? GEN_ROW_CHECK : GEN_QUOTIENT);
    return;
  }
  switch (kind) {
    case LCP2_GATE_NATIVE_POSEIDON: hipLaunchKernelGGL((k_q_gate<LCP2_GATE_NATIVE_POSEIDON, CHECK>), grid, block, stage, s, a, g, accumulate, flag); break;
    case LCP2_GATE_NATIVE_ARITHMETIC: hipLaunchKernelGGL((k_q_gate<LCP2_GATE_NATIVE_ARITHMETIC, CHECK>), grid, block, 0, s, a, g, accumulate, flag); break;
    case LCP2_GATE_NATIVE_BASE_SUM2: hipLaunchKernelGGL((k_q_gate<LCP2_GATE_NATIVE_BASE_SUM2, CHECK>), grid, block, 0, s, a, g, accumulate, flag); break;
    default: hipLaunchKernelGGL((k_q_gate<0, CHECK>), grid, block, interp, s, a, g, accumulate, flag); break;
  }
}
}  // namespace

namespace {
// a gate that goes into the light-gate launch: no native form, or one of the two small native ones
bool is_light(const QuotientArgs &a, const GateDev &G) {
  const u32 k = a.use_native ? (G.flags & LCP2_GATE_NATIVE_MASK) : 0;
  return k == 0 || k == LCP2_GATE_NATIVE_ARITHMETIC || k == LCP2_GATE_NATIVE_BASE_SUM2;
}
template <bool CHECK>
u32 launch_gates(hipStream_t s, const QuotientArgs &a, const std::vector<GateDev> &host_gates, unsigned long long *flag) {
  u32 launched = 0;
  LightGates L{};
  auto flush = [&] {
    if (!L.count) return;
    const dim3 grid((unsigned)((a.count + QUOTIENT_THREADS - 1) / QUOTIENT_THREADS)), block(QUOTIENT_THREADS);
    const size_t lds = (size_t)(a.num_regs + QUOTIENT_STAGE) * QUOTIENT_THREADS * sizeof(u64);
    hipLaunchKernelGGL((k_q_light<CHECK>), grid, block, lds, s, a, L, launched ? 1u : 0u, flag);
    launched++;
    L.count = 0;
  };
  for (u32 g = 0; g < host_gates.size(); g++) {
    if (host_gates[g].num_constraints == 0) continue;
    if (is_light(a, host_gates[g])) {
      L.g[L.count++] = g;
      if (L.count == 8) flush();
      continue;
    }
    launch_gate<CHECK>(s, a, host_gates[g], g, launched ? 1u : 0u, flag);
    launched++;
  }
  flush();
  return launched;
}
}  // namespace

// host_gates: the gate table as uploaded (staged code offsets); gates without constraints are skipped
void launch_quotient(hipStream_t s, const QuotientArgs &a, const std::vector<GateDev> &host_gates) {
  const u32 launched = launch_gates<false>(s, a, host_gates, nullptr);
  hipLaunchKernelGGL(k_q_perm, dim3((unsigned)((a.count + QUOTIENT_THREADS - 1) / QUOTIENT_THREADS)), dim3(QUOTIENT_THREADS), 0, s, a, launched ? 1u : 0u);
}
void launch_gate_check(hipStream_t s, const QuotientArgs &a, const std::vector<GateDev> &host_gates, unsigned long long *flag) {
  launch_gates<true>(s, a, host_gates, flag);
}
void launch_native_check(hipStream_t s, const QuotientArgs &a, const std::vector<GateDev> &host_gates, unsigned long long *flag) {
  const dim3 grid((unsigned)((a.count + QUOTIENT_THREADS - 1) / QUOTIENT_THREADS)), block(QUOTIENT_THREADS);
  const size_t lds = (size_t)(a.num_regs + QUOTIENT_STAGE) * QUOTIENT_THREADS * sizeof(u64);
  for (u32 g = 0; g < host_gates.size(); g++) {
    const u32 kind = host_gates[g].flags & LCP2_GATE_NATIVE_MASK;
    if (kind & 0x8000u) { launch_generated(s, a, (kind >> 8) & 0x7Fu, g, 0, flag, GEN_CLAIM_CHECK); continue; }
    switch (kind) {
      case LCP2_GATE_NATIVE_POSEIDON: hipLaunchKernelGGL((k_native_check<LCP2_GATE_NATIVE_POSEIDON>), grid, block, lds, s, a, g, flag); break;
      case LCP2_GATE_NATIVE_ARITHMETIC: hipLaunchKernelGGL((k_native_check<LCP2_GATE_NATIVE_ARITHMETIC>), grid, block, lds, s, a, g, flag); break;
      case LCP2_GATE_NATIVE_BASE_SUM2: hipLaunchKernelGGL((k_native_check<LCP2_GATE_NATIVE_BASE_SUM2>), grid, block, lds, s, a, g, flag); break;
      default: break;
    }
  }
}

// ------------------------------------------------------------------ K7a: evaluate coefficient polynomials at an extension point
// grid (chunks, polys); a chunk is EVAL_CHUNK coefficients; thread t owns coefficients t, t+256, ...
// partial[poly][chunk] = z^(chunk*EVAL_CHUNK) * sum_t z^t * Horner_m(c[t + 256 m]; z^256)
__global__ __launch_bounds__(256) void k_eval_polys(EvalArgs a) {
  __shared__ u64 sh0[256], sh1[256];
  const u32 t = threadIdx.x, chunk = blockIdx.x, poly = blockIdx.y;
  const u64 *c = a.coeffs + (u64)poly * a.col_stride + (u64)chunk * a.chunk_len;
  // Horner in lazy arithmetic: acc = acc * z^256 + c with acc any u64 pair congruent to the value (gl_mul_nc takes it), one
  // canonical product per component so that gl_add_nc has its canonical operand, and 7 z_1 hoisted out of the loop
  gl2 acc = gl2_make(0, 0);
  const u64 z0 = a.zstep[0], z1 = a.zstep[1], z1w = gl_mul(GL_W, z1);
  for (int m = (int)a.items - 1; m >= 0; m--) {
    u32 idx = t + 256u * (u32)m;
    const u64 coeff = idx < a.chunk_len ? c[idx] : 0;  // canonical: the coefficient buffers are the library's own
    const u64 n0 = gl_add_nc(gl_add_nc(gl_mul_nc(acc.c0, z0), gl_mul(acc.c1, z1w)), coeff);
    const u64 n1 = gl_add_nc(gl_mul_nc(acc.c0, z1), gl_mul(acc.c1, z0));
    acc = gl2_make(n0, n1);
  }
  acc = gl2_mul(gl2_make(gl_canon(acc.c0), gl_canon(acc.c1)), gl2_make(a.zpow_t[2 * t], a.zpow_t[2 * t + 1]));
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

// ------------------------------------------------------------------ challenge-dependent setup (prover_kernels.hpp)
__global__ void k_set_words(u64 *dst, SmallWords w, u32 n) {
  if (threadIdx.x < n) dst[threadIdx.x] = w.v[threadIdx.x];
}
void launch_set_words(hipStream_t s, u64 *dst, const SmallWords &w, u32 n) {
  hipLaunchKernelGGL(k_set_words, dim3(1), dim3(16), 0, s, dst, w, n);
}

__global__ __launch_bounds__(256) void k_quotient_setup(QuotientSetupArgs a) {
  const u32 t = threadIdx.x, CH = a.num_challenges;
  if (t < QUOTIENT_TERM_POWS) {  // alpha_c^t: the limb table and the first QUOTIENT_ALPHA_POWS plain powers
    for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) {
      const u64 pw = c < CH ? gl_pow(a.alphas[c], t) : 0;
      u32 *w = a.limbs + ((size_t)c * QUOTIENT_TERM_POWS + t) * 4;
      w[0] = (u32)(pw & 0x3FFFFF); w[1] = (u32)((pw >> 22) & 0x3FFFFF); w[2] = (u32)(pw >> 44); w[3] = 0;
      if (t < QUOTIENT_ALPHA_POWS) a.small[SMALL_ALPHA_POW + (size_t)c * QUOTIENT_ALPHA_POWS + t] = pw;
    }
  } else if (t < 255) {          // alpha_c^(m_g - 1) for every gate g
    for (u32 g = t - QUOTIENT_TERM_POWS; g < a.num_gates; g += 255 - QUOTIENT_TERM_POWS) {
      const u32 m = konst((const u32 *)a.gates)[(size_t)g * (sizeof(GateDev) / 4) + 6];
      for (u32 c = 0; c < QUOTIENT_MAX_CH; c++) a.small[SMALL_GATE_SCALE + (size_t)g * QUOTIENT_MAX_CH + c] = c < CH ? (m ? gl_pow(a.alphas[c], m - 1) : 1) : 0;
    }
  } else {                       // the scalars
    for (u32 c = 0; c < 4; c++) {
      const u64 al = c < CH ? a.alphas[c] : 0;
      a.small[SMALL_ALPHAS + c] = al;
      a.small[SMALL_ALPHA_INV + c] = al ? gl_inv(al) : 0;
      a.small[SMALL_PI_HASH + c] = a.pi_hash[c];
    }
    a.small[SMALL_CHECK] = ~0ull;
  }
}
void launch_quotient_setup(hipStream_t s, const QuotientSetupArgs &a) {
  static_assert(QUOTIENT_TERM_POWS < 255 && QUOTIENT_MAX_CH <= 4 && QUOTIENT_ALPHA_POWS <= QUOTIENT_TERM_POWS, "k_quotient_setup's thread map");
  hipLaunchKernelGGL(k_quotient_setup, dim3(1), dim3(256), 0, s, a);
}

__global__ __launch_bounds__(256) void k_eval_tables(u64 z0, u64 z1, u32 chunk_len, u32 nchunks, u64 *tab) {
  const u32 t = blockIdx.x * 256 + threadIdx.x;
  const gl2 z = gl2_make(z0, z1);
  if (t < 256) { const gl2 v = gl2_pow(z, t); tab[2 * t] = v.c0; tab[2 * t + 1] = v.c1; }
  else if (t - 256 < nchunks) { const u32 k = t - 256; const gl2 v = gl2_pow(z, (u64)chunk_len * k); tab[512 + 2 * k] = v.c0; tab[513 + 2 * k] = v.c1; }
}
void launch_eval_tables(hipStream_t s, u64 z0, u64 z1, u32 chunk_len, u32 nchunks, u64 *tab) {
  hipLaunchKernelGGL(k_eval_tables, dim3((256 + nchunks + 255) / 256), dim3(256), 0, s, z0, z1, chunk_len, nchunks, tab);
}

__global__ __launch_bounds__(256) void k_compose_tables(u64 a0, u64 a1, u64 z0, u64 z1, u64 g, u32 total_polys, u32 h, u64 hi_count, u64 *T) {
  const u64 t = (u64)blockIdx.x * 256 + threadIdx.x, per = (1ull << h) + hi_count;
  gl2 v;
  if (t < total_polys) v = gl2_pow(gl2_make(a0, a1), t);
  else {
    const u64 r = t - total_polys, b = r / per, j = r % per;
    if (b >= 4) return;
    gl2 base = gl2_make(z0, z1);
    if (b & 1) base = gl2_scale(base, g);
    if (b & 2) base = gl2_inv(base);
    v = j < (1ull << h) ? gl2_pow(base, j) : gl2_pow(base, (j - (1ull << h)) << h);
  }
  T[2 * t] = v.c0; T[2 * t + 1] = v.c1;
}
void launch_compose_tables(hipStream_t s, u64 a0, u64 a1, u64 z0, u64 z1, u64 g, u32 total_polys, u32 h, u64 hi_count, u64 *T) {
  const u64 entries = total_polys + 4 * ((1ull << h) + hi_count);
  hipLaunchKernelGGL(k_compose_tables, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, s, a0, a1, z0, z1, g, total_polys, h, hi_count, T);
}

__global__ void k_fill(u64 *p, u64 n, u64 v) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void launch_fill(hipStream_t s, u64 *p, u64 n, u64 v) {
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}

}  // namespace lcp2
