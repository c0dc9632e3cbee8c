// Poseidon / Merkle / SHA-256 kernels for gfx950 (rows a5, a6, a15 of SURVEY.md section 8).
//
//  k_poseidon_permute_batch  PoseidonPermutation::permute           (plonky2 hash/poseidon.rs)
//  k_hash_leaves             PoseidonHash::hash_or_noop per leaf    (hash/merkle_tree.rs, MerkleTree::new)
//  k_hash_ext_leaves         same for FRI layer leaves (flatten of 2^arity_bits extension elements)
//  k_merkle_level            PoseidonHash::two_to_one per node
//  k_sha256_level            two_to_one_sha256 (reference src/merkle_tree_gadget.rs:28-40)
//
//  k_merkle_level_coop / k_hash_ext_leaves_coop   the same two for levels of <= 4096 nodes, one 16-lane group per
//                            permutation with wavefront shuffles for the MDS layer (latency instead of throughput)
//
// One permutation per lane everywhere else: the 12-element state is 24 VGPRs, round constants
// are wave-uniform scalar loads, and the LDE matrix is column-major so that the
// lanes of a wave read 512 consecutive bytes of each column (no transpose pass,
// K3 of the survey is fused away).  These kernels are integer-ALU bound.
#include "internal.hpp"
#include "poseidon.hpp"

namespace lcp2 {

constexpr int HASH_THREADS = 256;

__global__ __launch_bounds__(HASH_THREADS) void k_poseidon_permute_batch(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                                          size_t count, const u64 *__restrict__ rc) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  u64 s[12];
#pragma unroll
  for (int j = 0; j < 12; j++) s[j] = gl_canon(in[i * 12 + j]);
  pos_permute(s, rc);
#pragma unroll
  for (int j = 0; j < 12; j++) out[i * 12 + j] = s[j];
}

// The device field arithmetic every kernel of the library shares (gl64.hpp), exposed so that the tests can drive it with operands
// that reach the rare branches (a borrow in the multiply's lo - hi_hi has probability 2^-32 on random operands).  Results are
// canonical; op as LCP2_FIELD_* of lcp2.h.
__global__ __launch_bounds__(HASH_THREADS) void k_field_op(const u64 *__restrict__ a, const u64 *__restrict__ b, u64 *__restrict__ out,
                                                            size_t count, u32 op) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
#if defined(__HIP_DEVICE_COMPILE__)
  const u64 x = a[i], y = b ? b[i] : 0;
  u64 r = 0;
  switch (op) {
    case LCP2_FIELD_MUL: r = gl_mul(x, y); break;
    case LCP2_FIELD_POW7: { u32 r0 = (u32)x, r1 = (u32)(x >> 32); pos_sbox_h(r0, r1); r = gl_canon(((u64)r1 << 32) | r0); break; }
    case LCP2_FIELD_ADD: r = gl_add(gl_canon(x), gl_canon(y)); break;
    case LCP2_FIELD_SUB: r = gl_sub(gl_canon(x), gl_canon(y)); break;
    case LCP2_FIELD_CANON: r = gl_canon(x); break;
    case LCP2_FIELD_ADD_LAZY: r = gl_canon(gl_add_nc(x, gl_canon(y))); break;
    case LCP2_FIELD_SUB_LAZY: r = gl_canon(gl_sub_nc(x, gl_canon(y))); break;
    case LCP2_FIELD_SHL + 1: r = gl_shl<12>(gl_canon(x)); break;
    case LCP2_FIELD_SHL + 2: r = gl_shl<24>(gl_canon(x)); break;
    case LCP2_FIELD_SHL + 3: r = gl_shl<36>(gl_canon(x)); break;
    case LCP2_FIELD_SHL + 4: r = gl_shl<48>(gl_canon(x)); break;
    case LCP2_FIELD_SHL + 5: r = gl_shl<60>(gl_canon(x)); break;
    case LCP2_FIELD_SHL + 6: r = gl_shl<72>(gl_canon(x)); break;
    case LCP2_FIELD_SHL + 7: r = gl_shl<84>(gl_canon(x)); break;
    case LCP2_FIELD_SHL + 8: r = gl_canon(gl_shl_nc<32>(x)); break;
    case LCP2_FIELD_SHL + 9: r = gl_canon(gl_mul_u32_nc(x, (u32)y)); break;
  }
  out[i] = r;
#endif
}

__global__ __launch_bounds__(HASH_THREADS, 5) void k_hash_leaves(const u64 *__restrict__ data, u64 leaf_stride, u64 col_stride,
                                                               u32 leaf_len, u64 nleaves, u64 *__restrict__ digests,
                                                               const u64 *__restrict__ rc) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nleaves) return;
  const u64 *row = data + i * leaf_stride;
  u64 s[12];
#pragma unroll
  for (int j = 0; j < 12; j++) s[j] = 0;
  if (leaf_len <= 4) {  // hash_or_noop: short leaves are padded, not hashed
    for (u32 c = 0; c < leaf_len; c++) s[c] = gl_canon(row[c * col_stride]);
  } else {
    for (u32 c0 = 0; c0 < leaf_len; c0 += 8) {
      if (c0 + 8 <= leaf_len) {
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] = gl_canon(row[(u64)(c0 + j) * col_stride]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++)
          if (c0 + j < leaf_len) s[j] = gl_canon(row[(u64)(c0 + j) * col_stride]);
      }
      pos_permute(s, rc);
    }
  }
  ulonglong2 *d = (ulonglong2 *)(digests + 4 * i);
  d[0] = make_ulonglong2(s[0], s[1]);
  d[1] = make_ulonglong2(s[2], s[3]);
}

// The same sponge over a CHUNK of the columns of a leaf, its state kept between launches (state[j][leaf], column-major): a sharded proof
// absorbs the coefficient chunks that have arrived while the next ones are still crossing the fabric (lcp2_commit_wires_chunk).  The
// chunks come in column order, each but the last a multiple of 8 columns, so the permutations fall where k_hash_leaves puts them.
__global__ __launch_bounds__(HASH_THREADS, 5) void k_hash_leaves_absorb(const u64 *__restrict__ data, u64 col_stride, u32 ncols, u64 nleaves,
                                                                      u64 *__restrict__ state, u32 first, u32 last, u64 *__restrict__ digests,
                                                                      const u64 *__restrict__ rc) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nleaves) return;
  const u64 *row = data + i;
  u64 s[12];
#pragma unroll
  for (int j = 0; j < 12; j++) s[j] = first ? 0 : state[(u64)j * nleaves + i];
  for (u32 c0 = 0; c0 < ncols; c0 += 8) {
    if (c0 + 8 <= ncols) {
#pragma unroll
      for (int j = 0; j < 8; j++) s[j] = gl_canon(row[(u64)(c0 + j) * col_stride]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (c0 + j < ncols) s[j] = gl_canon(row[(u64)(c0 + j) * col_stride]);
    }
    pos_permute(s, rc);
  }
  if (last) {
    ulonglong2 *d = (ulonglong2 *)(digests + 4 * i);
    d[0] = make_ulonglong2(s[0], s[1]);
    d[1] = make_ulonglong2(s[2], s[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 12; j++) state[(u64)j * nleaves + i] = s[j];
  }
}

__global__ __launch_bounds__(HASH_THREADS, 5) void k_hash_ext_leaves(const u64 *__restrict__ p0, const u64 *__restrict__ p1, u32 arity,
                                                                   u64 nleaves, u64 *__restrict__ digests,
                                                                   const u64 *__restrict__ rc) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nleaves) return;
  const u64 *a = p0 + i * arity, *b = p1 + i * arity;
  u64 s[12];
#pragma unroll
  for (int j = 0; j < 12; j++) s[j] = 0;
  u32 len = 2 * arity;
  if (len <= 4) {
    for (u32 e = 0; e < arity; e++) { s[2 * e] = a[e]; s[2 * e + 1] = b[e]; }
  } else {
    for (u32 e0 = 0; e0 < arity; e0 += 4) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (e0 + j < arity) { s[2 * j] = a[e0 + j]; s[2 * j + 1] = b[e0 + j]; }
      pos_permute(s, rc);
    }
  }
  ulonglong2 *d = (ulonglong2 *)(digests + 4 * i);
  d[0] = make_ulonglong2(s[0], s[1]);
  d[1] = make_ulonglong2(s[2], s[3]);
}

__global__ __launch_bounds__(HASH_THREADS) void k_merkle_level(const u64 *__restrict__ children, u64 *__restrict__ parents,
                                                                u64 nparents, const u64 *__restrict__ rc) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nparents) return;
  const ulonglong2 *c = (const ulonglong2 *)(children + 8 * i);
  ulonglong2 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
  u64 s[12] = {c0.x, c0.y, c1.x, c1.y, c2.x, c2.y, c3.x, c3.y, 0, 0, 0, 0};
  pos_permute(s, rc);
  ulonglong2 *d = (ulonglong2 *)(parents + 4 * i);
  d[0] = make_ulonglong2(s[0], s[1]);
  d[1] = make_ulonglong2(s[2], s[3]);
}

// ---- latency-bound sizes: one 16-lane group per permutation (poseidon.hpp: pos_permute_coop)
constexpr int COOP_THREADS = 256;
__device__ __forceinline__ void coop_load_rc(u64 *rcs, const u64 *__restrict__ rc) {
  for (u32 i = threadIdx.x; i < POS_ROUNDS * POS_W; i += COOP_THREADS) rcs[i] = rc[i];
  __syncthreads();
}
__global__ __launch_bounds__(COOP_THREADS) void k_merkle_level_coop(const u64 *__restrict__ children, u64 *__restrict__ parents,
                                                                      u64 nparents, const u64 *__restrict__ rc) {
  __shared__ u64 rcs[POS_ROUNDS * POS_W];
  coop_load_rc(rcs, rc);
  const u64 t = (u64)blockIdx.x * COOP_THREADS + threadIdx.x, node = t >> 4;
  const u32 j = (u32)t & 15;
  const bool live = node < nparents;  // every lane runs the shuffles; dead groups recompute node 0
  const u64 nd = live ? node : 0;
  const u64 v = j < 8 ? children[8 * nd + j] : 0;
  const u64 r = pos_permute_coop(v, j, rcs);
  if (live && j < 4) parents[4 * nd + j] = r;
}
// FRI layer leaves of 2 * arity > 4 elements: absorb 8 at a time (overwrite mode)
__global__ __launch_bounds__(COOP_THREADS) void k_hash_ext_leaves_coop(const u64 *__restrict__ p0, const u64 *__restrict__ p1, u32 arity,
                                                                         u64 nleaves, u64 *__restrict__ digests, const u64 *__restrict__ rc) {
  __shared__ u64 rcs[POS_ROUNDS * POS_W];
  coop_load_rc(rcs, rc);
  const u64 t = (u64)blockIdx.x * COOP_THREADS + threadIdx.x, leaf = t >> 4;
  const u32 j = (u32)t & 15;
  const bool live = leaf < nleaves;
  const u64 lf = live ? leaf : 0;
  const u64 *src = ((j & 1) ? p1 : p0) + lf * arity + (j >> 1);
  u64 v = 0;
  for (u32 e0 = 0; e0 < arity; e0 += 4) {
    if (j < 8 && e0 + (j >> 1) < arity) v = src[e0];
    v = pos_permute_coop(v, j, rcs);
  }
  if (live && j < 4) digests[4 * lf + j] = v;
}

// ------------------------------------------------------------------ SHA-256
__device__ __constant__ uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int r) { return __builtin_rotateright32(x, r); }

// one compression; trace (nullable): 48 schedule words then 64 (a_new, e_new) pairs
__device__ __forceinline__ void sha256_compress(uint32_t st[8], uint32_t w[16], uint32_t *trace) {
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll 1
  for (int i = 0; i < 64; i++) {
    uint32_t wi;
    if (i < 16) {
      wi = w[i & 15];
    } else {
      uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      uint32_t s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
      uint32_t s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
      wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      w[i & 15] = wi;
      if (trace) trace[i - 16] = wi;
    }
    uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = h + S1 + ch + SHA_K[i] + wi;
    uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    if (trace) { trace[48 + 2 * i] = a; trace[48 + 2 * i + 1] = e; }
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

constexpr int SHA_TRACE_WORDS = 48 + 128;  // per compression

// parents[t][i] = SHA256(children[t][2i] || children[t][2i+1]); digests are 32 bytes
__global__ __launch_bounds__(64) void k_sha256_level(const uint8_t *__restrict__ children, uint8_t *__restrict__ parents,
                                                      u64 nparents, u64 trees, u64 child_tree_stride, u64 parent_tree_stride,
                                                      uint32_t *__restrict__ trace, u64 trace_tree_stride, u64 trace_hash_offset) {
  u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nparents * trees) return;
  u64 t = gid / nparents, i = gid % nparents;
  const uint32_t *src = (const uint32_t *)(children + t * child_tree_stride + 64 * i);
  uint32_t w[16];
#pragma unroll
  for (int j = 0; j < 16; j++) w[j] = __builtin_bswap32(src[j]);
  uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  uint32_t *tr = trace ? trace + t * trace_tree_stride + (trace_hash_offset + i) * 2 * SHA_TRACE_WORDS : nullptr;
  sha256_compress(st, w, tr);
#pragma unroll
  for (int j = 0; j < 16; j++) w[j] = 0;
  w[0] = 0x80000000u; w[15] = 512;
  sha256_compress(st, w, tr ? tr + SHA_TRACE_WORDS : nullptr);
  uint32_t *dst = (uint32_t *)(parents + t * parent_tree_stride + 32 * i);
#pragma unroll
  for (int j = 0; j < 8; j++) dst[j] = __builtin_bswap32(st[j]);
}

// ------------------------------------------------------------------ query-phase gathers
// out[q][c] = data[c * col_stride + indices[q]]
__global__ void k_gather_rows(const u64 *__restrict__ data, u64 col_stride, u32 ncols, const u64 *__restrict__ indices, u32 k,
                              u64 *__restrict__ out) {
  u32 q = blockIdx.x;
  if (q >= k) return;
  u64 idx = indices[q];
  for (u32 c = threadIdx.x; c < ncols; c += blockDim.x) out[(u64)q * ncols + c] = data[(u64)c * col_stride + idx];
}
// out[q][l][0..4] = sibling digest of leaf indices[q] at level l
__global__ void k_gather_digests(const u64 *__restrict__ digests, const u64 *__restrict__ level_offsets, u32 nsib,
                                 const u64 *__restrict__ indices, u32 k, u64 *__restrict__ out) {
  u32 q = blockIdx.x;
  if (q >= k) return;
  u64 idx = indices[q];
  for (u32 t = threadIdx.x; t < nsib * 4; t += blockDim.x) {
    u32 l = t >> 2, e = t & 3;
    u64 node = (idx >> l) ^ 1;
    out[((u64)q * nsib + l) * 4 + e] = digests[4 * (level_offsets[l] + node) + e];
  }
}

// ------------------------------------------------------------------ launch wrappers
static inline unsigned blocks_for(u64 n, unsigned threads) { return (unsigned)((n + threads - 1) / threads); }

void launch_poseidon_permute_batch(hipStream_t s, const u64 *in, u64 *out, size_t count, const u64 *rc) {
  if (!count) return;
  hipLaunchKernelGGL(k_poseidon_permute_batch, dim3(blocks_for(count, HASH_THREADS)), dim3(HASH_THREADS), 0, s, in, out, count, rc);
}
void launch_field_op(hipStream_t s, const u64 *a, const u64 *b, u64 *out, size_t count, u32 op) {
  if (!count) return;
  hipLaunchKernelGGL(k_field_op, dim3(blocks_for(count, HASH_THREADS)), dim3(HASH_THREADS), 0, s, a, b, out, count, op);
}
void launch_hash_leaves(hipStream_t s, const u64 *data, u64 leaf_stride, u64 col_stride, u32 leaf_len, u64 nleaves, u64 *digests,
                        const u64 *rc) {
  hipLaunchKernelGGL(k_hash_leaves, dim3(blocks_for(nleaves, HASH_THREADS)), dim3(HASH_THREADS), 0, s, data, leaf_stride, col_stride,
                     leaf_len, nleaves, digests, rc);
}
void launch_hash_leaves_absorb(hipStream_t s, const u64 *data, u64 col_stride, u32 ncols, u64 nleaves, u64 *state, bool first, bool last,
                               u64 *digests, const u64 *rc) {
  hipLaunchKernelGGL(k_hash_leaves_absorb, dim3(blocks_for(nleaves, HASH_THREADS)), dim3(HASH_THREADS), 0, s, data, col_stride, ncols, nleaves,
                     state, first ? 1u : 0u, last ? 1u : 0u, digests, rc);
}
void launch_hash_ext_leaves(hipStream_t s, const u64 *p0, const u64 *p1, u32 arity, u64 nleaves, u64 *digests, const u64 *rc) {
  if (nleaves <= POS_COOP_MAX_NODES && 2 * arity > 4) {
    hipLaunchKernelGGL(k_hash_ext_leaves_coop, dim3(blocks_for(nleaves * 16, COOP_THREADS)), dim3(COOP_THREADS), 0, s, p0, p1, arity, nleaves, digests, rc);
    return;
  }
  hipLaunchKernelGGL(k_hash_ext_leaves, dim3(blocks_for(nleaves, HASH_THREADS)), dim3(HASH_THREADS), 0, s, p0, p1, arity, nleaves,
                     digests, rc);
}
void launch_merkle_level(hipStream_t s, const u64 *children, u64 *parents, u64 nparents, const u64 *rc) {
  if (nparents <= POS_COOP_MAX_NODES) {
    hipLaunchKernelGGL(k_merkle_level_coop, dim3(blocks_for(nparents * 16, COOP_THREADS)), dim3(COOP_THREADS), 0, s, children, parents, nparents, rc);
    return;
  }
  hipLaunchKernelGGL(k_merkle_level, dim3(blocks_for(nparents, HASH_THREADS)), dim3(HASH_THREADS), 0, s, children, parents, nparents, rc);
}
void launch_sha256_level(hipStream_t s, const uint8_t *children, uint8_t *parents, u64 nparents, u64 trees, u64 child_tree_stride,
                         u64 parent_tree_stride, uint32_t *trace, u64 trace_tree_stride, u64 trace_hash_offset) {
  hipLaunchKernelGGL(k_sha256_level, dim3(blocks_for(nparents * trees, 64)), dim3(64), 0, s, children, parents, nparents, trees,
                     child_tree_stride, parent_tree_stride, trace, trace_tree_stride, trace_hash_offset);
}
void launch_gather_rows(hipStream_t s, const u64 *data, u64 col_stride, u32 ncols, const u64 *indices, u32 k, u64 *out) {
  if (!k) return;
  hipLaunchKernelGGL(k_gather_rows, dim3(k), dim3(64), 0, s, data, col_stride, ncols, indices, k, out);
}
void launch_gather_digests(hipStream_t s, const u64 *digests, const u64 *level_offsets, u32 nsib, const u64 *indices, u32 k, u64 *out) {
  if (!k || !nsib) return;
  hipLaunchKernelGGL(k_gather_digests, dim3(k), dim3(64), 0, s, digests, level_offsets, nsib, indices, k, out);
}

}  // namespace lcp2
