// K10, second half: SHA-256 witness generation on the device (row a15 of SURVEY.md section 8).
//
// The reference fills the wires of every SHA-256 row on one host thread (plonky2's generator worklist running
// plonky2_crypto's U32 / SHA-256 generators).  Here a circuit built from the SHA-256 row layout of
// eth-lc-plonky2_amd/host/gates.cpp is filled in HBM by two kernels:
//   k_sha_jobs_level  one lane per two_to_one_sha256 of a dependency level: gathers its 16 message words (host
//                     supplied leaves or digests of earlier levels), runs both compressions and stores a 336-word
//                     record (message, schedule, per-round (a, e), chaining values)
//   k_sha_fill_rows   one lane per circuit ROW (310 per hash): expands the record into that row's cells (words, bit
//                     decompositions, carries) with column-major stores, so lanes of a wave write runs of a column
// plus k_scatter_cells for the handful of non-SHA cells (constants, arithmetic glue, public inputs).
#include "internal.hpp"
#include "sha_layout.hpp"
#include "poseidon.hpp"

namespace lcp2 {

__device__ __constant__ uint32_t WSHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
__device__ __constant__ uint32_t WSHA_IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

__device__ __forceinline__ uint32_t wrotr(uint32_t x, int r) { return __builtin_rotateright32(x, r); }

// message schedule of the constant padding block (0x80000000, 0, ..., 0, 512) of a 64-byte message
__device__ __constant__ uint32_t WSHA_PAD_W[64] = {0x80000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000000, 0x00000200, 0x80000000, 0x01400000, 0x00205000, 0x00005088, 0x22000800, 0x22550014, 0x05089742, 0xa0000020, 0x5a880000, 0x005c9400, 0x0016d49d, 0xfa801f00, 0xd33225d0, 0x11675959, 0xf6e6bfda, 0xb30c1549, 0x08b2b050, 0x9d7c4c27, 0x0ce2a393, 0x88e6e1ea, 0xa52b4335, 0x67a16f49, 0xd732016f, 0x4eeb2e91, 0x5dbf55e5, 0x8eee2335, 0xe2bc5ec2, 0xa83f4394, 0x45ad78f7, 0x36f3d0cd, 0xd99c05e8, 0xb0511dc7, 0x69bc7ac4, 0xbd11375b, 0xe3ba71e5, 0x3b209ff2, 0x18feee17, 0xe25ad9e7, 0x13375046, 0x0515089d, 0x4f0d0f04, 0x2627484e, 0x310128d2, 0xc668b434, 0x420841cc, 0x62d311b8, 0xe59ba771, 0x85a7a484};

__global__ __launch_bounds__(64) void k_sha_jobs_level(const ShaJobDev *__restrict__ jobs, u32 first, u32 count,
                                                        const uint32_t *__restrict__ words_in, uint32_t *__restrict__ rec) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const u32 j = first + k;
  const ShaJobDev job = jobs[j];
  uint32_t *R = rec + (u64)j * SHA_REC_WORDS;
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    int s = job.in_src[i];
    w[i] = s >= 0 ? words_in[s] : rec[(u64)((~s) >> 3) * SHA_REC_WORDS + SHA_REC_DIGEST + ((~s) & 7)];
    R[SHA_REC_IN + i] = w[i];
  }
  uint32_t chain[8];
#pragma unroll
  for (int i = 0; i < 8; i++) chain[i] = WSHA_IV[i];
  for (int c = 0; c < 2; c++) {
    uint32_t a = chain[0], b = chain[1], cc = chain[2], d = chain[3], e = chain[4], f = chain[5], g = chain[6], h = chain[7];
    if (c == 1) {
#pragma unroll
      for (int i = 0; i < 16; i++) w[i] = 0;
      w[0] = 0x80000000u; w[15] = 512;
    }
    uint32_t *AE = R + (c == 0 ? SHA_REC_AE0 : SHA_REC_AE1);
#pragma unroll 1
    for (int t0 = 0; t0 < 64; t0 += 16) {
#pragma unroll
     for (int ti = 0; ti < 16; ti++) {
      const int t = t0 + ti;
      uint32_t wt;
      if (t0 == 0) wt = w[ti];
      else {
        uint32_t w15 = w[(ti + 1) & 15], w2 = w[(ti + 14) & 15];
        uint32_t s0 = wrotr(w15, 7) ^ wrotr(w15, 18) ^ (w15 >> 3), s1 = wrotr(w2, 17) ^ wrotr(w2, 19) ^ (w2 >> 10);
        wt = w[ti] + s0 + w[(ti + 9) & 15] + s1;
        w[ti] = wt;
        if (c == 0) R[SHA_REC_SCHED + t - 16] = wt;
      }
      uint32_t S1 = wrotr(e, 6) ^ wrotr(e, 11) ^ wrotr(e, 25), ch = (e & f) ^ (~e & g);
      uint32_t t1 = h + S1 + ch + WSHA_K[t] + wt;
      uint32_t S0 = wrotr(a, 2) ^ wrotr(a, 13) ^ wrotr(a, 22), mj = (a & b) ^ (a & cc) ^ (b & cc);
      h = g; g = f; f = e; e = d + t1; d = cc; cc = b; b = a; a = t1 + S0 + mj;
      AE[2 * t] = a; AE[2 * t + 1] = e;
     }
    }
    chain[0] += a; chain[1] += b; chain[2] += cc; chain[3] += d; chain[4] += e; chain[5] += f; chain[6] += g; chain[7] += h;
    uint32_t *O = R + (c == 0 ? SHA_REC_MID : SHA_REC_DIGEST);
#pragma unroll
    for (int i = 0; i < 8; i++) O[i] = chain[i];
  }
}

// state word helpers on a record: a_t / e_t = register a / e AFTER round t of compression c; negative t = chaining input
__device__ __forceinline__ uint32_t rec_a(const uint32_t *R, int c, int t) {
  if (t >= 0) return R[(c == 0 ? SHA_REC_AE0 : SHA_REC_AE1) + 2 * t];
  return c == 0 ? WSHA_IV[-1 - t] : R[SHA_REC_MID + (-1 - t)];  // t = -1 -> a, -2 -> b, -3 -> c, -4 -> d
}
__device__ __forceinline__ uint32_t rec_e(const uint32_t *R, int c, int t) {
  if (t >= 0) return R[(c == 0 ? SHA_REC_AE0 : SHA_REC_AE1) + 2 * t + 1];
  return c == 0 ? WSHA_IV[4 + (-1 - t)] : R[SHA_REC_MID + 4 + (-1 - t)];
}

__global__ __launch_bounds__(256) void k_sha_fill_rows(const ShaJobDev *__restrict__ jobs, u32 njobs, const uint32_t *__restrict__ rec,
                                                        u64 *__restrict__ wires, u64 n) {
  u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (u64)njobs * SHA_ROWS) return;
  const u32 j = (u32)(gid / SHA_ROWS), lr = (u32)(gid % SHA_ROWS);
  const uint32_t *R = rec + (u64)j * SHA_REC_WORDS;
  const u64 row = (u64)jobs[j].first_row + lr;
  u64 *Wp = wires + row;
  auto put = [&](u32 col, u64 v) { Wp[(u64)col * n] = v; };
  auto bits = [&](u32 base, uint32_t x) {
    for (int i = 0; i < 32; i++) Wp[(u64)(base + i) * n] = (x >> i) & 1;
  };
  auto msg = [&](int t) -> uint32_t { return t < 16 ? R[SHA_REC_IN + t] : R[SHA_REC_SCHED + t - 16]; };
  // every cell of the row is written (unused ones with 0), so a reused witness buffer needs no clearing
  if (lr < SHA_ROW_ROUNDS0) {  // schedule row for W_t, t = 16 + lr
    const int t = 16 + (int)lr;
    uint32_t w2 = msg(t - 2), w7 = msg(t - 7), w15 = msg(t - 15), w16 = msg(t - 16);
    uint32_t s0 = wrotr(w15, 7) ^ wrotr(w15, 18) ^ (w15 >> 3), s1 = wrotr(w2, 17) ^ wrotr(w2, 19) ^ (w2 >> 10);
    u64 sum = (u64)s1 + w7 + s0 + w16;
    put(0, w2); put(1, w7); put(2, w15); put(3, w16); put(4, (uint32_t)sum); put(5, 0); put(6, 0); put(7, 0);
    bits(8, w2); bits(40, w15);
    for (u32 c = 72; c < 104; c++) put(c, 0);
    put(104, (sum >> 32) & 1); put(105, (sum >> 33) & 1); put(106, 0); put(107, 0);
    return;
  }
  u32 q = lr - SHA_ROW_ROUNDS0;
  int c = 0;
  if (q >= 128 + 3) { q -= 128 + 3; c = 1; }
  if (q < 128) {
    const int t = (int)(q >> 1);
    if ((q & 1) == 0) {  // round E row
      uint32_t e = rec_e(R, c, t - 1), f = rec_e(R, c, t - 2), g = rec_e(R, c, t - 3), h = rec_e(R, c, t - 4), d = rec_a(R, c, t - 4);
      uint32_t wv = c == 0 ? msg(t) : 0;
      u64 kw = WSHA_K[t];
      if (c == 1) kw += WSHA_PAD_W[t];
      uint32_t S1 = wrotr(e, 6) ^ wrotr(e, 11) ^ wrotr(e, 25), ch = (e & f) ^ (~e & g);
      u64 sum1 = (u64)h + S1 + ch + kw + wv;
      uint32_t t1 = (uint32_t)sum1;
      u64 sume = (u64)d + t1;
      put(0, e); put(1, f); put(2, g); put(3, h); put(4, d); put(5, wv); put(6, (uint32_t)sume); put(7, t1);
      bits(8, e); bits(40, f); bits(72, g);
      u64 k1 = sum1 >> 32;
      put(104, k1 & 1); put(105, (k1 >> 1) & 1); put(106, (k1 >> 2) & 1); put(107, sume >> 32);
    } else {  // round A row
      uint32_t a = rec_a(R, c, t - 1), b = rec_a(R, c, t - 2), cc = rec_a(R, c, t - 3);
      // t1 of this round = a_t - S0(a) - Maj(a,b,c)  (mod 2^32)
      uint32_t S0 = wrotr(a, 2) ^ wrotr(a, 13) ^ wrotr(a, 22), mj = (a & b) ^ (a & cc) ^ (b & cc);
      uint32_t a_new = rec_a(R, c, t);
      uint32_t t1 = a_new - S0 - mj;
      u64 suma = (u64)t1 + S0 + mj;
      put(0, a); put(1, b); put(2, cc); put(3, t1); put(4, a_new); put(5, 0); put(6, 0); put(7, 0);
      bits(8, a); bits(40, b); bits(72, cc);
      u64 k2 = suma >> 32;
      put(104, k2 & 1); put(105, (k2 >> 1) & 1); put(106, 0); put(107, 0);
    }
    return;
  }
  // addition rows: out_i = chain_i + state_i, three per row
  const u32 ar = q - 128;
  for (u32 col = 0; col < 108; col++) put(col, 0);
  for (int jj = 0; jj < 3; jj++) {
    int i = (int)ar * 3 + jj;
    if (i >= 8) break;
    uint32_t chain = c == 0 ? WSHA_IV[i] : R[SHA_REC_MID + i];
    uint32_t st = i < 4 ? rec_a(R, c, 63 - i) : rec_e(R, c, 63 - (i - 4));
    u64 sum = (u64)chain + st;
    put(3 * jj, chain); put(3 * jj + 1, st); put(3 * jj + 2, (uint32_t)sum);
    bits(9 + 33 * jj, (uint32_t)sum);
    put(9 + 33 * jj + 32, sum >> 32);
  }
}

__global__ void k_scatter_cells(const CellDev *__restrict__ cells, u64 ncells, u64 *__restrict__ wires, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncells) return;
  const CellDev c = cells[i];
  wires[(u64)c.col * n + c.row] = c.value;
}

// One lane per PoseidonGate row: the permutation in its plain round form (constants, S-box, MDS), recording what enters every
// S-box that has a wire - plonky2 gates/poseidon.rs PoseidonGenerator::run_once; wire layout as in kernels_prover.hip
// q_poseidon_native.  A few thousand rows per light-client proof (the recursive verifier's Merkle paths, its Challenger and the
// sponge over the inner proof's public inputs): canonical arithmetic throughout, speed is irrelevant here.
__global__ void k_poseidon_gate_rows(const PoseidonRowDev *__restrict__ rows, u64 nrows, u64 *__restrict__ wires, u64 n, const u64 *__restrict__ rc) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  const PoseidonRowDev job = rows[i];
  u64 *W = wires + job.row;
  auto put = [&](u32 col, u64 v) { W[(u64)col * n] = v; };
  u64 s[12];
#pragma unroll
  for (int j = 0; j < 12; j++) { s[j] = gl_canon(job.in[j]); put(j, s[j]); }
  const bool swap = job.swap != 0;
  put(24, swap ? 1 : 0);
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u64 delta = swap ? gl_sub(s[j + 4], s[j]) : 0;
    put(25 + j, delta);
    const u64 l = gl_add(s[j], delta), r = gl_sub(s[j + 4], delta);
    s[j] = l; s[j + 4] = r;
  }
#pragma unroll 1
  for (int round = 0; round < POS_ROUNDS; round++) {
#pragma unroll
    for (int j = 0; j < 12; j++) s[j] = gl_add(s[j], rc[12 * round + j]);
    const bool full = round < POS_FULL_HALF || round >= POS_FULL_HALF + POS_PARTIAL;
    if (full) {
#pragma unroll
      for (int j = 0; j < 12; j++) {
        if (round >= 1 && round < POS_FULL_HALF) put(29 + 12 * (round - 1) + j, s[j]);
        if (round >= POS_FULL_HALF + POS_PARTIAL) put(87 + 12 * (round - POS_FULL_HALF - POS_PARTIAL) + j, s[j]);
        s[j] = gl_canon(pos_sbox(s[j]));
      }
    } else {
      put(65 + (round - POS_FULL_HALF), s[0]);
      s[0] = gl_canon(pos_sbox(s[0]));
    }
    pos_mds(s);
#pragma unroll
    for (int j = 0; j < 12; j++) s[j] = gl_canon(s[j]);
  }
#pragma unroll
  for (int j = 0; j < 12; j++) put(12 + j, s[j]);
}
void launch_poseidon_gate_rows(hipStream_t s, const PoseidonRowDev *rows, u64 nrows, u64 *wires, u64 n, const u64 *rc) {
  if (!nrows) return;
  hipLaunchKernelGGL(k_poseidon_gate_rows, dim3((unsigned)((nrows + 63) / 64)), dim3(64), 0, s, rows, nrows, wires, n, rc);
}

void launch_sha_jobs_level(hipStream_t s, const ShaJobDev *jobs, u32 first, u32 count, const uint32_t *words_in, uint32_t *rec) {
  if (!count) return;
  hipLaunchKernelGGL(k_sha_jobs_level, dim3((count + 63) / 64), dim3(64), 0, s, jobs, first, count, words_in, rec);
}
void launch_sha_fill_rows(hipStream_t s, const ShaJobDev *jobs, u32 njobs, const uint32_t *rec, u64 *wires, u64 n) {
  if (!njobs) return;
  u64 threads = (u64)njobs * SHA_ROWS;
  hipLaunchKernelGGL(k_sha_fill_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, jobs, njobs, rec, wires, n);
}
void launch_scatter_cells(hipStream_t s, const CellDev *cells, u64 ncells, u64 *wires, u64 n) {
  if (!ncells) return;
  hipLaunchKernelGGL(k_scatter_cells, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, s, cells, ncells, wires, n);
}

}  // namespace lcp2
