/* ORACLE (test infrastructure, NOT product code).
 *
 * Restates plonky2 0.1.4 (@666f3151, /root/reference/Cargo.lock:2347-2350)
 *   hash/poseidon.rs + poseidon_goldilocks.rs : Poseidon-12 permutation
 *   hash/hashing.rs                           : hash_n_to_m_no_pad, two_to_one
 *   hash/hash_types.rs / poseidon.rs          : hash_or_noop
 *   hash/merkle_tree.rs, merkle_proofs.rs     : MerkleTree::new, prove, verify
 * reached from the reference via PoseidonGoldilocksConfig
 * (/root/reference/eth-lc-plonky2/src/main.rs:75, src/unit_tests.rs:26).
 * Source is absent from the container: the algorithm below is the published
 * definition; the round constants are re-derived from
 * ChaCha8Rng::seed_from_u64(0) and pinned by the three upstream permutation
 * test vectors (tests/golden/poseidon_kat.json).
 * The permutation is the NAIVE round form on purpose (no fast-partial-round
 * refactoring) so that it is an independent check of the device kernels.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static uint64_t RC[POSEIDON_N_ROUNDS * POSEIDON_WIDTH];
static int rc_ready = 0;
static const uint64_t MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const uint64_t MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

/* ---- ChaCha8 keystream, rand_chacha layout (64-bit counter in words 12,13) ---- */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static inline uint32_t rotr32(uint32_t x, int r) { r &= 31; return r ? (x >> r) | (x << (32 - r)) : x; }
#define QR(a, b, c, d) \
  a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); \
  a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7)

static void chacha8_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]) {
  uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
  for (int i = 0; i < 8; i++) in[4 + i] = key[i];
  in[12] = (uint32_t)counter; in[13] = (uint32_t)(counter >> 32); in[14] = 0; in[15] = 0;
  uint32_t x[16];
  memcpy(x, in, sizeof x);
  for (int r = 0; r < 4; r++) { /* 4 double rounds = 8 rounds */
    QR(x[0], x[4], x[8], x[12]);  QR(x[1], x[5], x[9], x[13]);
    QR(x[2], x[6], x[10], x[14]); QR(x[3], x[7], x[11], x[15]);
    QR(x[0], x[5], x[10], x[15]); QR(x[1], x[6], x[11], x[12]);
    QR(x[2], x[7], x[8], x[13]);  QR(x[3], x[4], x[9], x[14]);
  }
  for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}

/* rand_core SeedableRng::seed_from_u64 (PCG32 expansion), then rand 0.8
 * uniform sampling of [0,p) (zone = 2^64 - 2^32), 360 accepted values. */
static void derive_round_constants(void) {
  uint32_t key[8];
  uint64_t st = 0;
  for (int i = 0; i < 8; i++) {
    st = st * 6364136223846793005ULL + 11634580027462260723ULL;
    uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27);
    key[i] = rotr32(xs, (int)(st >> 59));
  }
  uint32_t blk[16];
  uint64_t ctr = 0;
  int pos = 16, got = 0;
  const uint64_t zone = 0xFFFFFFFF00000000ULL;
  while (got < POSEIDON_N_ROUNDS * POSEIDON_WIDTH) {
    uint32_t w[2];
    for (int k = 0; k < 2; k++) {
      if (pos == 16) { chacha8_block(key, ctr++, blk); pos = 0; }
      w[k] = blk[pos++];
    }
    uint64_t v = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    gl_u128 m = (gl_u128)v * GL_P;
    if ((uint64_t)m <= zone) RC[got++] = (uint64_t)(m >> 64);
  }
  rc_ready = 1;
}

const uint64_t *orc_poseidon_round_constants(void) {
  if (!rc_ready) derive_round_constants();
  return RC;
}

static inline uint64_t sbox7(uint64_t x) {
  uint64_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
  return gl_mul(x3, x4);
}

static void mds_layer(uint64_t s[12]) {
  uint64_t out[12];
  for (int r = 0; r < 12; r++) {
    gl_u128 acc = 0; /* 12 terms * 2^64 * 41 fits easily */
    for (int i = 0; i < 12; i++) acc += (gl_u128)s[(i + r) % 12] * MDS_CIRC[i];
    acc += (gl_u128)s[r] * MDS_DIAG[r];
    out[r] = gl_reduce128(acc);
  }
  memcpy(s, out, sizeof out);
}

void orc_poseidon_permute(uint64_t s[12]) {
  if (!rc_ready) derive_round_constants();
  for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
  int round = 0;
  for (int r = 0; r < 4; r++, round++) {
    for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], RC[round * 12 + i]));
    mds_layer(s);
  }
  for (int r = 0; r < 22; r++, round++) {
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RC[round * 12 + i]);
    s[0] = sbox7(s[0]);
    mds_layer(s);
  }
  for (int r = 0; r < 4; r++, round++) {
    for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], RC[round * 12 + i]));
    mds_layer(s);
  }
}

void orc_poseidon_permute_batch(const uint64_t *in, uint64_t *out, size_t count) {
#pragma omp parallel for schedule(static)
  for (size_t k = 0; k < count; k++) {
    uint64_t s[12];
    memcpy(s, in + 12 * k, sizeof s);
    orc_poseidon_permute(s);
    memcpy(out + 12 * k, s, sizeof s);
  }
}

/* hash_n_to_m_no_pad with m = 4: overwrite-mode sponge, rate 8 */
void orc_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]) {
  uint64_t s[12] = {0};
  for (size_t off = 0; off < len; off += 8) {
    size_t c = len - off < 8 ? len - off : 8;
    for (size_t i = 0; i < c; i++) s[i] = gl_canon(in[off + i]);
    orc_poseidon_permute(s);
  }
  memcpy(out, s, 4 * sizeof(uint64_t));
}

void orc_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]) {
  if (len <= 4) {
    for (size_t i = 0; i < 4; i++) out[i] = i < len ? gl_canon(in[i]) : 0;
  } else {
    orc_hash_no_pad(in, len, out);
  }
}

void orc_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
  uint64_t s[12] = {0};
  for (int i = 0; i < 4; i++) { s[i] = gl_canon(l[i]); s[4 + i] = gl_canon(r[i]); }
  orc_poseidon_permute(s);
  memcpy(out, s, 4 * sizeof(uint64_t));
}

/* MerkleTree::new(leaves, cap_height).  Storage: level 0 = leaf digests,
 * level k has nleaves>>k nodes, stored back to back; the last stored level is
 * the cap (nleaves >> (height - cap_height) = 2^cap_height nodes). */
orc_merkle *orc_merkle_build(const uint64_t *leaves, size_t nleaves, size_t leaf_len, unsigned cap_height) {
  unsigned height = gl_log2(nleaves);
  if (((size_t)1 << height) != nleaves || cap_height > height) return NULL;
  orc_merkle *t = (orc_merkle *)calloc(1, sizeof *t);
  t->nleaves = nleaves; t->height = height; t->cap_height = cap_height;
  t->nlevels = height - cap_height + 1;
  size_t total = 0;
  for (unsigned k = 0; k < t->nlevels; k++) total += nleaves >> k;
  t->digests = (uint64_t *)malloc(total * 4 * sizeof(uint64_t));
  uint64_t *lvl = t->digests;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < nleaves; i++) orc_hash_or_noop(leaves + i * leaf_len, leaf_len, lvl + 4 * i);
  for (unsigned k = 1; k < t->nlevels; k++) {
    size_t m = nleaves >> k;
    uint64_t *nxt = lvl + 4 * (nleaves >> (k - 1));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++) orc_two_to_one(lvl + 8 * i, lvl + 8 * i + 4, nxt + 4 * i);
    lvl = nxt;
  }
  t->cap = lvl;
  return t;
}

void orc_merkle_free(orc_merkle *t) { if (t) { free(t->digests); free(t); } }

/* MerkleTree::prove(leaf_index): siblings bottom-up, height - cap_height of them */
void orc_merkle_prove(const orc_merkle *t, size_t index, uint64_t *siblings) {
  const uint64_t *lvl = t->digests;
  for (unsigned k = 0; k + 1 < t->nlevels; k++) {
    memcpy(siblings + 4 * k, lvl + 4 * (index ^ 1), 4 * sizeof(uint64_t));
    lvl += 4 * (t->nleaves >> k);
    index >>= 1;
  }
}

/* verify_merkle_proof_to_cap */
int orc_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t index, const uint64_t *siblings,
                      unsigned nsiblings, const uint64_t *cap) {
  uint64_t cur[4];
  orc_hash_or_noop(leaf, leaf_len, cur);
  for (unsigned k = 0; k < nsiblings; k++) {
    uint64_t nxt[4];
    if (index & 1) orc_two_to_one(siblings + 4 * k, cur, nxt);
    else orc_two_to_one(cur, siblings + 4 * k, nxt);
    memcpy(cur, nxt, sizeof cur);
    index >>= 1;
  }
  return memcmp(cur, cap + 4 * index, sizeof cur) == 0;
}
