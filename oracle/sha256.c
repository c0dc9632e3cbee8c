/* ORACLE (test infrastructure, NOT product code).
 *
 * Native (out-of-circuit) statement of the SHA-256 work the reference's gadgets
 * constrain, following the four in-repo files:
 *   two_to_one_sha256 / compute_next_layer / MerkleTreeSha256Target
 *       /root/reference/eth-lc-plonky2/src/merkle_tree_gadget.rs:28-59
 *   VerifyMerkleProofTarget (left/right by index parity)     ibid. :61-87
 *   ssz_sync_committee (leaf packing, 2-leaf aggregate tree) src/sync_committee_pubkeys.rs:47-87
 *   ContractStateTarget (two height-2 trees)                 src/targets.rs:334-389
 *   BeaconBlockHeader (height-3 tree, 5 fields + 3 zero)     src/targets.rs:147-181
 *   SigningRoot (H(header_root || domain))                   src/targets.rs:121-145
 * Pinned by the reference's own KATs (tests/golden/sha256_kat.json, taken from
 * src/merkle_tree_gadget.rs:204-315, src/sync_committee_pubkeys.rs:107-622,
 * src/unit_tests.rs:44-212).
 * two_to_one_sha256(l, r) = SHA-256 of the 64-byte message l||r, i.e. two
 * compressions (data block + constant padding block).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static const uint32_t H0[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

static inline uint32_t ror(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

/* One compression.  If trace != NULL it receives the round-state trace used by
 * witness generation: 48 schedule words W[16..63] followed by 64 x (a_new, e_new). */
void orc_sha256_compress(uint32_t state[8], const uint32_t block[16], uint32_t *trace) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = block[i];
  for (int i = 16; i < 64; i++) {
    uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    if (trace) trace[i - 16] = w[i];
  }
  uint32_t a = state[0], b = state[1], c = state[2], d = state[3], e = state[4], f = state[5], g = state[6], h = state[7];
  for (int i = 0; i < 64; i++) {
    uint32_t S1 = ror(e, 6) ^ ror(e, 11) ^ ror(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = h + S1 + ch + K256[i] + w[i];
    uint32_t S0 = ror(a, 2) ^ ror(a, 13) ^ ror(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    if (trace) { trace[48 + 2 * i] = a; trace[48 + 2 * i + 1] = e; }
  }
  state[0] += a; state[1] += b; state[2] += c; state[3] += d;
  state[4] += e; state[5] += f; state[6] += g; state[7] += h;
}

static inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* two_to_one_sha256 (merkle_tree_gadget.rs:37): out = SHA256(left || right) */
void orc_sha256_two_to_one(const uint8_t left[32], const uint8_t right[32], uint8_t out[32]) {
  uint32_t st[8], blk[16];
  memcpy(st, H0, sizeof st);
  for (int i = 0; i < 8; i++) { blk[i] = be32(left + 4 * i); blk[8 + i] = be32(right + 4 * i); }
  orc_sha256_compress(st, blk, NULL);
  memset(blk, 0, sizeof blk);
  blk[0] = 0x80000000u; blk[15] = 512; /* padding block of a 64-byte message */
  orc_sha256_compress(st, blk, NULL);
  for (int i = 0; i < 8; i++) { out[4 * i] = st[i] >> 24; out[4 * i + 1] = st[i] >> 16; out[4 * i + 2] = st[i] >> 8; out[4 * i + 3] = st[i]; }
}

/* add_virtual_merkle_tree_sha256_target (merkle_tree_gadget.rs:42-59):
 * nodes[] receives every level: 2^h leaves, then 2^(h-1) ... then the root
 * ((2^(h+1) - 1) * 32 bytes); nodes may be NULL. */
void orc_sha256_merkle_root(const uint8_t *leaves, unsigned height, uint8_t root[32], uint8_t *nodes) {
  size_t n = (size_t)1 << height;
  uint8_t *buf = (uint8_t *)malloc(2 * n * 32);
  memcpy(buf, leaves, n * 32);
  uint8_t *cur = buf;
  for (unsigned l = 0; l < height; l++) {
    size_t m = n >> (l + 1);
    uint8_t *nxt = cur + (n >> l) * 32;
    for (size_t i = 0; i < m; i++) orc_sha256_two_to_one(cur + 64 * i, cur + 64 * i + 32, nxt + 32 * i);
    cur = nxt;
  }
  memcpy(root, cur, 32);
  if (nodes) memcpy(nodes, buf, (2 * n - 1) * 32);
  free(buf);
}

/* add_verify_merkle_proof_target (merkle_tree_gadget.rs:61-87) */
void orc_sha256_merkle_branch_root(const uint8_t leaf[32], const uint8_t *branch, unsigned height, size_t index, uint8_t root[32]) {
  uint8_t cur[32], nxt[32];
  memcpy(cur, leaf, 32);
  for (unsigned i = 0; i < height; i++) {
    if ((index & 1) == 0) orc_sha256_two_to_one(cur, branch + 32 * i, nxt);
    else orc_sha256_two_to_one(branch + 32 * i, cur, nxt);
    memcpy(cur, nxt, 32);
    index >>= 1;
  }
  memcpy(root, cur, 32);
}

/* ssz_sync_committee (sync_committee_pubkeys.rs:47-87): 512 pubkeys of 48 bytes
 * -> 1024 leaves (bytes 0..32 | bytes 32..48 + 16 zero), aggregate key -> 2
 * leaves, root = H(pubkeys_root, H(agg_leaf0, agg_leaf1)). */
void orc_ssz_sync_committee_leaves(const uint8_t *pubkeys /*[512][48]*/, uint8_t *leaves /*[1024][32]*/) {
  for (int i = 0; i < 512; i++) {
    memcpy(leaves + 64 * i, pubkeys + 48 * i, 32);
    memcpy(leaves + 64 * i + 32, pubkeys + 48 * i + 32, 16);
    memset(leaves + 64 * i + 48, 0, 16);
  }
}
void orc_ssz_sync_committee_root(const uint8_t *pubkeys, const uint8_t agg[48], uint8_t root[32]) {
  uint8_t *leaves = (uint8_t *)malloc(1024 * 32);
  uint8_t pk_root[32], agg_leaves[64], agg_root[32];
  orc_ssz_sync_committee_leaves(pubkeys, leaves);
  orc_sha256_merkle_root(leaves, 10, pk_root, NULL);
  memcpy(agg_leaves, agg, 48);
  memset(agg_leaves + 48, 0, 16);
  orc_sha256_two_to_one(agg_leaves, agg_leaves + 32, agg_root);
  orc_sha256_two_to_one(pk_root, agg_root, root);
  free(leaves);
}

static void u64_le_leaf(uint64_t v, uint8_t leaf[32]) {
  memset(leaf, 0, 32);
  for (int i = 0; i < 8; i++) leaf[i] = (uint8_t)(v >> (8 * i));
}

/* ContractState root (targets.rs:334-389, witness at unit_tests.rs:213-242):
 * root of [le(slot), header, sync_committee_i, sync_committee_ii]. */
void orc_contract_state_root(uint64_t slot, const uint8_t header[32], const uint8_t sc_i[32], const uint8_t sc_ii[32], uint8_t root[32]) {
  uint8_t leaves[4 * 32];
  u64_le_leaf(slot, leaves);
  memcpy(leaves + 32, header, 32); memcpy(leaves + 64, sc_i, 32); memcpy(leaves + 96, sc_ii, 32);
  orc_sha256_merkle_root(leaves, 2, root, NULL);
}

/* BeaconBlockHeader root (targets.rs:147-181; setter :685-707) */
void orc_beacon_header_root(uint64_t slot, uint64_t proposer, const uint8_t parent[32], const uint8_t state[32], const uint8_t body[32], uint8_t root[32]) {
  uint8_t leaves[8 * 32];
  memset(leaves, 0, sizeof leaves);
  u64_le_leaf(slot, leaves); u64_le_leaf(proposer, leaves + 32);
  memcpy(leaves + 64, parent, 32); memcpy(leaves + 96, state, 32); memcpy(leaves + 128, body, 32);
  orc_sha256_merkle_root(leaves, 3, root, NULL);
}
