/* ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the prover hot path of Electron-Labs/eth-lc-plonky2
 * (= plonky2 0.1.4 `prove()` + plonky2_crypto SHA-256 semantics, see
 * SURVEY.md section 8).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker.
 *
 * PARITY STATUS: SHA-256/SSZ results are pinned by the reference's own KATs.
 * Poseidon is pinned by the upstream permutation vectors.  Everything derived
 * from a whole proof (caps, challenges, openings, FRI) is "parity unpinned":
 * the reference holds no fixture for it and cannot be built here (no Rust).
 */
#ifndef ORACLE_H
#define ORACLE_H
#include "gl64.h"

#ifdef __cplusplus
extern "C" {
#endif

#define POSEIDON_WIDTH 12
#define POSEIDON_N_ROUNDS 30

/* ---- poseidon.c ---- */
const uint64_t *orc_poseidon_round_constants(void);
void orc_poseidon_permute(uint64_t s[12]);
void orc_poseidon_permute_batch(const uint64_t *in, uint64_t *out, size_t count);
void orc_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]);
void orc_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]);
void orc_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);

typedef struct {
  size_t nleaves;
  unsigned height, cap_height, nlevels;
  uint64_t *digests; /* level 0 (leaf digests) .. cap level, 4 u64 per node */
  uint64_t *cap;     /* points into digests: 2^cap_height nodes */
} orc_merkle;
orc_merkle *orc_merkle_build(const uint64_t *leaves, size_t nleaves, size_t leaf_len, unsigned cap_height);
void orc_merkle_free(orc_merkle *t);
void orc_merkle_prove(const orc_merkle *t, size_t index, uint64_t *siblings);
int orc_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t index, const uint64_t *siblings,
                      unsigned nsiblings, const uint64_t *cap);
/* convenience for ctypes: build, copy the cap out, free */
int orc_merkle_cap(const uint64_t *leaves, size_t nleaves, size_t leaf_len, unsigned cap_height, uint64_t *cap_out);

/* ---- sha256.c ---- */
void orc_sha256_compress(uint32_t state[8], const uint32_t block[16], uint32_t *trace);
void orc_sha256_two_to_one(const uint8_t left[32], const uint8_t right[32], uint8_t out[32]);
void orc_sha256_merkle_root(const uint8_t *leaves, unsigned height, uint8_t root[32], uint8_t *nodes);
void orc_sha256_merkle_branch_root(const uint8_t leaf[32], const uint8_t *branch, unsigned height, size_t index, uint8_t root[32]);
void orc_ssz_sync_committee_leaves(const uint8_t *pubkeys, uint8_t *leaves);
void orc_ssz_sync_committee_root(const uint8_t *pubkeys, const uint8_t agg[48], uint8_t root[32]);
void orc_contract_state_root(uint64_t slot, const uint8_t header[32], const uint8_t sc_i[32], const uint8_t sc_ii[32], uint8_t root[32]);
void orc_beacon_header_root(uint64_t slot, uint64_t proposer, const uint8_t parent[32], const uint8_t state[32], const uint8_t body[32], uint8_t root[32]);

/* ---- ntt.c ---- */
void orc_fft(uint64_t *a, size_t n);
void orc_ifft(uint64_t *a, size_t n);
void orc_coset_fft(uint64_t *a, size_t n, uint64_t shift);
void orc_coset_ifft(uint64_t *a, size_t n, uint64_t shift);
void orc_lde_coset_fft(const uint64_t *coeffs, size_t n, unsigned rate_bits, uint64_t shift, uint64_t *out);
void orc_ifft_batch(uint64_t *cols, size_t ncols, size_t n);
void orc_fft_batch(uint64_t *cols, size_t ncols, size_t n);
void orc_lde_batch(const uint64_t *coeffs, size_t ncols, size_t n, unsigned rate_bits, uint64_t shift, uint64_t *out);

/* ---- field helpers exported for ctypes-level tests ---- */
uint64_t orc_gl_mul(uint64_t a, uint64_t b);
uint64_t orc_gl_add(uint64_t a, uint64_t b);
uint64_t orc_gl_sub(uint64_t a, uint64_t b);
uint64_t orc_gl_inv(uint64_t a);
uint64_t orc_gl_pow(uint64_t a, uint64_t e);
uint64_t orc_gl_root_of_unity(unsigned k);
void orc_gl2_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);
void orc_gl2_inv(const uint64_t a[2], uint64_t out[2]);

#ifdef __cplusplus
}
#endif
#endif
