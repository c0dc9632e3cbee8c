/* ORACLE (test infrastructure, NOT product code): circuit description and proof layout
 * shared by plonk.c.  The struct layouts are the DATA FORMAT of include/lcp2.h
 * (lcp2_params, lcp2_gate) restated here so that the oracle does not include product headers.
 */
#ifndef ORACLE_PLONK_H
#define ORACLE_PLONK_H
#include "oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_FRI_LAYERS 8
typedef struct {
  uint32_t degree_bits, num_wires, num_routed_wires, num_constants, rate_bits, cap_height, num_challenges,
      quotient_degree_factor, proof_of_work_bits, num_query_rounds, num_fri_layers;
  uint32_t fri_arity_bits[ORC_MAX_FRI_LAYERS];
} orc_params;

typedef struct {
  uint32_t selector_index;  /* selector column of this gate's group */
  uint32_t selector_value;  /* value of that column on rows holding this gate */
  uint32_t group_start, group_end; /* selector values [start, end) share the column */
  uint32_t code_offset, code_len;  /* in instructions (2 words each) */
  uint32_t num_constraints;
  uint32_t flags;           /* ORC_GATE_EMIT_FORWARD: the program emits its constraints from the first to the last */
} orc_gate;
#define ORC_GATE_EMIT_FORWARD 1u

/* gate program instruction: word0 = op | dst << 8 | kind_a << 16 | kind_b << 20 ; word1 = idx_a | idx_b << 16 */
enum { ORC_OP_ADD = 0, ORC_OP_SUB = 1, ORC_OP_MUL = 2, ORC_OP_EMIT = 3,
       ORC_OP_XOR = 4,      /* dst = a + b - 2ab */
       ORC_OP_DBLADD = 5,   /* dst = 2a + b */
       ORC_OP_EMITBOOL = 6, /* emits a*a - a */
       ORC_OP_MULADD = 7,   /* dst = dst + a*b */
       ORC_OP_SBOX = 8,     /* dst = a^7 (Poseidon S-box) */
       ORC_OP_PMDS = 9 };   /* regs[dst..dst+12) = MDS * regs[idx_a..idx_a+12) + imm[idx_b..idx_b+12)  (Poseidon MDS layer + next constants) */
/* ORC_K_PI: element idx (< 4) of public_inputs_hash = hash_no_pad(public inputs), what plonky2's PublicInputGate compares with */
enum { ORC_K_REG = 0, ORC_K_WIRE = 1, ORC_K_CONST = 2, ORC_K_IMM = 3, ORC_K_PI = 4 };
#define ORC_MAX_REGS 64
#define ORC_UNUSED_SELECTOR 0xFFFFFFFFull

typedef struct orc_batch orc_batch;
typedef struct orc_circuit orc_circuit;

/* constants_sigmas: column-major [num_constants + num_routed][n] VALUES on the subgroup H (natural row order):
 * selectors first, then gate constants, then the sigma polynomials. */
orc_circuit *orc_circuit_new(const orc_params *p, const uint64_t *constants_sigmas, const uint64_t *k_is,
                             uint32_t num_selectors, const orc_gate *gates, uint32_t num_gates, const uint32_t *code,
                             size_t code_words, const uint64_t *imm, size_t num_imm, uint32_t num_public_inputs);
orc_circuit *orc_circuit_new_unbuilt(const orc_params *p, const uint64_t *constants_sigmas, const uint64_t *k_is,
                                     uint32_t num_selectors, const orc_gate *gates, uint32_t num_gates, const uint32_t *code,
                                     size_t code_words, const uint64_t *imm, size_t num_imm, uint32_t num_public_inputs);
/* verifier-only circuit from the digest and the constants/sigmas cap (no preprocessed values, no build()): orc_verify only */
orc_circuit *orc_verifier_new(const orc_params *p, const uint64_t *k_is, uint32_t num_selectors, const orc_gate *gates,
                              uint32_t num_gates, const uint32_t *code, size_t code_words, const uint64_t *imm, size_t num_imm,
                              uint32_t num_public_inputs, const uint64_t digest[4], const uint64_t *cap);
void orc_circuit_free(orc_circuit *c);
void orc_circuit_digest(const orc_circuit *c, uint64_t digest[4], uint64_t *cap /* 2^cap_height*4, nullable */);
size_t orc_proof_words(const orc_params *p);
/* wires: column-major [num_wires][n] witness values.  0 on success. */
int orc_prove(const orc_circuit *c, const uint64_t *wires, const uint64_t *public_inputs, uint64_t *proof);
/* 0 = accepted; otherwise a positive code naming the failed check */
int orc_verify(const orc_circuit *c, const uint64_t *proof, const uint64_t *public_inputs);
/* row-wise constraint check of a witness (debug aid): returns the number of violated (row, constraint) pairs,
 * first_bad (nullable) receives row and constraint index of the first one */
size_t orc_check_witness(const orc_circuit *c, const uint64_t *wires, const uint64_t *public_inputs, uint64_t first_bad[2]);
/* intermediate values of the last orc_prove on this circuit, for stage-by-stage parity tests */
typedef struct {
  uint64_t betas[4], gammas[4], alphas[4], zeta[2], fri_alpha[2], fri_betas[ORC_MAX_FRI_LAYERS][2];
  uint64_t pow_witness;
  uint64_t query_indices[64];
} orc_challenges;
void orc_last_challenges(const orc_circuit *c, orc_challenges *out);

#ifdef __cplusplus
}
#endif
#endif
