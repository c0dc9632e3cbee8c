/*
 * lcp2.h -- C ABI of the MI355X (gfx950) Plonky2 prover backend for the
 * Ethereum light-client circuit of Electron-Labs/eth-lc-plonky2.
 *
 * What this replaces.  The reference (Rust) reaches its prover through exactly
 *     builder.build::<C>() -> data.prove(pw) -> data.verify(proof)
 * (eth-lc-plonky2/src/main.rs:226-233, src/unit_tests.rs:29-35) with
 * F = GoldilocksField, C = PoseidonGoldilocksConfig, D = 2 and
 * CircuitConfig::standard_recursion_config() (src/main.rs:74-79).  Everything
 * below that call lives in the un-vendored crates plonky2 0.1.4 / plonky2_field
 * 0.1.1 (@666f3151, Cargo.lock:2347-2350,2425-2427) and plonky2_crypto
 * (@3f713785, Cargo.lock:2388-2390).  Each entry point names the plonky2
 * function a Rust fork would forward to it (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - return 0 (LCP2_OK) or a negative lcp2_status; nothing throws or aborts
 *    across the ABI.  A witness that violates a gate constraint or a copy
 *    constraint gives LCP2_E_UNSAT from lcp2_prove / lcp2_quotient (checked on the
 *    device over the n rows of H before the quotient is formed), mirroring the
 *    `Err` of `prove()` that the reference's #[should_panic] tests rely on.
 *  - field elements are little-endian uint64_t Goldilocks values; inputs may be
 *    non-canonical (any value < 2^64), outputs are canonical.
 *  - extension elements are two uint64_t [c0, c1] (X^2 = 7).
 *  - every buffer argument is followed by an lcp2_mem saying where it lives.
 *    Device buffers must belong to the context's device.  Calls are
 *    asynchronous on the context's stream only for LCP2_MEM_DEVICE outputs;
 *    host outputs are complete on return.
 *  - a context is bound to one device and is NOT thread-safe; distinct
 *    contexts may be used concurrently.  No global mutable state.
 *  - there is no CPU fallback: without a usable HIP device every call that
 *    needs one returns LCP2_E_NODEVICE.
 */
#ifndef LCP2_H
#define LCP2_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCP2_ABI_VERSION 2

typedef enum {
  LCP2_OK = 0,
  LCP2_E_INVALID = -1,     /* bad argument / shape */
  LCP2_E_NODEVICE = -2,    /* no usable HIP device */
  LCP2_E_HIP = -3,         /* HIP runtime error (see lcp2_last_error) */
  LCP2_E_OOM = -4,
  LCP2_E_UNSAT = -5,       /* witness does not satisfy the circuit */
  LCP2_E_UNSUPPORTED = -6,
  LCP2_E_VERIFY = -7       /* proof rejected */
} lcp2_status;

typedef enum { LCP2_MEM_HOST = 0, LCP2_MEM_DEVICE = 1 } lcp2_mem;

/* CircuitConfig + FriConfig (plonky2 `CircuitConfig::standard_recursion_config()`,
 * reference call site src/main.rs:78) plus the per-circuit degree. */
#define LCP2_MAX_FRI_LAYERS 8
typedef struct {
  uint32_t degree_bits;            /* n = 2^degree_bits rows */
  uint32_t num_wires;              /* 135 */
  uint32_t num_routed_wires;       /* 80  */
  uint32_t num_constants;          /* constant columns incl. selectors */
  uint32_t rate_bits;              /* 3 */
  uint32_t cap_height;             /* 4 */
  uint32_t num_challenges;         /* 2 */
  uint32_t quotient_degree_factor; /* 8 */
  uint32_t proof_of_work_bits;     /* 16 */
  uint32_t num_query_rounds;       /* 28 */
  uint32_t num_fri_layers;         /* derived by lcp2_params_standard */
  uint32_t fri_arity_bits[LCP2_MAX_FRI_LAYERS];
} lcp2_params;

/* Fills `p` with standard_recursion_config() for a circuit of 2^degree_bits rows,
 * including the ConstantArityBits(4, 5) reduction schedule. */
int lcp2_params_standard(uint32_t degree_bits, uint32_t num_constants, lcp2_params *p);

typedef struct lcp2_ctx lcp2_ctx;
typedef struct lcp2_oracle lcp2_oracle; /* = plonky2 PolynomialBatch, device resident */

const char *lcp2_status_str(int status);
int lcp2_abi_version(void);
int lcp2_device_count(void);

/* device: HIP ordinal.  stream: a hipStream_t to run on (e.g. the caller's current stream) or NULL for a private,
 * NON-BLOCKING stream.  The legacy default stream has the handle NULL and therefore cannot be named: a caller that works on
 * it (PyTorch's default stream is stream 0) gets the private stream, which does not order itself against the default stream -
 * synchronise the producer of a device buffer before handing it over, and lcp2_ctx_sync before consuming results elsewhere.
 * Every entry point that returns host data has synchronised the context's stream when it returns. */
int lcp2_ctx_create(int device, void *stream, lcp2_ctx **out);
/* The same with flags.  LCP2_CTX_ORDER_WITH_DEFAULT_STREAM (stream must be NULL): the private stream is created as a BLOCKING
 * stream (hipStreamDefault), i.e. it orders itself against the legacy default stream in both directions the way any ordinary
 * stream does - the choice for a caller whose other work runs on stream 0 (PyTorch's default stream) and who does not want to
 * place explicit synchronisations around the library's calls.  0 = lcp2_ctx_create. */
#define LCP2_CTX_ORDER_WITH_DEFAULT_STREAM 1u
int lcp2_ctx_create_ex(int device, void *stream, uint32_t flags, lcp2_ctx **out);
void lcp2_ctx_destroy(lcp2_ctx *ctx);
int lcp2_ctx_sync(lcp2_ctx *ctx);
/* the hipStream_t the context runs on (the caller's, or the private one): for callers that order their own streams against it with
 * events - e.g. a collective on a communication stream that must wait for, and be waited for by, the library's kernels */
void *lcp2_ctx_stream(lcp2_ctx *ctx);
const char *lcp2_last_error(lcp2_ctx *ctx);

/* ------------------------------------------------------------------ primitives
 * (tests / micro-benchmarks; the same kernels the prover uses) */

/* PoseidonPermutation::permute on `count` independent width-12 states
 * (plonky2 hash/poseidon.rs).  in/out: [count][12]. */
int lcp2_poseidon_permute_batch(lcp2_ctx *ctx, const uint64_t *in, uint64_t *out, size_t count, lcp2_mem mem);

/* GoldilocksField arithmetic on `count` operands with the device functions every kernel uses (plonky2_field goldilocks_field.rs
 * `impl Add / Sub / Mul`, reduce128; csrc/gl64.hpp).  Operands may be any uint64_t, results are canonical; b is unused (NULL allowed)
 * by the unary operations.  For tests: random operands reach the borrow branch of the multiply's reduction with probability
 * 2^-32, crafted ones reach it at will. */
enum {
  LCP2_FIELD_MUL = 0,       /* a * b */
  LCP2_FIELD_POW7 = 1,      /* a^7: Poseidon's S-box as the hash kernels compute it */
  LCP2_FIELD_ADD = 2,       /* canonical add of the canonicalised operands */
  LCP2_FIELD_SUB = 3,       /* canonical subtract of the canonicalised operands */
  LCP2_FIELD_CANON = 4,     /* a mod p */
  LCP2_FIELD_ADD_LAZY = 5,  /* the lazy add (a: any value, b canonicalised), result canonicalised */
  LCP2_FIELD_SUB_LAZY = 6,  /* the lazy subtract */
  LCP2_FIELD_SHL = 16       /* LCP2_FIELD_SHL + k, k = 1..7: a * 2^(12 k), the NTT's register twiddles; + 8: the lazy a * 2^32;
                               + 9: a * (uint32_t)b, the multiply by a 32-bit constant */
};
int lcp2_field_op_batch(lcp2_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t count, uint32_t op, lcp2_mem mem);

/* MerkleTree::new(leaves, cap_height) (plonky2 hash/merkle_tree.rs): leaves is
 * row-major [nleaves][leaf_len]; cap receives 2^cap_height digests of 4 elements. */
int lcp2_merkle_cap(lcp2_ctx *ctx, const uint64_t *leaves, size_t nleaves, size_t leaf_len,
                    uint32_t cap_height, lcp2_mem mem, uint64_t *cap /* host */);

/* plonky2_field fft / ifft / coset_fft / coset_ifft on `ncols` polynomials of
 * 2^log_n elements, column-major [ncols][2^log_n], natural order in and out,
 * in place.  inverse = 0: values_i = sum_j c_j (shift w^i)^j ; inverse = 1 undoes it.
 * shift = 1 for the plain transforms. */
int lcp2_ntt_batch(lcp2_ctx *ctx, uint64_t *data, size_t ncols, uint32_t log_n, int inverse,
                   uint64_t shift, lcp2_mem mem);

/* PolynomialCoeffs::lde(rate_bits) + coset_fft(7) for every column, emitted in
 * Merkle LEAF ORDER (plonky2 applies `reverse_index_bits_in_place` before
 * hashing): out[col][i] = f_col(7 * w_{n<<r}^bitrev(i)), out is [ncols][n << rate_bits]. */
int lcp2_lde_batch(lcp2_ctx *ctx, const uint64_t *coeffs, uint64_t *out, size_t ncols, uint32_t log_n,
                   uint32_t rate_bits, lcp2_mem mem);

/* Native SHA-256 Merkle tree of the reference's `add_virtual_merkle_tree_sha256_target`
 * (src/merkle_tree_gadget.rs:42-59): leaves [2^height][32] bytes; nodes receives every
 * level, leaves first then 2^(height-1) ... 1 digests ((2^(height+1)-1)*32 bytes).
 * `trees` independent trees are processed in one launch chain (leaves and nodes
 * are arrays of that many trees).  round_trace (nullable) receives, per
 * two_to_one hash and per compression (2 per hash), 48 schedule words followed
 * by 64 (a, e) register pairs: the values the in-circuit witness needs. */
int lcp2_sha256_tree(lcp2_ctx *ctx, const uint8_t *leaves, uint32_t height, size_t trees,
                     uint8_t *nodes, uint32_t *round_trace, lcp2_mem mem);

/* ------------------------------------------------------------------ SHA-256 witness generation (K10)
 * Replaces the plonky2_crypto SHA-256 / U32 generators that plonky2's generate_partial_witness runs on one host
 * thread for every `two_to_one_sha256` (reference call sites src/merkle_tree_gadget.rs:37,57,77,79).  For circuits
 * laid out with this repository's SHA-256 rows (eth-lc-plonky2_amd/host, csrc/sha_layout.hpp: 310 rows per hash)
 * the rows are filled directly in the device-resident witness matrix `wires` [num_wires][n] (column-major).
 * jobs are sorted by dependency level: jobs [level_start[l], level_start[l+1]) only read digests of earlier levels.
 * in_src[i] >= 0: message word = words_in[in_src[i]] ; in_src[i] < 0: digest word (~in_src[i]) & 7 of job (~in_src[i]) >> 3.
 * digests (host, njobs * 8 words, nullable) receives every job's digest for the host-side generators that depend on it. */
typedef struct {
  uint32_t first_row;
  int32_t in_src[16];
} lcp2_sha_job;
typedef struct {
  uint32_t row, col;
  uint64_t value;
} lcp2_cell;
int lcp2_sha256_witness(lcp2_ctx *ctx, const lcp2_sha_job *jobs, size_t njobs, const uint32_t *level_start, uint32_t nlevels,
                        const uint32_t *words_in, size_t nwords, uint64_t *wires, uint64_t n, uint32_t *digests);
/* wires[col][row] = value for a list of cells (the non-SHA rows: constants, arithmetic glue, public inputs) */
int lcp2_scatter_cells(lcp2_ctx *ctx, const lcp2_cell *cells, size_t ncells, uint64_t *wires, uint64_t n);
/* PoseidonGate rows generated on the device (plonky2 gates/poseidon.rs PoseidonGenerator::run_once; the reference's circuit holds
 * thousands of them inside verify_proof, eth-lc-plonky2/src/targets.rs:468-470): one job per row = the 12 input wires and the swap
 * flag; the kernel writes EVERY wire of the row (inputs 0..12, outputs 12..24, swap 24, delta 25..29 and the S-box inputs 29..135)
 * into the column-major witness matrix.  The host generator then only needs the 12 outputs (a plain permutation) for the
 * generators downstream. */
typedef struct {
  uint32_t row;
  uint32_t swap;     /* 0 or 1 */
  uint64_t in[12];   /* any u64 */
} lcp2_poseidon_row;
int lcp2_poseidon_gate_rows(lcp2_ctx *ctx, const lcp2_poseidon_row *rows, size_t nrows, uint64_t *wires, uint64_t n);

/* device buffers for callers that keep the witness resident in HBM */
int lcp2_buffer_alloc(lcp2_ctx *ctx, size_t bytes, void **dev);
int lcp2_buffer_free(lcp2_ctx *ctx, void *dev);
int lcp2_buffer_zero(lcp2_ctx *ctx, void *dev, size_t bytes);
int lcp2_buffer_read(lcp2_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
int lcp2_buffer_write(lcp2_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int lcp2_buffer_copy(lcp2_ctx *ctx, void *dev_dst, const void *dev_src, size_t bytes);   /* device to device, on the context's stream */
/* `height` (< 65536) runs of `width` bytes, `src_pitch` / `dst_pitch` bytes apart (device to device; everything a multiple of 8):
 * row blocks out of / into whole columns */
int lcp2_buffer_copy_2d(lcp2_ctx *ctx, void *dev_dst, size_t dst_pitch, const void *dev_src, size_t src_pitch, size_t width, size_t height);

/* ------------------------------------------------------------------ polynomial commitments
 * PolynomialBatch::from_values / from_coeffs (plonky2 fri/oracle.rs): ifft,
 * lde x 2^rate_bits on the coset 7H, leaf hashing, Merkle tree with cap.
 * cols: column-major [ncols][2^log_n].  The oracle keeps coefficients, LDE and
 * all digests in HBM.  cap: host, 2^cap_height * 4 elements (nullable). */
int lcp2_commit_values(lcp2_ctx *ctx, const uint64_t *cols, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                       uint32_t cap_height, lcp2_mem mem, lcp2_oracle **out, uint64_t *cap);
int lcp2_commit_coeffs(lcp2_ctx *ctx, const uint64_t *coeffs, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                       uint32_t cap_height, lcp2_mem mem, lcp2_oracle **out, uint64_t *cap);
/* Coset-sharded commitment for one rank of a multi-GPU proof.  The LDE of size n << rate_bits is
 * 2^rate_bits coset transforms of size n; in Merkle leaf order coset r is the contiguous LEAF BLOCK bitrev(r), and with
 * cap_height >= rate_bits every block is 2^(cap_height - rate_bits) whole cap subtrees.  A rank therefore takes the
 * coefficients of ALL columns (after the all-gather of the column-sharded iNTT outputs), computes only its blocks
 * [block_first, block_first + block_count) (an aligned power of two), hashes its own leaves and builds its own
 * subtrees with no cross-GPU hashing; the global cap is the concatenation of the ranks' cap_part arrays in block
 * order.  cap_part: host, block_count * 2^(cap_height - rate_bits) * 4 elements.  The returned oracle answers
 * lcp2_oracle_open for LOCAL leaf indices (global leaf index - block_first * n), siblings up to the local cap. */
int lcp2_commit_cosets(lcp2_ctx *ctx, const uint64_t *coeffs, size_t ncols, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height,
                       uint32_t block_first, uint32_t block_count, lcp2_mem mem, lcp2_oracle **out, uint64_t *cap_part);
void lcp2_oracle_destroy(lcp2_oracle *o);
/* MerkleTree::prove + leaf lookup for `k` leaf indices (fri_prover_query_round):
 * leaves [k][ncols], siblings [k][log2(nleaves) - cap_height][4]; host buffers. */
int lcp2_oracle_open(lcp2_oracle *o, const uint64_t *indices, size_t k, uint64_t *leaves, uint64_t *siblings);
/* copies for tests: coefficients [ncols][n] and LDE [ncols][n << rate_bits] (leaf order), host buffers */
int lcp2_oracle_read(lcp2_oracle *o, uint64_t *coeffs /* nullable */, uint64_t *lde /* nullable */);

/* ------------------------------------------------------------------ circuit, prove, verify
 * The boundary the reference calls (src/main.rs:226-233, src/unit_tests.rs:29-35):
 *     let data = builder.build::<C>();          -> lcp2_circuit_create
 *     let proof = data.prove(pw).unwrap();      -> lcp2_prove
 *     data.verify(proof)                        -> lcp2_verify
 * A circuit is what build() produces: preprocessed polynomials (selector and
 * constant columns, sigma polynomials) and the gate set.  plonky2's gate types
 * are Rust objects that cannot cross a C ABI, so each gate type is described by a
 * constraint program ("gate program") that the quotient kernel (K6) and the
 * verifier interpret:
 *     instruction = 2 words:  w0 = op | dst << 8 | kind_a << 16 | kind_b << 20,  w1 = idx_a | idx_b << 16
 *     op   0 ADD  1 SUB  2 MUL (dst <- a op b)   3 EMIT (the next constraint of the gate is a)
 *          4 XOR (dst <- a + b - 2ab)  5 DBLADD (dst <- 2a + b)  6 EMITBOOL (EMIT of a*a - a)  7 MULADD (dst <- dst + a*b)
 *          8 SBOX (dst <- a^7, the Poseidon S-box)
 *          9 PMDS (Poseidon MDS layer on a window of 12 registers, with the constants that follow it:
 *                  reg[dst + r] <- sum_i reg[idx_a + (i + r) % 12] * MDS_CIRC[i] + reg[idx_a + r] * MDS_DIAG[r] + imm[idx_b + r],
 *                  r < 12; kind_a = REG, kind_b = IMM; the two windows may be the same)
 *     kind 0 REG  1 WIRE (local wire)  2 CONST (gate constant, after the selector columns)
 *          3 IMM (imm[idx])  4 PI (public_inputs_hash[idx], idx < 4: what plonky2's PublicInputGate compares its wires with)
 * A gate's constraints c_0 .. c_{m-1} enter the quotient as sum_i alpha^i c_i.  By default a program lists them from the
 * LAST to the FIRST (EMIT is then a Horner step acc <- acc * alpha + a); with LCP2_GATE_EMIT_FORWARD in `flags` it lists
 * them from the first to the last (gates such as PoseidonGate whose constraints fall out of one forward pass).
 * Selectors follow plonky2 gates/selectors.rs: gate g is active on rows where
 * constants[selector_index] == selector_value; its filter is
 *     prod_{j in [group_start, group_end), j != selector_value} (j - s) * (num_selectors > 1 ? (2^32 - 1 - s) : 1). */
enum { LCP2_OP_ADD = 0, LCP2_OP_SUB = 1, LCP2_OP_MUL = 2, LCP2_OP_EMIT = 3, LCP2_OP_XOR = 4, LCP2_OP_DBLADD = 5,
       LCP2_OP_EMITBOOL = 6, LCP2_OP_MULADD = 7, LCP2_OP_SBOX = 8, LCP2_OP_PMDS = 9 };
#define LCP2_GATE_EMIT_FORWARD 1u
/* Bits 8..15 of `flags`: the caller's claim that the program is one of plonky2's gates in its standard wire layout, for which
 * the library has a native device evaluator (same constraints, no interpretation).  The claim is CHECKED at
 * lcp2_circuit_create: program and native evaluator must agree on random wire values, otherwise LCP2_E_INVALID.  The
 * verifier always interprets the program.  0 = none. */
#define LCP2_GATE_NATIVE_MASK 0xFF00u
#define LCP2_GATE_NATIVE_POSEIDON 0x0100u   /* PoseidonGate, gates/poseidon.rs: 135 wires, 123 constraints, EMIT_FORWARD order */
#define LCP2_GATE_NATIVE_ARITHMETIC 0x0200u /* ArithmeticGate { num_ops = num_constraints }: wires 4k..4k+3, constants 0 and 1 */
#define LCP2_GATE_NATIVE_BASE_SUM2 0x0300u  /* BaseSumGate<2> { num_limbs = num_constraints - 1 }: wire 0 = sum, wires 1.. = bits */
/* straight-line device forms generated offline from gate programs (tools/gen/, csrc/generated_gates.hpp): program k of that file */
#define LCP2_GATE_NATIVE_GENERATED(k) (0x8000u | ((uint32_t)(k) << 8))
typedef struct {
  uint32_t selector_index, selector_value, group_start, group_end;
  uint32_t code_offset, code_len; /* in instructions */
  uint32_t num_constraints;
  uint32_t flags;                 /* LCP2_GATE_EMIT_FORWARD | LCP2_GATE_NATIVE_* */
} lcp2_gate;

typedef struct {
  lcp2_params params;
  /* VALUES on the subgroup H, natural row order, column-major [num_constants + num_routed_wires][n]:
   * selector columns, gate-constant columns, then sigma_j(w^i) = k_{j'} w^{i'} of the copy-constraint permutation */
  const uint64_t *constants_sigmas;
  lcp2_mem constants_sigmas_mem;
  const uint64_t *k_is;            /* host, [num_routed_wires] coset shifts of the permutation argument */
  uint32_t num_selectors;
  uint32_t num_gates;
  const lcp2_gate *gates;          /* host */
  const uint32_t *code;            /* host, 2 words per instruction */
  size_t code_words;
  const uint64_t *imm;             /* host */
  size_t num_imm;
  uint32_t num_public_inputs;      /* length of the public-input vector; the circuit binds it through its hash (kind PI) */
  uint32_t num_regs;               /* registers the programs use (<= 64) */
} lcp2_circuit_desc;

typedef struct lcp2_circuit lcp2_circuit; /* = CircuitData: prover_only + verifier_only + common */

/* build(): uploads the description, commits constants_sigmas (PolynomialBatch::from_values),
 * derives the circuit digest = hash_no_pad(constants_sigmas_cap || hash_pad(domain separator = []) || degree_bits)
 * (plonky2 circuit_builder.rs::build) and allocates the per-proof workspace in HBM. */
int lcp2_circuit_create(lcp2_ctx *ctx, const lcp2_circuit_desc *desc, lcp2_circuit **out);
void lcp2_circuit_destroy(lcp2_circuit *c);
/* circuit_digest (4 elements) and constants_sigmas_cap (2^cap_height * 4), host buffers, cap nullable */
int lcp2_circuit_digest(const lcp2_circuit *c, uint64_t digest[4], uint64_t *cap);

/* Size in uint64_t words of a proof (layout documented in DESIGN.md, same field order as plonky2's
 * ProofWithPublicInputs: wires_cap, plonk_zs_partial_products_cap, quotient_polys_cap, openings,
 * opening_proof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness }). */
size_t lcp2_proof_words(const lcp2_params *p);

/* Word offsets of every field of the flat proof (what a caller needs to map it onto plonky2's Proof / FriProof structs, and
 * what the recursive verifier gadget of the host layer walks).  Openings are extension elements (2 words each), caps are
 * 4 << cap_height words.  Query round q starts at queries + q * query_words; inside it, initial-tree opening o (0 constants
 * and sigmas, 1 wires, 2 Zs and partial products, 3 quotient chunks) is q_init_cols[o] leaf elements at q_init_off[o] followed
 * by q_init_sib sibling digests; FRI layer l is 2 << fri_arity_bits[l] words of evaluations at q_step_off[l] followed by
 * q_step_sib[l] sibling digests. */
typedef struct {
  uint64_t cap_words, wires_cap, zs_cap, quot_cap;
  uint64_t op_constants, op_sigmas, op_wires, op_zs, op_zs_next, op_partial_products, op_quotient;
  uint64_t fri_caps, queries, query_words;
  uint64_t q_init_off[4], q_init_cols[4], q_init_sib;
  uint64_t q_step_off[LCP2_MAX_FRI_LAYERS], q_step_sib[LCP2_MAX_FRI_LAYERS];
  uint64_t final_poly, final_len, pow_witness, total;
} lcp2_proof_layout;
int lcp2_proof_layout_of(const lcp2_params *p, lcp2_proof_layout *out);

/* data.prove(pw): wires is the full witness (generate_partial_witness output), column-major
 * [num_wires][n]; public_inputs (num_public_inputs elements, must equal the circuit's count) and proof
 * (proof_words words, must equal lcp2_proof_words) are host buffers.  The proof-of-work witness is the
 * smallest valid one (plonky2 searches with a nondeterministic find_any).  LCP2_E_UNSAT: the witness violates
 * a gate or a copy constraint (nothing useful is in `proof`). */
int lcp2_prove(lcp2_circuit *c, const uint64_t *wires, lcp2_mem wires_mem, const uint64_t *public_inputs, size_t num_public_inputs,
               uint64_t *proof, size_t proof_words);

/* ---- a HOST witness without its upload on the critical path.  data.prove(pw) of a fork that runs plonky2's own generators ends up with
 * the full witness in host memory (4.5 GB at n = 2^22); lcp2_prove(.., LCP2_MEM_HOST, ..) copies it first and proves afterwards.  For a
 * sequence of proofs (BASELINE configs[4]: a batch of updates) the copy of witness i + 1 runs while proof i is computed:
 *   lcp2_host_register     pins a caller-owned host buffer (a Rust Vec's storage) so that the copy is a plain DMA; optional, but a
 *                          pageable buffer is staged by the runtime and lcp2_witness_stage then blocks for the whole copy
 *   lcp2_witness_stage     starts the upload of `wires` ([num_wires][n], any u64 values) into staging slot 0 or 1 of the circuit, on a
 *                          copy stream of the context; returns at once.  The host buffer must stay unchanged until the
 *                          lcp2_prove_staged of that slot has returned.  A slot must not be staged again before its proof returned.
 *   lcp2_prove_staged      lcp2_prove on the slot's device copy (the context's stream waits for the upload, the host does not)
 * The two slots are allocated at the first use (2 x num_wires x n x 8 bytes of HBM). */
int lcp2_host_register(lcp2_ctx *ctx, void *host, size_t bytes);
int lcp2_host_unregister(lcp2_ctx *ctx, void *host);
int lcp2_witness_stage(lcp2_circuit *c, const uint64_t *wires, uint32_t slot);
int lcp2_prove_staged(lcp2_circuit *c, uint32_t slot, const uint64_t *public_inputs, size_t num_public_inputs, uint64_t *proof, size_t proof_words);

/* ---- the seams inside data.prove() a plonky2 fork binds one by one; lcp2_prove is exactly their
 * composition under the Fiat-Shamir transcript, and the caller keeps its own Challenger in between.  The commitments
 * stay on the device inside the circuit handle; the calls must come in this order (LCP2_E_INVALID otherwise).
 *   lcp2_commit_wires  PolynomialBatch::from_values(wires)                        -> wires cap             (K1-K4)
 *   lcp2_perm_zs       wires_permutation_partial_products_and_zs + from_values    -> Z/partial-product cap (K5, K1-K4)
 *   lcp2_quotient      compute_quotient_polys + from_coeffs                       -> quotient cap          (K6, K1-K4)
 *   lcp2_fri_open      OpeningSet::new + PolynomialBatch::prove_openings           -> openings + FriProof   (K7-K9, a13)
 * betas/gammas/alphas: num_challenges base-field elements each; caps: 4 << cap_height words;
 * public_inputs_hash: the 4 elements of hash_no_pad(public inputs), as compute_quotient_polys takes it.
 * LIFETIME of a device witness: with wires_mem = LCP2_MEM_DEVICE the library keeps the caller's pointer, not a copy - the
 * permutation argument (lcp2_perm_zs) and the witness check behind LCP2_E_UNSAT (lcp2_quotient / lcp2_quotient_values) read the
 * values again.  The buffer must stay valid and unchanged until lcp2_quotient[_values] has returned (lcp2_prove: until it
 * returns).  A host witness is copied by lcp2_commit_wires and may be released as soon as that call returns.  The values may
 * be non-canonical (any u64) in either case. */
int lcp2_commit_wires(lcp2_circuit *c, const uint64_t *wires, lcp2_mem wires_mem, uint64_t *cap);
int lcp2_perm_zs(lcp2_circuit *c, const uint64_t *betas, const uint64_t *gammas, uint64_t *cap);
int lcp2_quotient(lcp2_circuit *c, const uint64_t *alphas, const uint64_t public_inputs_hash[4], uint64_t *cap);
/* plonky2's Challenger { sponge_state, input_buffer, output_buffer } (iop/challenger.rs), by value */
typedef struct {
  uint64_t sponge[12];
  uint64_t input[8];
  uint64_t output[8];
  uint32_t input_len, output_len;
} lcp2_challenger;
/* ch: state after observing the quotient cap and drawing zeta; updated to the state after the query indices were
 * drawn.  proof: a buffer of lcp2_proof_words(); words from the openings to the end are written (the three caps in
 * front of them are the caller's). */
int lcp2_fri_open(lcp2_circuit *c, const uint64_t zeta[2], lcp2_challenger *ch, uint64_t *proof);

/* lcp2_fri_open in its three phases (lcp2_fri_open is their composition, one code path), for callers that need the points
 * between them - a coset-sharded proof exchanges shares there:
 *   lcp2_fri_open_begin    OpeningSet::new: the openings at zeta and g*zeta                    -> LCP2_SECTION_OPENINGS
 *   lcp2_fri_open_commit   observes the openings, draws alpha, composes the final polynomial of the batch and commits
 *                          FRI layer 0 (the 8n-point LDE, its leaves and Merkle levels)         -> LCP2_SECTION_FRI_CAP0
 *   lcp2_fri_open_finish   the remaining FRI layers, final polynomial, proof of work, query answers; ch (nullable): the
 *                          challenger state after the query indices
 * lcp2_proof_section gives the word range of a section inside the proof array. */
enum { LCP2_SECTION_OPENINGS = 0, LCP2_SECTION_FRI_CAP0 = 1, LCP2_SECTION_AFTER_CAPS = 2 };
int lcp2_fri_open_begin(lcp2_circuit *c, const uint64_t zeta[2], const lcp2_challenger *ch, uint64_t *proof);
int lcp2_fri_open_commit(lcp2_circuit *c, uint64_t *proof);
int lcp2_fri_open_finish(lcp2_circuit *c, lcp2_challenger *ch, uint64_t *proof);
int lcp2_proof_section(const lcp2_circuit *c, int section, size_t *first_word, size_t *num_words);

/* host-side transcript helpers for callers that do not bring their own Challenger / PoseidonHash (no device work) */
void lcp2_challenger_init(lcp2_challenger *ch);
int lcp2_challenger_observe(lcp2_challenger *ch, const uint64_t *values, size_t count);
int lcp2_challenger_get(lcp2_challenger *ch, uint64_t *out, size_t count);   /* challenges pop from the back of the output buffer */
int lcp2_hash_no_pad(const uint64_t *values, size_t count, uint64_t out[4]); /* PoseidonHash::hash_no_pad (public-input hash) */

/* ---- one proof sharded over the GPUs of a node by LDE coset (BASELINE configs[3]).
 * A sharded circuit holds the leaf blocks [block_first, block_first + block_count) of every LDE and Merkle tree
 * (block_count a power of two dividing 2^rate_bits, block_first aligned to it, cap_height >= rate_bits so that a block is
 * whole cap subtrees: no cross-GPU hashing).  Every rank runs the same call sequence as the seams above; what a rank returns
 * is its SHARE of the result - its own cap entries / openings / query answers at their global position, zeros elsewhere; the
 * proof words every rank holds identically come from the rank holding block 0 only - so one SUM all-reduce (uint64 wrap-around;
 * RCCL has no bitwise reductions) of a share assembles the result.  The bulk exchanges are (1) the witness: the ranks may hold
 * column shards, all-gather the values, transform their own columns and all-gather the coefficients (lcp2_commit_wires_coeffs);
 * (2) the quotient (num_challenges * 8n words): each rank fills its blocks of lcp2_quotient_buffer - per challenge
 * plane they are one contiguous run at offset block_first * n, in rank order - so an in-place all-gather of each plane
 * completes the buffer (a SUM all-reduce works too: the rest is zeros); then every rank calls lcp2_quotient_commit.  (What a
 * block holds is the library's business: with rate_bits <= 3 it is the interpolant of the quotient on the block's coset - a rank
 * transforms only its own cosets, and lcp2_quotient_commit combines the interpolants into the quotient chunks.)
 * In the opening stage a rank evaluates its share of the columns of every batch (it holds all coefficients) and commits its
 * own leaf blocks of FRI layer 0; folding is done in coefficient form, so no FRI values cross the ranks, and the smaller
 * layers are computed by every rank.
 * Order per proof:
 *   lcp2_commit_wires[_coeffs] -> sum caps -> lcp2_perm_zs -> sum caps -> lcp2_quotient_values -> all-gather planes ->
 *   lcp2_quotient_commit -> sum caps -> lcp2_fri_open_begin -> sum LCP2_SECTION_OPENINGS -> lcp2_fri_open_commit ->
 *   sum LCP2_SECTION_FRI_CAP0 -> lcp2_fri_open_finish -> sum LCP2_SECTION_AFTER_CAPS.
 * At build: lcp2_circuit_create_sharded, lcp2_circuit_digest (cap share; digest not valid yet), sum the cap,
 * lcp2_circuit_set_constants_cap.  lcp2_prove / lcp2_quotient / lcp2_fri_open refuse a sharded circuit. */
int lcp2_circuit_create_sharded(lcp2_ctx *ctx, const lcp2_circuit_desc *desc, uint32_t block_first, uint32_t block_count,
                                lcp2_circuit **out);
int lcp2_circuit_set_constants_cap(lcp2_circuit *c, const uint64_t *cap);
/* lcp2_commit_wires with the iNTT already done: `wires` = witness values, `coeffs` = their coefficients, both device, column-major
 * [num_wires][n].  A sharded proof runs the iNTT polynomial-parallel (rank g transforms its column shard with lcp2_ntt_batch)
 * and all-gathers the coefficients over RCCL (the "column transpose" of the commitment); the values are still needed
 * for the permutation argument and the LCP2_E_UNSAT check.  The coefficients are copied into the circuit's oracle. */
int lcp2_commit_wires_coeffs(lcp2_circuit *c, const uint64_t *wires, const uint64_t *coeffs, uint64_t *cap);
/* Row exchange form of the same proof: the witness VALUES are needed only by the permutation argument (K5) and the gate check,
 * both row-wise, so rank r of `world` = 2^rate_bits / block_count takes just the rows [r * n / world, (r + 1) * n / world) of
 * every column - an all-to-all of row blocks out of the column shards (1 / world of the witness per rank) instead of the
 * all-gather of all values - and the ranks exchange what K5 makes of them:
 *   lcp2_commit_wires_rows      wire_rows: device, [num_wires][n / world]; coeffs as in lcp2_commit_wires_coeffs -> cap share
 *   lcp2_perm_zs_rows_begin     the quotient chunks of the rank's rows and their running product inside the block;
 *                               block_products[world][num_challenges]: this rank's entry, zeros elsewhere (a share to be summed)
 *   lcp2_perm_zs_rows_finish    block_products: the sum of the shares.  Z and the partial products of the rank's rows, times the
 *                               product of the blocks before it, go into the rank's slot of the exchange buffer
 *                               [world][num_challenges * (1 + npp)][n / world] (*device_ptr, *words in total): an in-place
 *                               all-gather completes it.  LCP2_E_UNSAT (on every rank alike) if the product over all blocks is not 1.
 *   lcp2_perm_zs_commit         iNTT / LDE / Merkle tree of the completed buffer -> cap share
 * (in this order, once each per proof: anything else is LCP2_E_INVALID)
 * lcp2_quotient_values then checks the gates on the rank's rows only: a caller must exchange the status (a rank that got
 * LCP2_E_UNSAT stops, and so must the others) before the next collective. */
int lcp2_commit_wires_rows(lcp2_circuit *c, const uint64_t *wire_rows, const uint64_t *coeffs, uint64_t *cap);
/* lcp2_commit_wires_rows with the coefficient exchange OVERLAPPED (round 4).  The sponge of a leaf absorbs the columns in order, 8 per
 * permutation, so the commitment can proceed chunk by chunk - coset LDE of the chunk's columns, absorption into a persistent 12-word
 * state per leaf - while later chunks are still crossing the fabric (eth-lc-plonky2_amd/parallel.py: chunks of 8 columns, one or a few per
 * rank, gathered on a second stream):
 *   lcp2_commit_wires_rows_begin    wire_rows as above; no coefficients yet
 *   lcp2_commit_wires_chunk         coeffs: device, the coefficient columns [first_col, first_col + ncols) as [ncols][n].  Chunks come in
 *                                   column order; first_col is a multiple of 8, ncols a multiple of 8 unless the chunk ends at num_wires
 *   lcp2_commit_wires_rows_finish   after the chunk that ends at num_wires: the Merkle levels -> cap share
 * The result is the commitment lcp2_commit_wires_rows makes (same coefficients, LDE, digests, cap). */
int lcp2_commit_wires_rows_begin(lcp2_circuit *c, const uint64_t *wire_rows);
int lcp2_commit_wires_chunk(lcp2_circuit *c, const uint64_t *coeffs, uint32_t first_col, uint32_t ncols);
int lcp2_commit_wires_rows_finish(lcp2_circuit *c, uint64_t *cap);
int lcp2_perm_zs_rows_begin(lcp2_circuit *c, const uint64_t *betas, const uint64_t *gammas, uint64_t *block_products);
int lcp2_perm_zs_rows_finish(lcp2_circuit *c, const uint64_t *block_products, uint64_t **device_ptr, size_t *words);
int lcp2_perm_zs_commit(lcp2_circuit *c, uint64_t *cap);
int lcp2_quotient_values(lcp2_circuit *c, const uint64_t *alphas, const uint64_t public_inputs_hash[4]);
int lcp2_quotient_buffer(lcp2_circuit *c, uint64_t **device_ptr, size_t *words);
int lcp2_quotient_commit(lcp2_circuit *c, uint64_t *cap);

/* data.verify(proof): host only (no device work).  proof_words / num_public_inputs are the lengths of the two
 * buffers: anything but lcp2_proof_words() / the circuit's public-input count is LCP2_E_INVALID (a truncated proof is
 * never read).  LCP2_OK or LCP2_E_VERIFY; *failed_check (nullable): 1 encoding, 2 proof of work, 3 vanishing identity,
 * 4 initial Merkle proof, 5 FRI consistency, 6 FRI layer Merkle proof, 7 final polynomial. */
int lcp2_verify(const lcp2_circuit *c, const uint64_t *proof, size_t proof_words, const uint64_t *public_inputs,
                size_t num_public_inputs, int *failed_check);

/* ---- byte serialisation of a proof: plonky2 0.1.4 util/serialization.rs `write_proof_with_public_inputs`
 * ([RECALL] of the published format; the reference holds no serialised proof - src/main.rs:230-233 moves the object - so
 * this layout is PARITY UNPINNED).  Little-endian throughout: a field element is its canonical u64, an extension element two
 * of them, a hash four; Merkle caps and opening vectors carry no length (the reader knows them from CommonCircuitData);
 * every MerkleProof is a u8 sibling count followed by the siblings; field order: wires_cap, plonk_zs_partial_products_cap,
 * quotient_polys_cap, openings { constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys },
 * opening_proof { commit_phase_merkle_caps, query_round_proofs { initial_trees_proof { (leaf, proof) x 4 }, steps { evals,
 * proof } }, final_poly, pow_witness }, then the public inputs, preceded by their count as a u64 when
 * LCP2_SER_PUBLIC_INPUT_COUNT is set (later 0.1.x snapshots write it, earlier ones do not).
 * lcp2_proof_bytes: the byte length; lcp2_proof_to_bytes / lcp2_proof_from_bytes: LCP2_E_INVALID on a length mismatch, a
 * non-canonical element or a wrong sibling count (nothing is read past `len`). */
#define LCP2_SER_PUBLIC_INPUT_COUNT 1u
size_t lcp2_proof_bytes(const lcp2_params *p, size_t num_public_inputs, uint32_t flags);
int lcp2_proof_to_bytes(const lcp2_params *p, const uint64_t *proof, size_t proof_words, const uint64_t *public_inputs,
                        size_t num_public_inputs, uint32_t flags, uint8_t *out, size_t out_len);
int lcp2_proof_from_bytes(const lcp2_params *p, const uint8_t *bytes, size_t len, uint32_t flags, uint64_t *proof, size_t proof_words,
                          uint64_t *public_inputs, size_t num_public_inputs);
/* VerifierOnlyCircuitData { constants_sigmas_cap, circuit_digest }: the cap's hashes, then the digest ((4 << cap_height) + 4) * 8 bytes */
int lcp2_verifier_data_to_bytes(const lcp2_circuit *c, uint8_t *out, size_t out_len);

/* Verifier-only circuit (VerifierCircuitData): no device, no context.  Takes the gate set, k_is and
 * parameters from `desc` (constants_sigmas is ignored and may be NULL) plus the circuit digest and the
 * constants_sigmas cap published by the prover's build().  lcp2_prove on it returns LCP2_E_NODEVICE. */
int lcp2_verifier_create(const lcp2_circuit_desc *desc, const uint64_t digest[4], const uint64_t *cap, lcp2_circuit **out);

/* challenges of the last lcp2_prove (for stage-wise parity tests): betas[4], gammas[4], alphas[4], zeta[2],
 * fri_alpha[2], fri_betas[8][2], pow_witness, query_indices[64] -- 4+4+4+2+2+16+1+64 = 97 words */
int lcp2_last_challenges(const lcp2_circuit *c, uint64_t out[97]);

/* ------------------------------------------------------------------ timing
 * Per-kernel-family HIP-event timing on the context's stream.  Accumulates
 * while enabled; lcp2_prof_get synchronises and reports totals since the last reset. */
typedef enum {
  LCP2_K_INTT = 0,      /* K1 */
  LCP2_K_LDE = 1,       /* K2 */
  LCP2_K_LEAF_HASH = 2, /* K4a */
  LCP2_K_MERKLE = 3,    /* K4b */
  LCP2_K_PERM_Z = 4,    /* K5 */
  LCP2_K_QUOTIENT = 5,  /* K6 */
  LCP2_K_OPENINGS = 6,  /* K7 */
  LCP2_K_FRI = 7,       /* K8 */
  LCP2_K_POW = 8,       /* K9 */
  LCP2_K_SHA256 = 9,    /* K10 */
  LCP2_K_OTHER = 10,
  LCP2_K_COUNT = 11
} lcp2_kernel_family;
int lcp2_prof_enable(lcp2_ctx *ctx, int on);
int lcp2_prof_reset(lcp2_ctx *ctx);
int lcp2_prof_get(lcp2_ctx *ctx, int family, double *total_ms, uint64_t *launches, double *algorithmic_bytes);

#ifdef __cplusplus
}
#endif
#endif
