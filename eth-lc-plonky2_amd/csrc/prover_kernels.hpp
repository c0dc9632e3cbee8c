// Argument blocks and launch wrappers of the prover kernels (kernels_prover.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "gl64.hpp"
#include "ntt.hpp"

namespace lcp2 {

constexpr u32 PERM_MAX_CHUNKS = 10;   // ceil(80 routed wires / quotient degree factor 8)
constexpr u32 QUOTIENT_THREADS = 128;
constexpr u32 QUOTIENT_MAX_CH = 2;
// Operand staging of the gate-program interpreter (device-side form of the programs, made by stage_gate_programs at build()):
// every WIRE / CONST operand is fetched by an LDG instruction that issues up to QUOTIENT_STAGE column loads back to back and
// parks the values in LDS slots; the instructions after it read kind STAGE.  One exposed HBM round trip per 16 operands
// instead of one per operand (the interpreter executes one instruction at a time, so an operand load cannot overlap anything).
constexpr u32 QUOTIENT_STAGE = 12;
constexpr u32 QUOTIENT_ALPHA_POWS = 32; // >= CH + CH * PERM_MAX_CHUNKS + 1 exponents
constexpr u32 QUOTIENT_TERM_POWS = 128; // alpha^e, e < 128, as 22-bit limbs: the constraint weights of the generated gates (at most 128 constraints)
constexpr u32 QUOTIENT_GENERATED_GATES = 17; // programs in csrc/generated_gates*.hpp (static_assert-ed against the index file)
constexpr u32 QOP_LDG = 10;      // w0 = 10 | count << 8, w1 = offset into stage_list
constexpr u32 QKIND_STAGE = 5;   // operand = staging slot idx
constexpr u32 EVAL_CHUNK = 4096;

// word offsets inside lcp2_circuit::small (per-proof scalars on the device)
constexpr size_t SMALL_BETAS = 0, SMALL_GAMMAS = 4, SMALL_ALPHAS = 8, SMALL_ALPHA_INV = 12, SMALL_PI_HASH = 16, SMALL_POW = 20,
                 SMALL_CHECK = 21, SMALL_NONCANON = 22, SMALL_PERM_PREFIX = 24, SMALL_ALPHA_POW = 32, SMALL_GATE_SCALE = SMALL_ALPHA_POW + QUOTIENT_MAX_CH * QUOTIENT_ALPHA_POWS;

struct GateDev {  // = lcp2_gate
  u32 selector_index, selector_value, group_start, group_end, code_offset, code_len, num_constraints, flags;
};

struct PermArgs {
  const u64 *wires;    // witness values on the rows [row0, row0 + n) of H, [num_wires][wires_stride]
  const u64 *sigmas;   // sigma values on the same rows (the pointer is to row0), [num_routed][sigma_stride]
  const u64 *k_is;     // [num_routed]
  TwoLevelTable subgroup;  // w_n^row
  const u64 *betas, *gammas;  // device, [num_challenges]
  const u64 *prefix;   // nullable, [num_challenges]: k_perm_finalize multiplies Z and the partial products by it (a row block of a
                       // sharded proof: the product of the blocks before it)
  u64 *chunk_q;        // scratch [num_challenges][nchunks][n]
  u64 *row_tot;        // scratch [num_challenges][n]
  u64 *zs_out;         // [num_challenges * (1 + npp)][n]: Z_0.., then partial products per challenge
  u64 n, row0;         // rows of this launch (all of H: row0 = 0) and the global index of the first
  u64 wires_stride, sigma_stride;
  u32 num_routed, chunk, nchunks, num_challenges;
};

struct QuotientArgs {
  const u64 *wires;   // LDE, leaf order, [num_wires][N]
  const u64 *consts;  // LDE of constants then sigmas, [num_constants + num_routed][N]
  const u64 *zs;      // LDE of Z / partial products, [CH * (1 + npp)][N]
  const u64 *l0;      // L_0 on the LDE points, leaf order [N]
  const u64 *zh_inv;  // 1 / Z_H per top-bits block of the leaf index, [2^rate_bits]
  TwoLevelTable points;  // 7 * w_N^j
  const u64 *k_is, *betas, *gammas, *alphas, *pis, *imm;  // pis: public_inputs_hash[4]
  u32 kis_pow7;            // k_is[j] = 7^j (plonky2's get_unique_coset_shifts): beta x k_j is carried from wire to wire with a multiply by 7
  // forward-emitting gates (LCP2_GATE_EMIT_FORWARD): sum_i alpha^i c_i = alpha^(m-1) * Horner(c_0 .. c_{m-1}; 1/alpha)
  const u32 *stage_list;   // LDG column lists: entry < num_wires: wire column, else constants column (entry - num_wires)
  u32 num_wires;
  u32 use_native;          // 1: gates flagged LCP2_GATE_NATIVE_* run their native evaluator, 0: everything is interpreted
  const u64 *rc;           // Poseidon round constants (native PoseidonGate evaluator)
  const u64 *alpha_pow;    // [QUOTIENT_MAX_CH][QUOTIENT_ALPHA_POWS] alpha_c^e: weights of the permutation-term blocks
  u32 limbs_lds_word;       // where in a kernel's dynamic LDS (in u64 words) the copy of alpha_limbs for the generated gates sits (set per launch)
  const u32 *alpha_limbs;  // [QUOTIENT_MAX_CH][QUOTIENT_TERM_POWS][4]: alpha_c^e cut into three 22-bit limbs (+ one word of padding): a
                           // generated gate adds constraint x limbs into six 64-bit column sums per challenge, one multiply-accumulate
                           // each and no reduction, and folds the columns once per point (kernels_prover.hip QTerms)
  const u64 *alpha_inv;    // [CH], 0 where alpha = 0
  const u64 *gate_scale;   // [num_gates][QUOTIENT_MAX_CH] alpha^(num_constraints - 1)
  const u32 *code;
  const GateDev *gates;
  u64 *out;           // [CH][N] quotient values, leaf order (always the full domain)
  u64 N;              // size of the LDE domain
  // coset-sharded circuits hold only the leaf blocks [leaf0, leaf0 + count) of every LDE: `stride` is the column stride of
  // wires / consts / zs (= count), l0 / points / out are indexed with the global leaf index.  Unsharded: 0, N, N.
  u64 leaf0, count, stride;
  u32 lgN, rate_bits, num_gates, num_selectors, num_constants, num_routed, chunk, nchunks, num_challenges, num_regs;
};

struct EvalArgs {
  const u64 *coeffs;  // [npolys][col_stride]
  u64 col_stride;
  u32 chunk_len, items, nchunks;
  u64 zstep[2];            // z^256
  const u64 *zpow_t;       // z^t, t < 256 (ext, interleaved)
  const u64 *zpow_chunk;   // z^(chunk * chunk_len)
  u64 *partial;            // [npolys][nchunks] ext
};

struct ComposeArgs {
  const u64 *coeffs[4];
  u32 ncols[4];
  u32 num_challenges;
  u64 n;
  const u64 *alpha_pows;   // alpha^j ext, j < total polys
  const u64 *z0_lo, *z0_hi, *z1_lo, *z1_hi;      // zeta^i, (g zeta)^i two-level ext tables
  const u64 *zi0_lo, *zi0_hi, *zi1_lo, *zi1_hi;  // inverse powers
  u32 zh;
  u64 zmask;
  u64 alpha_shift[2];      // alpha^CH
  u64 *planes;             // [4][n]
};

struct PowArgs {
  u64 state[12];
  u32 pos, bits;
  u64 start;
  const u64 *rc;
  u64 *result;
};

void launch_scan(hipStream_t s, bool mul, const u64 *in, u64 *out, u64 *block_tot, u64 n, bool reverse, u32 batches, u64 batch_stride);
u64 scan_scratch_words(u64 n, u32 batches);
void launch_quotient_combine(hipStream_t s, const u64 *in, u64 *out, const u64 *m, u64 n, u32 R, u64 plane, u32 num_challenges);
void launch_perm_chunks(hipStream_t s, const PermArgs &a);
void launch_perm_finalize(hipStream_t s, const PermArgs &a);
void launch_quotient(hipStream_t s, const QuotientArgs &a, const std::vector<GateDev> &host_gates);
// host: rewrites validated gate programs into the staged device form (new code, per-gate offsets in `gates`, column lists)
void stage_gate_programs(const std::vector<uint32_t> &code, std::vector<GateDev> &gates, u32 num_wires, u32 num_selectors,
                         std::vector<uint32_t> &staged_code, std::vector<uint32_t> &stage_list);
// Row-wise check of the gate constraints over the n rows of H (the Err of prove() for an unsatisfiable witness): the same
// gate programs on the witness VALUES (a.wires = witness [W][n], a.consts = constants values [NC][n], a.stride = a.count = n);
// *flag (device, zeroed by the caller) receives 1 + the smallest row with a non-zero filtered constraint combination.
void launch_gate_check(hipStream_t s, const QuotientArgs &a, const std::vector<GateDev> &host_gates, unsigned long long *flag);
// build()-time check of the LCP2_GATE_NATIVE_* claims: on a.count random points (a.wires / a.consts hold random field
// elements) the native evaluators and the interpreted programs must give the same combination; *flag as in launch_gate_check
void launch_native_check(hipStream_t s, const QuotientArgs &a, const std::vector<GateDev> &host_gates, unsigned long long *flag);
void launch_eval_polys(hipStream_t s, const EvalArgs &a, u32 npolys, u64 *out);
void launch_compose(hipStream_t s, const ComposeArgs &a);
void launch_divide_finalize(hipStream_t s, const ComposeArgs &a, u64 *out0, u64 *out1);
void launch_fri_fold(hipStream_t s, const u64 *c0, const u64 *c1, u64 *o0, u64 *o1, u64 nout, u32 arity, u64 b0, u64 b1);
void launch_gather_ext_leaves(hipStream_t s, const u64 *p0, const u64 *p1, u32 arity, const u64 *leaf_idx, u32 k, u64 *out);
void launch_pow_search(hipStream_t s, const PowArgs &a, u64 count);
void launch_fill(hipStream_t s, u64 *p, u64 n, u64 v);

// ---- challenge-dependent setup on the device.  The challenges of a proof are a handful of words that the host draws from the
// transcript; everything the kernels need that is derived from them (powers, inverses, limb tables) is computed by these small
// kernels from the challenges passed BY VALUE in the kernel arguments: no staging vector, no host-to-device copy, no
// synchronisation to keep a stack buffer alive.
struct SmallWords { u64 v[16]; };
// dst[i] = w.v[i], i < n <= 16 (betas / gammas, a flag reset, the prefix products of a row block)
void launch_set_words(hipStream_t s, u64 *dst, const SmallWords &w, u32 n);
// what compute_quotient_polys needs of alpha: small[SMALL_ALPHAS ..], the inverses, alpha^e (e < QUOTIENT_ALPHA_POWS), alpha^(m_g - 1) per gate,
// the public-input hash, the reset gate-check flag, and the 22-bit limb table of alpha^e, e < QUOTIENT_TERM_POWS (QuotientArgs::alpha_limbs)
struct QuotientSetupArgs {
  u64 alphas[QUOTIENT_MAX_CH], pi_hash[4];
  u32 num_challenges, num_gates;
  const GateDev *gates;
  u64 *small;
  u32 *limbs;
};
void launch_quotient_setup(hipStream_t s, const QuotientSetupArgs &a);
// K7a tables: tab[2 t ..] = z^t (t < 256), tab[512 + 2 k ..] = z^(chunk_len k) (k < nchunks)
void launch_eval_tables(hipStream_t s, u64 z0, u64 z1, u32 chunk_len, u32 nchunks, u64 *tab);
// K7b tables: alpha^j (j < total_polys) at T[0], then for each of zeta, g zeta, 1 / zeta, 1 / (g zeta) a two-level power table
// lo[j] = b^j (j < 2^h) followed by hi[j] = b^(j << h) (j < hi_count), every entry an extension element [c0, c1]
void launch_compose_tables(hipStream_t s, u64 a0, u64 a1, u64 z0, u64 z1, u64 g, u32 total_polys, u32 h, u64 hi_count, u64 *T);

}  // namespace lcp2
