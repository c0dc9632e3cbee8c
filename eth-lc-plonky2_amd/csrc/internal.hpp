// Internal declarations shared by the .hip translation units of liblcp2.so.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include "../../include/lcp2.h"
#include "gl64.hpp"
#include "ntt.hpp"

namespace lcp2 {

struct ProfFamily {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double ms = 0;
  uint64_t launches = 0;
  double bytes = 0;
};

}  // namespace lcp2

struct lcp2_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string last_error;
  lcp2::u64 *d_rc = nullptr;  // POS_RC_WORDS: the 360 Poseidon round constants + the group constants of the partial rounds
  std::map<std::string, lcp2::u64 *> tables;
  bool prof_on = false;
  lcp2::ProfFamily fam[LCP2_K_COUNT];
  std::vector<hipEvent_t> event_pool;
  // pinned host staging for the small device-to-host copies of a proof (caps, openings, flags, query answers): a copy into pinned
  // memory is a plain DMA that the stream orders; into pageable memory the runtime stages it and blocks
  void *pin = nullptr;
  static constexpr size_t PIN_BYTES = 4u << 20;
  hipStream_t copy_stream = nullptr;  // uploads of staged host witnesses (lcp2_witness_stage), created at the first use
  // device scratch of the witness-generation entry points (job lists, message words, round records, cell lists): kept and grown,
  // not allocated and freed per call (a hipMalloc / hipFree pair costs more than the kernels of a light-client step's witness)
  void *scratch[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t scratch_bytes[4] = {0, 0, 0, 0};

  int fail(int code, const std::string &msg) {
    last_error = msg;
    return code;
  }
};

namespace lcp2 {

#define LCP2_HIP(ctx, expr)                                                                  \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return (ctx)->fail(e_ == hipErrorOutOfMemory ? LCP2_E_OOM : LCP2_E_HIP,                \
                         std::string(#expr) + ": " + hipGetErrorString(e_));                 \
  } while (0)

// RAII device allocation
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  hipError_t alloc(size_t n) {
    release();
    if (n == 0) n = 8;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n; else p = nullptr;
    return e;
  }
  // keep the allocation if it is already large enough (persistent prover workspaces)
  hipError_t ensure(size_t n) { return (p && bytes >= n) ? hipSuccess : alloc(n); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
  u64 *u() const { return (u64 *)p; }
};

// scope-based event timing of one kernel family (no-op unless profiling is enabled)
struct ProfScope {
  lcp2_ctx *ctx;
  int family;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(lcp2_ctx *c, int fam, double algorithmic_bytes);
  ~ProfScope();
};

// ---- kernel launch wrappers (device pointers, asynchronous on `s`) ----
void launch_poseidon_permute_batch(hipStream_t s, const u64 *in, u64 *out, size_t count, const u64 *rc);
void launch_field_op(hipStream_t s, const u64 *a, const u64 *b, u64 *out, size_t count, u32 op);
// leaf i, element c at  data[i * leaf_stride + c * col_stride]
void launch_hash_leaves(hipStream_t s, const u64 *data, u64 leaf_stride, u64 col_stride, u32 leaf_len, u64 nleaves,
                        u64 *digests, const u64 *rc);
// leaves of `arity` extension elements taken from two planes (c0, c1): leaf k = flatten(e[arity*k .. arity*(k+1)))
// the sponge of k_hash_leaves over `ncols` columns starting at `data` (element c of leaf i at data[c * col_stride + i]), state[12][nleaves]
// carried between launches: first = start from the zero state, last = write the digests instead of the state
void launch_hash_leaves_absorb(hipStream_t s, const u64 *data, u64 col_stride, u32 ncols, u64 nleaves, u64 *state, bool first, bool last,
                               u64 *digests, const u64 *rc);
void launch_hash_ext_leaves(hipStream_t s, const u64 *p0, const u64 *p1, u32 arity, u64 nleaves, u64 *digests, const u64 *rc);
void launch_merkle_level(hipStream_t s, const u64 *children, u64 *parents, u64 nparents, const u64 *rc);
void launch_ntt_pass(hipStream_t s, bool inverse, const NttPassParams &p, u32 wgs, u32 cols, u32 nz);
void launch_bitrev_tile(hipStream_t s, const BitrevTile &b, u32 wgs, u32 cols);
void launch_bitrev_small(hipStream_t s, const u64 *in, u64 is, u64 *out, u64 os, u32 lg, u32 cols, unsigned long long *noncanonical = nullptr);
void launch_copy_2d(hipStream_t s, u64 *dst, u64 dst_pitch_words, const u64 *src, u64 src_pitch_words, u64 width_words, u32 height);
void launch_canon_copy(hipStream_t s, const u64 *in, u64 *out /* nullable: scan only */, u64 count, unsigned long long *noncanonical);
void launch_sha256_level(hipStream_t s, const uint8_t *children, uint8_t *parents, u64 nparents_per_tree, u64 trees,
                         u64 child_tree_stride, u64 parent_tree_stride, uint32_t *trace, u64 trace_tree_stride,
                         u64 trace_hash_offset);
void launch_gather_rows(hipStream_t s, const u64 *data, u64 col_stride, u32 ncols, const u64 *indices, u32 k, u64 *out);
void launch_gather_digests(hipStream_t s, const u64 *digests, const u64 *level_offsets, u32 nsib, const u64 *indices,
                           u32 k, u64 *out);

struct ShaJobDev;
struct CellDev;
void launch_sha_jobs_level(hipStream_t s, const ShaJobDev *jobs, u32 first, u32 count, const uint32_t *words_in, uint32_t *rec);
void launch_sha_fill_rows(hipStream_t s, const ShaJobDev *jobs, u32 njobs, const uint32_t *rec, u64 *wires, u64 n);
void launch_scatter_cells(hipStream_t s, const CellDev *cells, u64 ncells, u64 *wires, u64 n);
struct PoseidonRowDev;
void launch_poseidon_gate_rows(hipStream_t s, const PoseidonRowDev *rows, u64 nrows, u64 *wires, u64 n, const u64 *rc);

// ---- device-side NTT backend over the launch wrappers ----
struct DeviceNttBackend {
  lcp2_ctx *ctx;
  int status = 0;
  const u64 *table(const std::string &key, std::function<std::vector<u64>()> make);
  void launch_pass(bool inv, const NttPassParams &p, u32 wgs, u32 cols, u32 nz) { launch_ntt_pass(ctx->stream, inv, p, wgs, cols, nz); }
  void launch_bitrev(const BitrevTile &b, u32 wgs, u32 cols) { launch_bitrev_tile(ctx->stream, b, wgs, cols); }
  void launch_bitrev_small(const u64 *in, u64 is, u64 *out, u64 os, u32 lg, u32 cols, unsigned long long *noncanonical = nullptr) {
    lcp2::launch_bitrev_small(ctx->stream, in, is, out, os, lg, cols, noncanonical);
  }
};

}  // namespace lcp2

// = PolynomialBatch
struct lcp2_oracle {
  lcp2_ctx *ctx = nullptr;
  uint32_t ncols = 0, log_n = 0, rate_bits = 0, cap_height = 0;
  lcp2::DevBuf coeffs;   // [ncols][n], natural coefficient order
  lcp2::DevBuf lde;      // [ncols][n << rate_bits], Merkle leaf order
  lcp2::DevBuf digests;  // level 0 (leaf digests) ... cap level, 4 u64 per node
  std::vector<uint64_t> level_off;  // node offset of each level inside `digests`
  lcp2::DevBuf d_level_off;
  // coset-sharded oracle: only the leaf blocks [block_first, block_first + block_count) are held;
  // block_count = 0 means all 2^rate_bits blocks
  uint32_t block_first = 0, block_count = 0;
  uint32_t blocks() const { return block_count ? block_count : (1u << rate_bits); }
  uint32_t log_blocks() const { uint32_t l = 0; while ((1u << l) < blocks()) l++; return l; }
  uint64_t nleaves() const { return (uint64_t)blocks() << log_n; }
  // the local cap: 2^(cap_height - rate_bits) entries per block
  uint32_t local_cap_height() const { return cap_height - rate_bits + log_blocks(); }
  uint32_t nlevels() const { return log_n + log_blocks() - local_cap_height() + 1; }
  const lcp2::u64 *cap_dev() const { return digests.u() + 4 * level_off.back(); }
};

namespace lcp2 {
// internal commitment builders on device-resident input
// shape checks shared by build(), the verifier-only constructor and every entry point that computes a proof layout: nullptr or
// the reason; *unsupported tells LCP2_E_UNSUPPORTED from LCP2_E_INVALID (prover.hip)
const char *params_problem(const lcp2_params &p, bool *unsupported);
int commit_values_dev(lcp2_ctx *ctx, const u64 *d_vals, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                      uint32_t cap_height, lcp2_oracle *o, unsigned long long *noncanonical = nullptr /* device word, set to 1 if a value is >= p */);
int commit_coeffs_dev(lcp2_ctx *ctx, const u64 *d_coeffs, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                      uint32_t cap_height, lcp2_oracle *o, bool take_copy);
int build_merkle_dev(lcp2_ctx *ctx, lcp2_oracle *o);
// its two halves for a commitment that hashes its leaves chunk by chunk: the digest storage and level offsets, then the levels above the leaves
int merkle_alloc_dev(lcp2_ctx *ctx, lcp2_oracle *o);
int merkle_levels_dev(lcp2_ctx *ctx, lcp2_oracle *o);
}  // namespace lcp2
