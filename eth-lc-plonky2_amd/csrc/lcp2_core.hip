// Context, timing, primitive entry points and polynomial commitments of liblcp2.so.
// See include/lcp2.h for the ABI contract and the plonky2 functions each call replaces.
#include <cstring>
#include "internal.hpp"
#include "ntt_host.hpp"
#include "poseidon.hpp"
#include "sha_layout.hpp"

using namespace lcp2;

// ------------------------------------------------------------------ misc
extern "C" const char *lcp2_status_str(int s) {
  switch (s) {
    case LCP2_OK: return "ok";
    case LCP2_E_INVALID: return "invalid argument";
    case LCP2_E_NODEVICE: return "no usable HIP device";
    case LCP2_E_HIP: return "HIP runtime error";
    case LCP2_E_OOM: return "out of device memory";
    case LCP2_E_UNSAT: return "witness does not satisfy the circuit";
    case LCP2_E_UNSUPPORTED: return "unsupported";
    case LCP2_E_VERIFY: return "proof rejected";
    default: return "unknown status";
  }
}
extern "C" int lcp2_abi_version(void) { return LCP2_ABI_VERSION; }
extern "C" int lcp2_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int lcp2_params_standard(uint32_t degree_bits, uint32_t num_constants, lcp2_params *p) {
  if (!p || degree_bits == 0 || degree_bits > 28) return LCP2_E_INVALID;
  memset(p, 0, sizeof *p);
  p->degree_bits = degree_bits;
  p->num_wires = 135; p->num_routed_wires = 80; p->num_constants = num_constants;
  p->rate_bits = 3; p->cap_height = 4; p->num_challenges = 2; p->quotient_degree_factor = 8;
  p->proof_of_work_bits = 16; p->num_query_rounds = 28;
  // FriReductionStrategy::ConstantArityBits(4, 5)
  uint32_t d = degree_bits, n = 0;
  while (d > 5 && d + p->rate_bits - 4 >= p->cap_height && n < LCP2_MAX_FRI_LAYERS) { p->fri_arity_bits[n++] = 4; d -= 4; }
  p->num_fri_layers = n;
  return LCP2_OK;
}

// ------------------------------------------------------------------ context
extern "C" int lcp2_ctx_create(int device, void *stream, lcp2_ctx **out) { return lcp2_ctx_create_ex(device, stream, 0, out); }
extern "C" int lcp2_ctx_create_ex(int device, void *stream, uint32_t flags, lcp2_ctx **out) {
  if (!out) return LCP2_E_INVALID;
  *out = nullptr;
  if ((flags & ~LCP2_CTX_ORDER_WITH_DEFAULT_STREAM) || ((flags & LCP2_CTX_ORDER_WITH_DEFAULT_STREAM) && stream)) return LCP2_E_INVALID;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LCP2_E_NODEVICE;
  if (device < 0 || device >= ndev) return LCP2_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return LCP2_E_NODEVICE;
  lcp2_ctx *ctx = new lcp2_ctx();
  ctx->device = device;
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    const unsigned how = (flags & LCP2_CTX_ORDER_WITH_DEFAULT_STREAM) ? hipStreamDefault : hipStreamNonBlocking;
    if (hipStreamCreateWithFlags(&ctx->stream, how) != hipSuccess) { delete ctx; return LCP2_E_HIP; }
    ctx->own_stream = true;
  }
  if (hipHostMalloc(&ctx->pin, lcp2_ctx::PIN_BYTES, hipHostMallocDefault) != hipSuccess) ctx->pin = nullptr;  // without it copies go to pageable memory
  u64 rc[POS_RC_WORDS];  // the 360 round constants, then the group constants of the partial rounds (poseidon.hpp)
  pos_derive_round_constants(rc);
  pos_extend_round_constants(rc);
  if (hipMalloc((void **)&ctx->d_rc, sizeof rc) != hipSuccess ||
      hipMemcpy(ctx->d_rc, rc, sizeof rc, hipMemcpyHostToDevice) != hipSuccess) {
    lcp2_ctx_destroy(ctx);
    return LCP2_E_HIP;
  }
  *out = ctx;
  return LCP2_OK;
}

extern "C" void *lcp2_ctx_stream(lcp2_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" void lcp2_ctx_destroy(lcp2_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &kv : ctx->tables) (void)hipFree(kv.second);
  for (auto &f : ctx->fam)
    for (auto &pr : f.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->d_rc) (void)hipFree(ctx->d_rc);
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  for (void *q : ctx->scratch) if (q) (void)hipFree(q);
  if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int lcp2_ctx_sync(lcp2_ctx *ctx) {
  if (!ctx) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LCP2_OK;
}
extern "C" const char *lcp2_last_error(lcp2_ctx *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

// ------------------------------------------------------------------ timing
namespace lcp2 {
static hipEvent_t take_event(lcp2_ctx *ctx) {
  if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
ProfScope::ProfScope(lcp2_ctx *c, int fam, double algorithmic_bytes) : ctx(c), family(fam) {
  if (!ctx->prof_on) return;
  a = take_event(ctx); b = take_event(ctx);
  ctx->fam[family].bytes += algorithmic_bytes;
  (void)hipEventRecord(a, ctx->stream);
}
ProfScope::~ProfScope() {
  if (!a) return;
  (void)hipEventRecord(b, ctx->stream);
  ctx->fam[family].pending.emplace_back(a, b);
}
static void prof_collect(lcp2_ctx *ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &f : ctx->fam) {
    for (auto &pr : f.pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { f.ms += ms; f.launches++; }
      ctx->event_pool.push_back(pr.first); ctx->event_pool.push_back(pr.second);
    }
    f.pending.clear();
  }
}
}  // namespace lcp2

extern "C" int lcp2_prof_enable(lcp2_ctx *ctx, int on) {
  if (!ctx) return LCP2_E_INVALID;
  if (!on) prof_collect(ctx);
  ctx->prof_on = on != 0;
  return LCP2_OK;
}
extern "C" int lcp2_prof_reset(lcp2_ctx *ctx) {
  if (!ctx) return LCP2_E_INVALID;
  prof_collect(ctx);
  for (auto &f : ctx->fam) { f.ms = 0; f.launches = 0; f.bytes = 0; }
  return LCP2_OK;
}
extern "C" int lcp2_prof_get(lcp2_ctx *ctx, int family, double *total_ms, uint64_t *launches, double *algorithmic_bytes) {
  if (!ctx || family < 0 || family >= LCP2_K_COUNT) return LCP2_E_INVALID;
  prof_collect(ctx);
  if (total_ms) *total_ms = ctx->fam[family].ms;
  if (launches) *launches = ctx->fam[family].launches;
  if (algorithmic_bytes) *algorithmic_bytes = ctx->fam[family].bytes;
  return LCP2_OK;
}

// ------------------------------------------------------------------ device tables
const u64 *DeviceNttBackend::table(const std::string &key, std::function<std::vector<u64>()> make) {
  auto it = ctx->tables.find(key);
  if (it != ctx->tables.end()) return it->second;
  std::vector<u64> v = make();
  u64 *d = nullptr;
  if (hipMalloc((void **)&d, v.size() * sizeof(u64)) != hipSuccess ||
      hipMemcpyAsync(d, v.data(), v.size() * sizeof(u64), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    status = LCP2_E_HIP;
    ctx->last_error = "table upload failed: " + key;
    return nullptr;
  }
  ctx->tables[key] = d;
  return d;
}

// ------------------------------------------------------------------ helpers for host/device staging
namespace {
struct Staged {  // device view of a caller buffer
  DevBuf own;
  u64 *d = nullptr;
};
int stage_in(lcp2_ctx *ctx, const void *src, size_t bytes, lcp2_mem mem, Staged &s) {
  if (mem == LCP2_MEM_DEVICE) { s.d = (u64 *)src; return LCP2_OK; }
  LCP2_HIP(ctx, s.own.alloc(bytes));
  LCP2_HIP(ctx, hipMemcpyAsync(s.own.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  s.d = s.own.u();
  return LCP2_OK;
}
int stage_out_alloc(lcp2_ctx *ctx, void *dst, size_t bytes, lcp2_mem mem, Staged &s) {
  if (mem == LCP2_MEM_DEVICE) { s.d = (u64 *)dst; return LCP2_OK; }
  LCP2_HIP(ctx, s.own.alloc(bytes));
  s.d = s.own.u();
  return LCP2_OK;
}
int stage_out_finish(lcp2_ctx *ctx, void *dst, size_t bytes, lcp2_mem mem, Staged &s) {
  if (mem == LCP2_MEM_DEVICE) return LCP2_OK;
  LCP2_HIP(ctx, hipMemcpyAsync(dst, s.d, bytes, hipMemcpyDeviceToHost, ctx->stream));
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LCP2_OK;
}
#define LCP2_TRY(expr) do { int rc_ = (expr); if (rc_ != LCP2_OK) return rc_; } while (0)
}  // namespace

// ------------------------------------------------------------------ primitives
extern "C" int lcp2_poseidon_permute_batch(lcp2_ctx *ctx, const uint64_t *in, uint64_t *out, size_t count, lcp2_mem mem) {
  if (!ctx || (count && (!in || !out))) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  size_t bytes = count * 12 * sizeof(u64);
  Staged si, so;
  LCP2_TRY(stage_in(ctx, in, bytes, mem, si));
  LCP2_TRY(stage_out_alloc(ctx, out, bytes, mem, so));
  {
    ProfScope ps(ctx, LCP2_K_OTHER, 2.0 * bytes);
    launch_poseidon_permute_batch(ctx->stream, si.d, so.d, count, ctx->d_rc);
  }
  LCP2_HIP(ctx, hipGetLastError());
  return stage_out_finish(ctx, out, bytes, mem, so);
}

extern "C" int lcp2_field_op_batch(lcp2_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t count, uint32_t op, lcp2_mem mem) {
  const bool binary = op == LCP2_FIELD_MUL || op == LCP2_FIELD_ADD || op == LCP2_FIELD_SUB || op == LCP2_FIELD_ADD_LAZY ||
                      op == LCP2_FIELD_SUB_LAZY || op == LCP2_FIELD_SHL + 9;
  const bool unary = op == LCP2_FIELD_POW7 || op == LCP2_FIELD_CANON || (op > LCP2_FIELD_SHL && op <= LCP2_FIELD_SHL + 8);
  if (!ctx || !(binary || unary) || (count && (!a || !out || (binary && !b)))) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  const size_t bytes = count * sizeof(u64);
  Staged sa, sb, so;
  LCP2_TRY(stage_in(ctx, a, bytes, mem, sa));
  if (binary) LCP2_TRY(stage_in(ctx, b, bytes, mem, sb));
  LCP2_TRY(stage_out_alloc(ctx, out, bytes, mem, so));
  launch_field_op(ctx->stream, sa.d, binary ? sb.d : nullptr, so.d, count, op);
  LCP2_HIP(ctx, hipGetLastError());
  return stage_out_finish(ctx, out, bytes, mem, so);
}

namespace lcp2 {
// digests: level 0 at offset 0; fills o->level_off and all levels up to the cap
int merkle_alloc_dev(lcp2_ctx *ctx, lcp2_oracle *o) {
  const u64 N = o->nleaves();
  const u32 nlev = o->nlevels();
  o->level_off.resize(nlev);
  u64 total = 0;
  for (u32 l = 0; l < nlev; l++) { o->level_off[l] = total; total += N >> l; }
  LCP2_HIP(ctx, o->digests.ensure(total * 4 * sizeof(u64)));
  LCP2_HIP(ctx, o->d_level_off.ensure(nlev * sizeof(u64)));
  LCP2_HIP(ctx, hipMemcpyAsync(o->d_level_off.p, o->level_off.data(), nlev * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
  return LCP2_OK;
}
int merkle_levels_dev(lcp2_ctx *ctx, lcp2_oracle *o) {
  const u64 N = o->nleaves();
  const u32 nlev = o->nlevels();
  ProfScope ps(ctx, LCP2_K_MERKLE, 96.0 * (double)(N - (N >> (nlev - 1))));
  for (u32 l = 1; l < nlev; l++)
    launch_merkle_level(ctx->stream, o->digests.u() + 4 * o->level_off[l - 1], o->digests.u() + 4 * o->level_off[l], N >> l, ctx->d_rc);
  LCP2_HIP(ctx, hipGetLastError());
  return LCP2_OK;
}
int build_merkle_dev(lcp2_ctx *ctx, lcp2_oracle *o) {
  const u64 N = o->nleaves();
  LCP2_TRY(merkle_alloc_dev(ctx, o));
  {
    ProfScope ps(ctx, LCP2_K_LEAF_HASH, (double)N * (8.0 * o->ncols + 32.0));
    launch_hash_leaves(ctx->stream, o->lde.u(), 1, N, o->ncols, N, o->digests.u(), ctx->d_rc);
  }
  return merkle_levels_dev(ctx, o);
}

static int lde_and_merkle(lcp2_ctx *ctx, lcp2_oracle *o) {
  const u64 n = (u64)1 << o->log_n, N = o->nleaves();
  LCP2_HIP(ctx, o->lde.ensure((size_t)o->ncols * N * sizeof(u64)));
  DeviceNttBackend be{ctx};
  NttHost<DeviceNttBackend> ntt(be);
  {
    ProfScope ps(ctx, LCP2_K_LDE, (double)o->ncols * (8.0 * n + 8.0 * N));
    ntt.forward(o->coeffs.u(), n, o->lde.u(), N, o->log_n, o->ncols, GL_GENERATOR, o->rate_bits, o->block_first, o->block_count);
  }
  if (be.status) return be.status;
  LCP2_HIP(ctx, hipGetLastError());
  return build_merkle_dev(ctx, o);
}

int commit_values_dev(lcp2_ctx *ctx, const u64 *d_vals, size_t ncols, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height,
                      lcp2_oracle *o, unsigned long long *noncanonical) {
  if (cap_height > log_n + rate_bits || ncols == 0 || ncols > 65535) return ctx->fail(LCP2_E_INVALID, "commit: bad shape");
  o->ctx = ctx; o->ncols = (uint32_t)ncols; o->log_n = log_n; o->rate_bits = rate_bits; o->cap_height = cap_height;
  const u64 n = (u64)1 << log_n;
  LCP2_HIP(ctx, o->coeffs.ensure(ncols * n * sizeof(u64)));
  DeviceNttBackend be{ctx};
  NttHost<DeviceNttBackend> ntt(be);
  {
    ProfScope ps(ctx, LCP2_K_INTT, 16.0 * n * ncols);
    ntt.inverse_natural(d_vals, n, o->coeffs.u(), n, log_n, (u32)ncols, noncanonical);
  }
  if (be.status) return be.status;
  LCP2_HIP(ctx, hipGetLastError());
  return lde_and_merkle(ctx, o);
}

int commit_coeffs_dev(lcp2_ctx *ctx, const u64 *d_coeffs, size_t ncols, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height,
                      lcp2_oracle *o, bool take_copy) {
  if (cap_height > log_n + rate_bits || ncols == 0 || ncols > 65535) return ctx->fail(LCP2_E_INVALID, "commit: bad shape");
  o->ctx = ctx; o->ncols = (uint32_t)ncols; o->log_n = log_n; o->rate_bits = rate_bits; o->cap_height = cap_height;
  const u64 n = (u64)1 << log_n;
  LCP2_HIP(ctx, o->coeffs.ensure(ncols * n * sizeof(u64)));
  if (take_copy && d_coeffs != o->coeffs.u())  // take_copy = false: the caller already wrote o->coeffs in place
    LCP2_HIP(ctx, hipMemcpyAsync(o->coeffs.p, d_coeffs, ncols * n * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
  return lde_and_merkle(ctx, o);
}
}  // namespace lcp2

extern "C" int lcp2_merkle_cap(lcp2_ctx *ctx, const uint64_t *leaves, size_t nleaves, size_t leaf_len, uint32_t cap_height,
                               lcp2_mem mem, uint64_t *cap) {
  if (!ctx || !leaves || !cap || nleaves == 0 || (nleaves & (nleaves - 1))) return LCP2_E_INVALID;
  uint32_t h = 0;
  while (((size_t)1 << h) < nleaves) h++;
  if (cap_height > h) return ctx->fail(LCP2_E_INVALID, "cap_height exceeds tree height");
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  Staged si;
  LCP2_TRY(stage_in(ctx, leaves, nleaves * leaf_len * sizeof(u64), mem, si));
  const u32 nlev = h - cap_height + 1;
  std::vector<u64> off(nlev);
  u64 total = 0;
  for (u32 l = 0; l < nlev; l++) { off[l] = total; total += nleaves >> l; }
  DevBuf dig;
  LCP2_HIP(ctx, dig.alloc(total * 4 * sizeof(u64)));
  {
    ProfScope ps(ctx, LCP2_K_LEAF_HASH, (double)nleaves * (8.0 * leaf_len + 32.0));
    launch_hash_leaves(ctx->stream, si.d, leaf_len, 1, (u32)leaf_len, nleaves, dig.u(), ctx->d_rc);
  }
  {
    ProfScope ps(ctx, LCP2_K_MERKLE, 96.0 * (double)(nleaves - (nleaves >> (nlev - 1))));
    for (u32 l = 1; l < nlev; l++) launch_merkle_level(ctx->stream, dig.u() + 4 * off[l - 1], dig.u() + 4 * off[l], nleaves >> l, ctx->d_rc);
  }
  LCP2_HIP(ctx, hipGetLastError());
  LCP2_HIP(ctx, hipMemcpyAsync(cap, dig.u() + 4 * off[nlev - 1], ((size_t)4 << cap_height) * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LCP2_OK;
}

extern "C" int lcp2_ntt_batch(lcp2_ctx *ctx, uint64_t *data, size_t ncols, uint32_t log_n, int inverse, uint64_t shift, lcp2_mem mem) {
  if (!ctx || !data || ncols == 0 || ncols > 65535 || log_n > 30) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  shift = gl_canon(shift);
  if (shift == 0) return ctx->fail(LCP2_E_INVALID, "coset shift must be non-zero");
  const u64 n = (u64)1 << log_n;
  const size_t bytes = ncols * n * sizeof(u64);
  Staged s;
  LCP2_TRY(stage_in(ctx, data, bytes, mem, s));
  DevBuf tmp;
  LCP2_HIP(ctx, tmp.alloc(bytes));
  DeviceNttBackend be{ctx};
  NttHost<DeviceNttBackend> ntt(be);
  if (log_n == 0) {
    // size-1 transform: identity
  } else if (!inverse) {
    ProfScope ps(ctx, LCP2_K_LDE, 16.0 * n * ncols);
    ntt.forward(s.d, n, tmp.u(), n, log_n, (u32)ncols, shift, 0);
    if (log_n >= 12) { BitrevTile b{tmp.u(), s.d, n, n, log_n}; launch_bitrev_tile(ctx->stream, b, 1u << (log_n - 12), (u32)ncols); }
    else launch_bitrev_small(ctx->stream, tmp.u(), n, s.d, n, log_n, (u32)ncols);
  } else {
    ProfScope ps(ctx, LCP2_K_INTT, 16.0 * n * ncols);
    if (log_n >= 12) { BitrevTile b{s.d, tmp.u(), n, n, log_n}; launch_bitrev_tile(ctx->stream, b, 1u << (log_n - 12), (u32)ncols); }
    else launch_bitrev_small(ctx->stream, s.d, n, tmp.u(), n, log_n, (u32)ncols);
    ntt.inverse_bitrev_in(tmp.u(), n, s.d, n, log_n, (u32)ncols, shift);
  }
  if (be.status) return be.status;
  LCP2_HIP(ctx, hipGetLastError());
  if (mem == LCP2_MEM_HOST) {
    LCP2_HIP(ctx, hipMemcpyAsync(data, s.d, bytes, hipMemcpyDeviceToHost, ctx->stream));
  }
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));  // tmp is freed on return
  return LCP2_OK;
}

extern "C" int lcp2_lde_batch(lcp2_ctx *ctx, const uint64_t *coeffs, uint64_t *out, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                              lcp2_mem mem) {
  if (!ctx || !coeffs || !out || ncols == 0 || ncols > 65535 || log_n + rate_bits > 30) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  const u64 n = (u64)1 << log_n, N = n << rate_bits;
  Staged si, so;
  LCP2_TRY(stage_in(ctx, coeffs, ncols * n * sizeof(u64), mem, si));
  LCP2_TRY(stage_out_alloc(ctx, out, ncols * N * sizeof(u64), mem, so));
  DeviceNttBackend be{ctx};
  NttHost<DeviceNttBackend> ntt(be);
  if (log_n == 0) {
    for (size_t c = 0; c < ncols; c++)  // constant polynomial: every evaluation equals the coefficient
      for (u64 i = 0; i < N; i++) LCP2_HIP(ctx, hipMemcpyAsync(so.d + c * N + i, si.d + c, sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    ProfScope ps(ctx, LCP2_K_LDE, (double)ncols * (8.0 * n + 8.0 * N));
    ntt.forward(si.d, n, so.d, N, log_n, (u32)ncols, GL_GENERATOR, rate_bits);
  }
  if (be.status) return be.status;
  LCP2_HIP(ctx, hipGetLastError());
  LCP2_TRY(stage_out_finish(ctx, out, ncols * N * sizeof(u64), mem, so));
  if (mem == LCP2_MEM_DEVICE) LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LCP2_OK;
}

extern "C" int lcp2_sha256_tree(lcp2_ctx *ctx, const uint8_t *leaves, uint32_t height, size_t trees, uint8_t *nodes,
                                uint32_t *round_trace, lcp2_mem mem) {
  if (!ctx || !leaves || !nodes || height == 0 || height > 24 || trees == 0) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  const u64 nl = (u64)1 << height;
  const u64 node_bytes = (2 * nl - 1) * 32;           // per tree
  const u64 nhash = nl - 1;                            // per tree
  const u64 trace_words = nhash * 2 * (48 + 128);      // per tree
  Staged sn, st;
  LCP2_TRY(stage_out_alloc(ctx, nodes, trees * node_bytes, mem, sn));
  if (round_trace) LCP2_TRY(stage_out_alloc(ctx, round_trace, trees * trace_words * 4, mem, st));
  uint8_t *dn = (uint8_t *)sn.d;
  // leaves -> level 0 of every tree
  for (size_t t = 0; t < trees; t++)
    LCP2_HIP(ctx, hipMemcpyAsync(dn + t * node_bytes, leaves + t * nl * 32, nl * 32,
                                 mem == LCP2_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  {
    ProfScope ps(ctx, LCP2_K_SHA256, 96.0 * nhash * trees);
    u64 child_off = 0, hash_off = 0;
    for (uint32_t l = 1; l <= height; l++) {
      u64 np = nl >> l;
      u64 parent_off = child_off + (nl >> (l - 1)) * 32;
      launch_sha256_level(ctx->stream, dn + child_off, dn + parent_off, np, trees, node_bytes, node_bytes,
                          round_trace ? (uint32_t *)st.d : nullptr, trace_words, hash_off);
      child_off = parent_off;
      hash_off += np;
    }
  }
  LCP2_HIP(ctx, hipGetLastError());
  LCP2_TRY(stage_out_finish(ctx, nodes, trees * node_bytes, mem, sn));
  if (round_trace) LCP2_TRY(stage_out_finish(ctx, round_trace, trees * trace_words * 4, mem, st));
  if (mem == LCP2_MEM_DEVICE) LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LCP2_OK;
}

// ------------------------------------------------------------------ commitments
static int commit_common(lcp2_ctx *ctx, const uint64_t *cols, size_t ncols, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height,
                         lcp2_mem mem, lcp2_oracle **out, uint64_t *cap, bool values) {
  if (!ctx || !cols || !out) return LCP2_E_INVALID;
  *out = nullptr;
  if (log_n == 0 || log_n + rate_bits > 30) return ctx->fail(LCP2_E_INVALID, "commit: log_n out of range");
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  Staged si;
  LCP2_TRY(stage_in(ctx, cols, (ncols << log_n) * sizeof(u64), mem, si));
  lcp2_oracle *o = new lcp2_oracle();
  int rc = values ? commit_values_dev(ctx, si.d, ncols, log_n, rate_bits, cap_height, o)
                  : commit_coeffs_dev(ctx, si.d, ncols, log_n, rate_bits, cap_height, o, true);
  if (rc == LCP2_OK && cap) {
    hipError_t e = hipMemcpyAsync(cap, o->cap_dev(), ((size_t)4 << cap_height) * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) rc = ctx->fail(LCP2_E_HIP, "cap copy failed");
  }
  if (rc == LCP2_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ctx->fail(LCP2_E_HIP, "sync failed");
  if (rc != LCP2_OK) { (void)hipStreamSynchronize(ctx->stream); delete o; return rc; }
  *out = o;
  return LCP2_OK;
}
extern "C" int lcp2_commit_values(lcp2_ctx *ctx, const uint64_t *cols, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                                  uint32_t cap_height, lcp2_mem mem, lcp2_oracle **out, uint64_t *cap) {
  return commit_common(ctx, cols, ncols, log_n, rate_bits, cap_height, mem, out, cap, true);
}
extern "C" int lcp2_commit_coeffs(lcp2_ctx *ctx, const uint64_t *coeffs, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                                  uint32_t cap_height, lcp2_mem mem, lcp2_oracle **out, uint64_t *cap) {
  return commit_common(ctx, coeffs, ncols, log_n, rate_bits, cap_height, mem, out, cap, false);
}
extern "C" int lcp2_commit_cosets(lcp2_ctx *ctx, const uint64_t *coeffs, size_t ncols, uint32_t log_n, uint32_t rate_bits,
                                  uint32_t cap_height, uint32_t block_first, uint32_t block_count, lcp2_mem mem, lcp2_oracle **out,
                                  uint64_t *cap_part) {
  if (!ctx || !coeffs || !out) return LCP2_E_INVALID;
  *out = nullptr;
  if (log_n == 0 || log_n + rate_bits > 30 || cap_height < rate_bits || cap_height > log_n + rate_bits)
    return ctx->fail(LCP2_E_INVALID, "commit_cosets: needs rate_bits <= cap_height <= log_n + rate_bits");
  if (block_count == 0 || (block_count & (block_count - 1)) || block_first % block_count || block_first + block_count > (1u << rate_bits))
    return ctx->fail(LCP2_E_INVALID, "commit_cosets: block range must be an aligned power of two");
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  Staged si;
  LCP2_TRY(stage_in(ctx, coeffs, (ncols << log_n) * sizeof(u64), mem, si));
  lcp2_oracle *o = new lcp2_oracle();
  o->block_first = block_first; o->block_count = block_count;
  int rc = commit_coeffs_dev(ctx, si.d, ncols, log_n, rate_bits, cap_height, o, true);
  const size_t capw = ((size_t)4 << (cap_height - rate_bits)) * block_count;
  if (rc == LCP2_OK && cap_part) {
    hipError_t e = hipMemcpyAsync(cap_part, o->cap_dev(), capw * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) rc = ctx->fail(LCP2_E_HIP, "cap copy failed");
  }
  if (rc == LCP2_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = ctx->fail(LCP2_E_HIP, "sync failed");
  if (rc != LCP2_OK) { (void)hipStreamSynchronize(ctx->stream); delete o; return rc; }
  *out = o;
  return LCP2_OK;
}

extern "C" void lcp2_oracle_destroy(lcp2_oracle *o) {
  if (!o) return;
  if (o->ctx) { (void)hipSetDevice(o->ctx->device); (void)hipStreamSynchronize(o->ctx->stream); }
  delete o;
}

extern "C" int lcp2_oracle_open(lcp2_oracle *o, const uint64_t *indices, size_t k, uint64_t *leaves, uint64_t *siblings) {
  if (!o || !indices || !leaves || !siblings) return LCP2_E_INVALID;
  lcp2_ctx *ctx = o->ctx;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  const u64 N = o->nleaves();
  for (size_t i = 0; i < k; i++)
    if (indices[i] >= N) return ctx->fail(LCP2_E_INVALID, "leaf index out of range");
  const u32 nsib = o->nlevels() - 1;
  DevBuf d_idx, d_leaves, d_sib;
  LCP2_HIP(ctx, d_idx.alloc(k * sizeof(u64)));
  LCP2_HIP(ctx, d_leaves.alloc(k * o->ncols * sizeof(u64)));
  LCP2_HIP(ctx, d_sib.alloc(k * (size_t)nsib * 4 * sizeof(u64)));
  LCP2_HIP(ctx, hipMemcpyAsync(d_idx.p, indices, k * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
  launch_gather_rows(ctx->stream, o->lde.u(), N, o->ncols, d_idx.u(), (u32)k, d_leaves.u());
  launch_gather_digests(ctx->stream, o->digests.u(), (const u64 *)o->d_level_off.p, nsib, d_idx.u(), (u32)k, d_sib.u());
  LCP2_HIP(ctx, hipGetLastError());
  LCP2_HIP(ctx, hipMemcpyAsync(leaves, d_leaves.p, k * o->ncols * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
  if (nsib) LCP2_HIP(ctx, hipMemcpyAsync(siblings, d_sib.p, k * (size_t)nsib * 4 * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LCP2_OK;
}

extern "C" int lcp2_oracle_read(lcp2_oracle *o, uint64_t *coeffs, uint64_t *lde) {
  if (!o) return LCP2_E_INVALID;
  lcp2_ctx *ctx = o->ctx;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  if (coeffs) LCP2_HIP(ctx, hipMemcpyAsync(coeffs, o->coeffs.p, ((size_t)o->ncols << o->log_n) * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
  if (lde) LCP2_HIP(ctx, hipMemcpyAsync(lde, o->lde.p, (size_t)o->ncols * o->nleaves() * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LCP2_OK;
}

// scratch slot `slot` of the context with at least `bytes` bytes (contents undefined); the stream orders its reuse
static int scratch_ensure(lcp2_ctx *ctx, int slot, size_t bytes, void **out) {
  if (ctx->scratch_bytes[slot] < bytes) {
    if (ctx->scratch[slot]) { LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream)); LCP2_HIP(ctx, hipFree(ctx->scratch[slot])); ctx->scratch[slot] = nullptr; ctx->scratch_bytes[slot] = 0; }
    const size_t want = bytes + bytes / 4 + 4096;
    LCP2_HIP(ctx, hipMalloc(&ctx->scratch[slot], want));
    ctx->scratch_bytes[slot] = want;
  }
  *out = ctx->scratch[slot];
  return LCP2_OK;
}
// host -> device through the pinned staging buffer when the piece fits (a plain DMA the stream orders; the caller's memory is free
// again on return), else straight from the caller's memory
static int upload_small(lcp2_ctx *ctx, void *dst, const void *src, size_t bytes, size_t *pin_used) {
  if (ctx->pin && *pin_used + bytes <= lcp2_ctx::PIN_BYTES) {
    memcpy((char *)ctx->pin + *pin_used, src, bytes);
    LCP2_HIP(ctx, hipMemcpyAsync(dst, (char *)ctx->pin + *pin_used, bytes, hipMemcpyHostToDevice, ctx->stream));
    *pin_used += (bytes + 63) & ~(size_t)63;
    return LCP2_OK;
  }
  LCP2_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return LCP2_OK;
}

// ------------------------------------------------------------------ K10: witness generation, device buffers
static_assert(sizeof(lcp2_sha_job) == sizeof(ShaJobDev) && sizeof(lcp2_cell) == sizeof(CellDev), "ABI structs must match the kernels'");

extern "C" int lcp2_sha256_witness(lcp2_ctx *ctx, const lcp2_sha_job *jobs, size_t njobs, const uint32_t *level_start, uint32_t nlevels,
                                   const uint32_t *words_in, size_t nwords, uint64_t *wires, uint64_t n, uint32_t *digests) {
  if (!ctx || !wires || (njobs && (!jobs || !level_start || nlevels == 0)) || (nwords && !words_in)) return LCP2_E_INVALID;
  if (njobs == 0) return LCP2_OK;
  if (level_start[0] != 0 || level_start[nlevels] != njobs) return ctx->fail(LCP2_E_INVALID, "sha witness: level table does not cover the jobs");
  // validate once so that the kernels cannot read or write out of range
  for (uint32_t l = 0; l < nlevels; l++) {
    if (level_start[l] > level_start[l + 1]) return ctx->fail(LCP2_E_INVALID, "sha witness: level table not monotone");
    for (uint32_t j = level_start[l]; j < level_start[l + 1]; j++) {
      if ((uint64_t)jobs[j].first_row + SHA_ROWS > n) return ctx->fail(LCP2_E_INVALID, "sha witness: rows out of range");
      for (int i = 0; i < 16; i++) {
        int32_t s = jobs[j].in_src[i];
        if (s >= 0 ? (size_t)s >= nwords : (uint32_t)((~s) >> 3) >= level_start[l]) return ctx->fail(LCP2_E_INVALID, "sha witness: bad message source");
      }
    }
  }
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  // (the pinned staging buffer is free here: every earlier transfer through it has been waited for)
  void *d_jobs, *d_words, *d_rec;
  LCP2_TRY(scratch_ensure(ctx, 0, njobs * sizeof(lcp2_sha_job), &d_jobs));
  LCP2_TRY(scratch_ensure(ctx, 1, std::max<size_t>(nwords, 1) * 4, &d_words));
  LCP2_TRY(scratch_ensure(ctx, 2, njobs * (size_t)SHA_REC_WORDS * 4, &d_rec));
  size_t pin_used = 0;
  LCP2_TRY(upload_small(ctx, d_jobs, jobs, njobs * sizeof(lcp2_sha_job), &pin_used));
  if (nwords) LCP2_TRY(upload_small(ctx, d_words, words_in, nwords * 4, &pin_used));
  {
    ProfScope ps(ctx, LCP2_K_SHA256, 96.0 * njobs + 8.0 * 108 * SHA_ROWS * njobs);
    for (uint32_t l = 0; l < nlevels; l++)
      launch_sha_jobs_level(ctx->stream, (const ShaJobDev *)d_jobs, level_start[l], level_start[l + 1] - level_start[l],
                            (const uint32_t *)d_words, (uint32_t *)d_rec);
    launch_sha_fill_rows(ctx->stream, (const ShaJobDev *)d_jobs, (u32)njobs, (const uint32_t *)d_rec, (u64 *)wires, n);
  }
  LCP2_HIP(ctx, hipGetLastError());
  // the 8 digest words of every job's record, as one strided copy (not the whole record buffer), through the pinned staging buffer
  // when they fit behind the uploads
  const bool via_pin = digests && ctx->pin && pin_used + njobs * 32 <= lcp2_ctx::PIN_BYTES;
  if (digests)
    LCP2_HIP(ctx, hipMemcpy2DAsync(via_pin ? (void *)((char *)ctx->pin + pin_used) : (void *)digests, 32, (const uint32_t *)d_rec + SHA_REC_DIGEST,
                                   (size_t)SHA_REC_WORDS * 4, 32, njobs, hipMemcpyDeviceToHost, ctx->stream));
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's job list and words may go; the digests have landed
  if (via_pin) memcpy(digests, (const char *)ctx->pin + pin_used, njobs * 32);
  return LCP2_OK;
}

extern "C" int lcp2_scatter_cells(lcp2_ctx *ctx, const lcp2_cell *cells, size_t ncells, uint64_t *wires, uint64_t n) {
  if (!ctx || !wires || (ncells && !cells)) return LCP2_E_INVALID;
  if (!ncells) return LCP2_OK;
  for (size_t i = 0; i < ncells; i++)
    if (cells[i].row >= n) return ctx->fail(LCP2_E_INVALID, "scatter: row out of range");
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  void *d;
  LCP2_TRY(scratch_ensure(ctx, 3, ncells * sizeof(lcp2_cell), &d));
  size_t pin_used = 0;
  LCP2_TRY(upload_small(ctx, d, cells, ncells * sizeof(lcp2_cell), &pin_used));  // (a list pinned by the caller - lcp2_host_register - goes up as a DMA too)
  launch_scatter_cells(ctx->stream, (const CellDev *)d, ncells, (u64 *)wires, n);
  LCP2_HIP(ctx, hipGetLastError());
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's list may go
  return LCP2_OK;
}

extern "C" int lcp2_poseidon_gate_rows(lcp2_ctx *ctx, const lcp2_poseidon_row *rows, size_t nrows, uint64_t *wires, uint64_t n) {
  static_assert(sizeof(lcp2_poseidon_row) == sizeof(PoseidonRowDev), "row job layouts must agree");
  if (!ctx || !wires || (nrows && !rows)) return LCP2_E_INVALID;
  if (!nrows) return LCP2_OK;
  for (size_t i = 0; i < nrows; i++)
    if (rows[i].row >= n || rows[i].swap > 1) return ctx->fail(LCP2_E_INVALID, "poseidon rows: row out of range or swap flag not boolean");
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  void *d;
  LCP2_TRY(scratch_ensure(ctx, 0, nrows * sizeof(lcp2_poseidon_row), &d));
  size_t pin_used = 0;
  LCP2_TRY(upload_small(ctx, d, rows, nrows * sizeof(lcp2_poseidon_row), &pin_used));
  launch_poseidon_gate_rows(ctx->stream, (const PoseidonRowDev *)d, nrows, (u64 *)wires, n, ctx->d_rc);
  LCP2_HIP(ctx, hipGetLastError());
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the job list is freed on return
  return LCP2_OK;
}

extern "C" int lcp2_host_register(lcp2_ctx *ctx, void *host, size_t bytes) {
  if (!ctx || !host || !bytes) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  LCP2_HIP(ctx, hipHostRegister(host, bytes, hipHostRegisterDefault));
  return LCP2_OK;
}
extern "C" int lcp2_host_unregister(lcp2_ctx *ctx, void *host) {
  if (!ctx || !host) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipHostUnregister(host));
  return LCP2_OK;
}
extern "C" int lcp2_buffer_alloc(lcp2_ctx *ctx, size_t bytes, void **dev) {
  if (!ctx || !dev) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  LCP2_HIP(ctx, hipMalloc(dev, bytes ? bytes : 8));
  return LCP2_OK;
}
extern "C" int lcp2_buffer_free(lcp2_ctx *ctx, void *dev) {
  if (!ctx) return LCP2_E_INVALID;
  if (!dev) return LCP2_OK;
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LCP2_HIP(ctx, hipFree(dev));
  return LCP2_OK;
}
extern "C" int lcp2_buffer_zero(lcp2_ctx *ctx, void *dev, size_t bytes) {
  if (!ctx || !dev) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipMemsetAsync(dev, 0, bytes, ctx->stream));
  return LCP2_OK;
}
extern "C" int lcp2_buffer_write(lcp2_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes) {
  if (!ctx || !dev_dst || !host_src) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  LCP2_HIP(ctx, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LCP2_OK;
}
extern "C" int lcp2_buffer_copy(lcp2_ctx *ctx, void *dev_dst, const void *dev_src, size_t bytes) {
  if (!ctx || !dev_dst || !dev_src) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  LCP2_HIP(ctx, hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LCP2_OK;
}
extern "C" int lcp2_buffer_copy_2d(lcp2_ctx *ctx, void *dev_dst, size_t dst_pitch, const void *dev_src, size_t src_pitch, size_t width, size_t height) {
  if (!ctx || !dev_dst || !dev_src || width > dst_pitch || width > src_pitch || height > 65535) return LCP2_E_INVALID;
  if ((dst_pitch | src_pitch | width | (size_t)dev_dst | (size_t)dev_src) & 7) return LCP2_E_INVALID;  // whole field elements
  if (width == 0 || height == 0) return LCP2_OK;
  LCP2_HIP(ctx, hipSetDevice(ctx->device));
  launch_copy_2d(ctx->stream, (u64 *)dev_dst, dst_pitch / 8, (const u64 *)dev_src, src_pitch / 8, width / 8, (u32)height);
  LCP2_HIP(ctx, hipGetLastError());
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LCP2_OK;
}
extern "C" int lcp2_buffer_read(lcp2_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) {
  if (!ctx || !host_dst || !dev_src) return LCP2_E_INVALID;
  LCP2_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  LCP2_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LCP2_OK;
}
