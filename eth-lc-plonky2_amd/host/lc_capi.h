/*
 * lc_capi.h -- the C++ host layer's light-client workloads behind plain C entry points, so that a process that is not C++
 * (bench.py, the Python tests) drives the SAME path examples/lc_prover drives: the reference's main() (eth-lc-plonky2/src/main.rs:
 * 56-233: parse two updates, add_virtual_proof_target + register_public_input, builder.build(), set_proof_target, data.prove(pw),
 * data.verify(proof)) in-process, on a context and stream the caller owns.  lch_prove is data.prove(pw): witness generation (the
 * reference times generate_partial_witness inside prove(), src/main.rs:229-232) and lcp2_prove.
 * This is NOT part of the drop-in boundary (include/lcp2.h is): a plonky2 fork has its own CircuitBuilder.  No oracle, no CPU fallback.
 * Every function returns 0 or a negative lcp2_status; lch_last_error() says why (thread-local).
 */
#ifndef LC_CAPI_H
#define LC_CAPI_H
#include <stddef.h>
#include <stdint.h>
#include "../../include/lcp2.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lch_session lch_session;

#define LCH_BLS_PROOF_STAND_IN 1u    /* the circuit also verifies, recursively, a proof with the BLS proof's 25 216 public inputs (stand-in statement circuit) */
#define LCH_SYNC_COMMITTEE_ONLY 2u   /* BASELINE configs[1]: the SyncCommitteeSSZ gadget of cur_json's next_sync_committee alone */

typedef struct {
  uint32_t degree_bits;
  uint32_t num_public_inputs;
  uint64_t num_gates;          /* builder.num_gates() before padding */
  uint64_t proof_words;
  double build_ms;             /* CircuitBuilder -> build() on the host */
  double attach_ms;            /* lcp2_circuit_create: the preprocessed polynomials committed on the GPU */
  double inner_prove_ms;       /* LCH_BLS_PROOF_STAND_IN: the inner proof (not part of lch_prove) */
  uint32_t inner_degree_bits, inner_public_inputs;
} lch_info;

/* prev_json / cur_json: two consecutive light-client updates (beacon-API V1_5 layout or the layout of the reference's fixture files).
 * extra_committees: that many more SyncCommitteeSSZ gadgets in the same circuit (6 -> 7 207 two_to_one_sha256, 2.24 M gates, 2^22 rows:
 * the reference's scale made of real gadgets).  ctx: the caller's context (its device and stream); it must outlive the session. */
int lch_light_client_step_create(lcp2_ctx *ctx, const char *prev_json, const char *cur_json, uint32_t flags, uint32_t extra_committees, lch_session **out);
void lch_destroy(lch_session *s);
int lch_get_info(const lch_session *s, lch_info *out);
/* data.prove(pw): proof (info.proof_words) and public inputs (info.num_public_inputs), host buffers.  LCP2_E_UNSAT for a witness
 * the circuit does not accept. */
int lch_prove(lch_session *s, uint64_t *proof, size_t proof_words, uint64_t *public_inputs, size_t num_public_inputs);
/* data.verify(proof): LCP2_OK or LCP2_E_VERIFY */
int lch_verify(const lch_session *s, const uint64_t *proof, size_t proof_words, const uint64_t *public_inputs, size_t num_public_inputs);
/* LCH_SYNC_COMMITTEE_ONLY: the native SSZ root the proved public inputs must equal (8 big-endian u32 words); else cur_state then new_state (16 words) */
int lch_expected_public_inputs(const lch_session *s, uint64_t *out, size_t count);
const char *lch_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
