// Row layout of one two_to_one_sha256 in the own SHA-256 circuit (host/gates.cpp, host/builder.cpp) as the
// witness kernels see it.  310 rows:  [0,48) schedule rows W16..W63 | [48,176) 64 x (round E row, round A row) of the
// data block | [176,179) 3 addition rows | [179,307) rounds of the constant padding block | [307,310) additions.
#pragma once
#include <stdint.h>

namespace lcp2 {
constexpr uint32_t SHA_ROWS = 310;
constexpr uint32_t SHA_ROW_ROUNDS0 = 48;
// per-hash record written by k_sha_jobs_level (32-bit words)
constexpr uint32_t SHA_REC_IN = 0;       // 16 message words
constexpr uint32_t SHA_REC_SCHED = 16;   // 48 schedule words W16..W63
constexpr uint32_t SHA_REC_AE0 = 64;     // data block: 64 x (a, e) after each round
constexpr uint32_t SHA_REC_MID = 192;    // chaining value after the data block
constexpr uint32_t SHA_REC_AE1 = 200;    // padding block rounds
constexpr uint32_t SHA_REC_DIGEST = 328; // final digest words
constexpr uint32_t SHA_REC_WORDS = 336;

struct ShaJobDev {          // = lcp2_sha_job
  uint32_t first_row;
  int32_t in_src[16];       // >= 0: index into words_in ; < 0: ~(job * 8 + word) = digest word of an earlier job
};
struct CellDev {            // = lcp2_cell
  uint32_t row, col;
  uint64_t value;
};
struct PoseidonRowDev {     // = lcp2_poseidon_row: one PoseidonGate row to generate (kernels_witness.hip k_poseidon_gate_rows)
  uint32_t row, swap;
  unsigned long long in[12];
};
}  // namespace lcp2
