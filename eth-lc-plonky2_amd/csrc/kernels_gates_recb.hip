// K6: the generated straight-line evaluators of csrc/generated_gates_recb.hpp as kernels (kernels_gates.hpp says how).
#include "kernels_gates.hpp"
#if defined(__HIP_DEVICE_COMPILE__)
#include "generated_gates_recb.hpp"
#endif

namespace lcp2 {
LCP2_DEFINE_GENERATED_UNIT(recb, Q_GENERATED_RECB_FIRST, Q_GENERATED_RECB_COUNT)
}  // namespace lcp2
