// K6: the generated straight-line evaluators of csrc/generated_gates_reca.hpp as kernels (kernels_gates.hpp says how).
#include "kernels_gates.hpp"
#if defined(__HIP_DEVICE_COMPILE__)
#include "generated_gates_reca.hpp"
#endif

namespace lcp2 {
LCP2_DEFINE_GENERATED_UNIT(reca, Q_GENERATED_RECA_FIRST, Q_GENERATED_RECA_COUNT)
}  // namespace lcp2
