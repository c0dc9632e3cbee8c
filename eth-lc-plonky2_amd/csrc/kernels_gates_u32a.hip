// K6: the generated straight-line evaluators of csrc/generated_gates_u32a.hpp as kernels (kernels_gates.hpp says how).
#include "kernels_gates.hpp"
#if defined(__HIP_DEVICE_COMPILE__)
#include "generated_gates_u32a.hpp"
#endif

namespace lcp2 {
LCP2_DEFINE_GENERATED_UNIT(u32a, Q_GENERATED_U32A_FIRST, Q_GENERATED_U32A_COUNT)
}  // namespace lcp2
