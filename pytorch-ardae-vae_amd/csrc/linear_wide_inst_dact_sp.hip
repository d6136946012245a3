// Kernel instantiations of linear_wide_kernel.h (dact_sp); see linear_wide.hip for the dispatcher.
#define ARDAE_WIDE_INST_TU
#include "linear_wide_kernel.h"

namespace ardae {
namespace wide {
ARDAE_WIDE_FOR_DACT_FLAGS(ARDAE_WIDE_INSTANTIATE, ACT_SOFTPLUS)
}  // namespace wide
}  // namespace ardae
