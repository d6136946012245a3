// Kernel instantiations of linear_wide_kernel.h (dact_lin); see linear_wide.hip for the dispatcher.
#define ARDAE_WIDE_INST_TU
#include "linear_wide_kernel.h"

namespace ardae {
namespace wide {
ARDAE_WIDE_FOR_DACT_FLAGS(ARDAE_WIDE_INSTANTIATE, ACT_NONE)
ARDAE_WIDE_FOR_DACT_FLAGS(ARDAE_WIDE_INSTANTIATE, ACT_RELU)
}  // namespace wide
}  // namespace ardae
