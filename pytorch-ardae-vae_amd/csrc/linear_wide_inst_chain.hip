// Kernel instantiations of linear_wide_kernel.h (chain); see linear_wide.hip for the dispatcher.
#define ARDAE_WIDE_INST_TU
#include "linear_wide_kernel.h"

namespace ardae {
namespace wide {
ARDAE_WIDE_FOR_GEOS(ARDAE_WIDE_INSTANTIATE, EPI_CHAIN, ACT_SOFTPLUS, false, false)
}  // namespace wide
}  // namespace ardae
