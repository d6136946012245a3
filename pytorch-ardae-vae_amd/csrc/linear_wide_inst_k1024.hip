// Kernel instantiations of linear_wide_kernel.h (K = 1024, rolling slab window); see linear_wide.hip for the dispatcher.
#define ARDAE_WIDE_INST_TU
#include "linear_wide_kernel.h"

namespace ardae {
namespace wide {
ARDAE_WIDE_FOR_K1024(ARDAE_WIDE_INSTANTIATE)
}  // namespace wide
}  // namespace ardae
