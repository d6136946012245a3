// Kernel instantiations of linear_wide_kernel.h (layer-major multi-layer launches: forward and forward-mode runs); dispatcher: linear_wide.hip.
#define ARDAE_WIDE_INST_TU
#include <string.h>

#include "linear_wide_kernel.h"

namespace ardae {
namespace wide {
ARDAE_WIDE_LAYERS_INSTANTIATE(8, 4, 2, EPI_ACT, ACT_SOFTPLUS, false, false)
ARDAE_WIDE_LAYERS_INSTANTIATE(8, 4, 2, EPI_ACT, ACT_RELU, false, false)
ARDAE_WIDE_LAYERS_INSTANTIATE(8, 4, 2, EPI_CHAIN, ACT_SOFTPLUS, false, false)
}  // namespace wide
}  // namespace ardae
