// Kernel instantiations of linear_chain_kernel.h (act); see linear_chain.hip for the dispatcher.
#define ARDAE_WIDE_INST_TU
#define ARDAE_CHAIN_INST_TU
#include "linear_chain_kernel.h"

namespace ardae {
namespace wide {
ARDAE_CHAIN_FOR_ACT(ARDAE_CHAIN_INSTANTIATE, ACT_SOFTPLUS)
ARDAE_CHAIN_FOR_ACT(ARDAE_CHAIN_INSTANTIATE, ACT_RELU)
}  // namespace wide
}  // namespace ardae
