// Kernel instantiations of linear_chain_kernel.h (chain); see linear_chain.hip for the dispatcher.
#define ARDAE_WIDE_INST_TU
#define ARDAE_CHAIN_INST_TU
#include "linear_chain_kernel.h"

namespace ardae {
namespace wide {
ARDAE_CHAIN_INSTANTIATE(EPI_CHAIN, ACT_SOFTPLUS, false, false, false)
}  // namespace wide
}  // namespace ardae
