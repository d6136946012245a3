// Kernel instantiation of linear_fchain_kernel.h (dactq); see linear_wide.hip for the dispatcher.
#define ARDAE_WIDE_INST_TU
#define ARDAE_FCHAIN_INST_TU
#include "linear_fchain_kernel.h"

namespace ardae {
namespace wide {
ARDAE_FCHAIN_INSTANTIATE(EPI_DACT, ACT_SOFTPLUS, true, false)
}  // namespace wide
}  // namespace ardae
