// Kernel instantiation of linear_fchain_kernel.h (act); see linear_wide.hip for the dispatcher.
#define ARDAE_WIDE_INST_TU
#define ARDAE_FCHAIN_INST_TU
#include "linear_fchain_kernel.h"

namespace ardae {
namespace wide {
ARDAE_FCHAIN_INSTANTIATE(EPI_ACT, ACT_SOFTPLUS, true, true)
}  // namespace wide
}  // namespace ardae
