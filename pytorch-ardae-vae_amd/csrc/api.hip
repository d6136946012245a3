// extern "C" surface of libardae_hip.so (declarations + reference citations: include/ardae_hip.h).
#include <stdarg.h>

#include "ardae_hip.h"
#include "common.h"
#include "linear.h"

namespace ardae {
static thread_local char g_last_error[512] = "";
void set_last_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}
}  // namespace ardae

using namespace ardae;

extern "C" {

const char* ardae_last_error(void) { return g_last_error; }
int ardae_abi_version(void) { return ARDAE_ABI_VERSION; }

size_t ardae_packed_floats(int nout, int k) { return packed_floats(nout, k); }
int ardae_linear_row_tiles(int M, int nout) { return linear_row_tiles(M, nout); }
int ardae_linear_col_panels(int nout) { return linear_col_panels(nout); }

int ardae_pack_weight(const float* W, int ldw, int nout, int k, int transpose, float* out, void* stream) {
  return launch_pack_weight(W, ldw, nout, k, transpose != 0, out, (hipStream_t)stream);
}

int ardae_linear(const ardae_linear_args* args, int epilogue, void* stream) {
  ARDAE_CHECK_ARG(args != nullptr, "ardae_linear: args is NULL");
  return launch_linear(*args, epilogue, (hipStream_t)stream);
}

}  // extern "C"
