// extern "C" surface of libardae_hip.so (declarations + reference citations: include/ardae_hip.h).
#include <stdarg.h>

#include "ardae_hip.h"
#include "common.h"
#include "elementwise.h"
#include "linear.h"
#include "wgrad.h"

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
int ardae_linear_col_panels(int M, int nout) { return linear_col_panels(M, nout); }

int ardae_pack_weight(const float* W, int ldw, int nout, int k, int transpose, float* out, void* stream) {
  return launch_pack_weight(W, ldw, nout, k, transpose != 0, out, (hipStream_t)stream);
}

int ardae_linear(const ardae_linear_args* args, int epilogue, void* stream) {
  ARDAE_CHECK_ARG(args != nullptr, "ardae_linear: args is NULL");
  return launch_linear(*args, epilogue, (hipStream_t)stream);
}

int ardae_linear_wide_layers_eligible(const ardae_linear_args* layers, int nl, int epilogue) {
  return layers && linear_wide_layers_eligible(layers, nl, epilogue) ? 1 : 0;
}
int ardae_linear_wide_layers(const ardae_linear_args* layers, int nl, int epilogue, void* stream) {
  ARDAE_CHECK_ARG(layers != nullptr, "ardae_linear_wide_layers: layers is NULL");
  return launch_linear_wide_layers(layers, nl, epilogue, (hipStream_t)stream);
}
int ardae_linear_chain_eligible(const ardae_linear_args* layers, int nl, int epilogue) {
  return (layers != nullptr && linear_chain_eligible(layers, nl, epilogue)) ? 1 : 0;
}
int ardae_linear_chain(const ardae_linear_args* layers, int nl, int epilogue, void* stream) {
  ARDAE_CHECK_ARG(layers != nullptr, "ardae_linear_chain: layers is NULL");
  return launch_linear_chain(layers, nl, epilogue, (hipStream_t)stream);
}

int ardae_wgrad_splits(int M, int O, int I, int nproblems_hint) { return wgrad_splits(M, O, I, nproblems_hint); }
int ardae_wgrad_batch(const ardae_wgrad_problem* problems, int nproblems, void* stream) {
  return launch_wgrad_batch(problems, nproblems, (hipStream_t)stream);
}

int ardae_latent_perturb(const float* latent, const float* z0, const float* xi, const float* eps, int B, int nz, int z,
                         float std_scale, float delta, float* xbar, float* sigma, float* std_b, void* stream) {
  return launch_latent_perturb(latent, z0, xi, eps, B, nz, 1, z, std_scale, delta, xbar, sigma, std_b, (hipStream_t)stream);
}
int ardae_latent_perturb_nstd(const float* latent, const float* z0, const float* xi, const float* eps, int B, int nz, int nstd, int z,
                              float std_scale, float delta, float* xbar, float* sigma, float* std_b, void* stream) {
  return launch_latent_perturb(latent, z0, xi, eps, B, nz, nstd, z, std_scale, delta, xbar, sigma, std_b, (hipStream_t)stream);
}
int ardae_latent_perturb_draw_ok(int nz, int nstd, int z) { return latent_perturb_draw_ok(nz, nstd, z) ? 1 : 0; }
int ardae_latent_perturb_draw(const float* latent, const float* z0, int B, int nz, int z, float std_scale, float delta, uint64_t seed,
                              uint64_t offset_xi, uint64_t offset_eps, const void* state, uint64_t first_row, float* xbar, float* sigma,
                              float* eps_out, float* std_b, void* stream) {
  return launch_latent_perturb_draw(latent, z0, B, nz, z, std_scale, delta, seed, offset_xi, offset_eps, state, first_row, xbar, sigma, eps_out,
                                    std_b, (hipStream_t)stream);
}
int ardae_center_scale(const float* latent, const float* z0, int B, int nz, int z, float std_scale, float* u, void* stream) {
  return launch_center_scale(latent, z0, B, nz, z, std_scale, u, (hipStream_t)stream);
}
int ardae_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  return launch_philox_normal(out, n, seed, offset, (hipStream_t)stream);
}
int ardae_philox_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  return launch_philox_uniform(out, n, seed, offset, (hipStream_t)stream);
}
int ardae_bernoulli(const float* p, int64_t rows, int cols, float* out, uint64_t seed, uint64_t offset, void* stream) {
  return launch_bernoulli(p, rows, cols, out, seed, offset, (hipStream_t)stream);
}
int ardae_adam_ref_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, int64_t n,
                        double lr, double beta1, double beta2, double eps, int step, void* stream) {
  return launch_adam_ref(p, g, exp_avg, exp_avg_sq, max_exp_avg_sq, n, lr, beta1, beta2, eps, step, (hipStream_t)stream);
}
int ardae_step_state_advance(void* state, uint64_t rng_inc, double lr, double beta1, double beta2, void* stream) {
  return launch_step_state_advance(state, rng_inc, lr, beta1, beta2, (hipStream_t)stream);
}
int ardae_philox_normal_dev(float* out, int64_t n, uint64_t seed, const void* state, uint64_t offset_add, void* stream) {
  return launch_philox_normal_dev(out, n, seed, state, offset_add, (hipStream_t)stream);
}
int ardae_philox_normal_at(float* out, int64_t n, uint64_t seed, uint64_t offset, const void* state, uint64_t first_element, void* stream) {
  return launch_philox_normal_at(out, n, seed, offset, state, first_element, (hipStream_t)stream);
}
int ardae_adam_ref_step_dev(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, int64_t n,
                            double beta1, double beta2, double eps, const void* state, void* stream) {
  return launch_adam_ref_dev(p, g, exp_avg, exp_avg_sq, max_exp_avg_sq, n, beta1, beta2, eps, state, (hipStream_t)stream);
}
int ardae_rmsprop_step(float* p, const float* g, float* square_avg, float* momentum_buffer, int64_t n, double lr,
                       double alpha, double eps, double momentum, void* stream) {
  return launch_rmsprop(p, g, square_avg, momentum_buffer, n, lr, alpha, eps, momentum, (hipStream_t)stream);
}
/* torch.optim.SGD as constructed at ivae_ardae.py:546-547,613-614 (no momentum, no weight decay): p -= lr g */
int ardae_sgd_step(float* p, const float* g, int64_t n, double lr, void* stream) {
  return launch_axpy(g, n, (float)-lr, p, (hipStream_t)stream);
}

}  // extern "C"
