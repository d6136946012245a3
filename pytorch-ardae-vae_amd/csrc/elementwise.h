// Bandwidth-bound helper kernels of the AR-DAE-VAE step (latent statistics, noise, reductions, optimisers, losses).
#pragma once
#include "ardae_hip.h"
#include "common.h"

namespace ardae {

// ivae_ardae.py:753-761 + models/graddae/mlp.py:21-23:
//   u = s (z - z0[b]);  std_b = delta * mean_d( std_unbiased_over_nz(u[:, d]) );
//   sigma[b, i] = std_b * xi[b, i];  xbar = u + sigma * eps
int launch_latent_perturb(const float* latent, const float* z0, const float* xi, const float* eps, int B, int nz, int nstd, int zd,
                          float std_scale, float delta, float* xbar, float* sigma, float* std_b, hipStream_t st);
// the same with xi and eps DRAWN IN THE KERNEL (Philox keyed like launch_philox_normal_at: element i of this rank's rows = element
// first_row (* zd) + i of the global draws (seed, off_xi / off_eps [+ state])); eps_out receives the eps rows (the DAE loss reads them)
bool latent_perturb_draw_ok(int nz, int nstd, int zd);
int launch_latent_perturb_draw(const float* latent, const float* z0, int B, int nz, int zd, float std_scale, float delta, uint64_t seed,
                               uint64_t off_xi, uint64_t off_eps, const void* state, uint64_t first_row, float* xbar, float* sigma,
                               float* eps_out, float* std_b, hipStream_t st);
// the same with the first layer of the score network's input encoder on the perturbed rows: a_1 = act(xbar A_1^T + b_1) (wp1: packed A_1)
bool latent_perturb_draw_fwd_ok(int nz, int zd, int h, int act);
int launch_latent_perturb_draw_fwd(const float* latent, const float* z0, int B, int nz, int zd, float std_scale, float delta, uint64_t seed,
                                   uint64_t off_xi, uint64_t off_eps, const void* state, uint64_t first_row, float* xbar, float* sigma,
                                   float* eps_out, float* std_b, const float* wp1, const float* bias1, int h, int act, float* a1_out,
                                   hipStream_t st);
// u = s (z - z0[b]) only (VAE phase, sigma = 0: ivae_ardae.py:827)
int launch_center_scale(const float* latent, const float* z0, int B, int nz, int zd, float std_scale, float* u,
                        hipStream_t st);
// out[g][c] = scale * sum_{r < rows_per_group} in[g*rows_per_group + r][c]
int launch_segment_sum(const float* in, int ld, int groups, int rows_per_group, int cols, float scale, float* out, int ldout,
                       hipStream_t st);
// out[0] = scale * sum_i in[i]   (fixed order)
int launch_sum_scale(const float* in, int n, float scale, float* out, hipStream_t st);
// dst[i] = src[i*stride]
int launch_gather_strided(const float* src, int stride, int n, float* dst, hipStream_t st);
// Philox4x32-10 + Box-Muller standard normals; element i depends only on (seed, offset, i)
int launch_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t st);
// device-resident step state (graph-replayable step): see ardae_step_state_advance in ardae_hip.h
int launch_philox_normal_dev(float* out, int64_t n, uint64_t seed, const void* state, uint64_t offset_add, hipStream_t st);
// elements [first_element, first_element + n) of the draw (seed, offset [+ state]); state may be null
int launch_philox_normal_at(float* out, int64_t n, uint64_t seed, uint64_t offset, const void* state, uint64_t first_element, hipStream_t st);
int launch_step_state_advance(void* state, uint64_t rng_inc, double lr, double beta1, double beta2, hipStream_t st);
int launch_adam_ref_dev(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, double beta1, double beta2, double eps,
                        const void* state, hipStream_t st);
int launch_philox_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t st);
int launch_bernoulli(const float* p, int64_t rows, int cols, float* out, uint64_t seed, uint64_t offset, hipStream_t st);

// utils/optim.py:86-106 (old-style Adam: eps added before the bias correction)
int launch_adam_ref(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, double lr, double beta1, double beta2,
                    double eps, int step, hipStream_t st);
// torch.optim.RMSprop(lr, momentum) as constructed at ivae_ardae.py:625-626 (alpha, eps defaults; not centred)
int launch_rmsprop(float* p, const float* g, float* sq, float* buf, int64_t n, double lr, double alpha, double eps,
                   double momentum, hipStream_t st);

// y += alpha*x (the entropy-gradient seed of ivae_ardae.py:834 added to dL/dz)
int launch_axpy(const float* x, int64_t n, float alpha, float* y, hipStream_t st);
// y[i] = v;  y = x;  y[r][c] = x[r][c] (row strides ldx / ldy) - kernels instead of hipMemsetAsync / hipMemcpyAsync / hipMemcpy2DAsync: as
// graph nodes those are not ordered against their neighbour kernels when a linear graph goes out as one AQL batch (elementwise.hip)
int launch_fill(float* y, int64_t n, float v, hipStream_t st);
int launch_copy(const float* x, int64_t n, float* y, hipStream_t st);
int launch_copy2d(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t cols, hipStream_t st);
// y = alpha*x + beta (the encoder's 2x-1 rescale, ivae/mnist.py:81)
int launch_affine(const float* x, int64_t n, float alpha, float beta, float* y, hipStream_t st);
// Row losses of ImplicitPosteriorVAE.loss (ivae/mnist.py:240-249, toy.py:777-786) and, when write_grads, their gradients:
//   kind 0: rec = sum_d BCE_with_logits(o0, x)      (utils/vae.py:21-30);  do0 = gscale*(sigmoid(o0) - x)
//   kind 1: rec = .5 sum_d (o1 + (x-o0)^2/exp(o1) + log 2pi) (utils/vae.py:36-52); do0 = -gscale (x-o0)/exp(o1), do1 = .5 gscale (1-(x-o0)^2/exp(o1))
//   pri = .5 sum_d (z^2 + log 2pi) (utils/energy.py:69-77);  dzq = gscale*beta*z + dz_extra
// x is [rows/nz, D] (row r belongs to image r/nz).
int launch_vae_loss(int kind, const float* o0, const float* o1, const float* x, const float* z, int rows, int nz, int D, int zd,
                    float beta, int write_grads, float gscale, const float* dz_extra, float* rec_row, float* pri_row, float* do0,
                    float* do1, float* dzq, hipStream_t st);
// losses[0] = mean(rec + beta*pri), losses[1] = mean(rec), losses[2] = mean(pri)
int launch_vae_loss_finalize(const float* rec_row, const float* pri_row, int rows, float beta, float* losses, hipStream_t st);

}  // namespace ardae
