// Small kernels of the evaluation / visualisation side of the model surface (not on the timed train step):
//   * relaxed-Bernoulli decoder sample   models/reparam.py:111-158 (BernoulliDistribution.sample_logistic_sigmoid, T = 1),
//                                        returned by ImplicitPosteriorVAE.forward / .generate (ivae/mnist.py:188-199,300,316)
//   * Gaussian decoder sample            models/reparam.py:42-51 (sample_gaussian), ivae/toy.py:725-737
//   * batched Cholesky factorisation     the proposal of the IWAE evaluator: MultivariateNormal(mu, cov) at ivae/mnist.py:397-406
//                                        factorises cov per image; here all images in one launch (z_dim <= 64)
#include "ardae_hip.h"
#include "common.h"

namespace ardae {
namespace {

// sample = sigmoid((logit + log(u / (1 - u) + 1e-20)) / T), mean = sigmoid(logit); u ~ U[0,1) supplied by the caller
__global__ void relaxed_bernoulli_kernel(const float* __restrict__ logit, const float* __restrict__ u, int64_t n, float inv_t,
                                         float* __restrict__ sample, float* __restrict__ mean) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float l = logit[i];
  if (sample) {
    const float v = u[i];
    const float y = l + logf(v / (1.f - v) + 1e-20f);
    sample[i] = 1.f / (1.f + expf(-y * inv_t));
  }
  if (mean) mean[i] = 1.f / (1.f + expf(-l));
}

// sample = mu + exp(logvar / 2) * eps
__global__ void gaussian_sample_kernel(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps,
                                       int64_t n, float* __restrict__ sample) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sample[i] = mu[i] + expf(0.5f * logvar[i]) * eps[i];
}

// One 64-thread workgroup per matrix, right-looking, matrix in LDS; thread i owns row i.  A not positive definite gives NaN
// on and below the failing pivot (the caller checks, as torch.linalg.cholesky would raise).
__global__ __launch_bounds__(64) void cholesky_kernel(const float* __restrict__ A, int n, float* __restrict__ Lout) {
  __shared__ float a[64][65];
  const int i = threadIdx.x;
  const float* Ab = A + (size_t)blockIdx.x * n * n;
  float* Lb = Lout + (size_t)blockIdx.x * n * n;
  if (i < n)
    for (int j = 0; j < n; ++j) a[i][j] = Ab[(size_t)i * n + j];
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const float d = sqrtf(a[k][k]);
    __syncthreads();
    if (i == k) a[k][k] = d;
    if (i > k && i < n) a[i][k] = a[i][k] / d;
    __syncthreads();
    if (i > k && i < n)
      for (int j = k + 1; j <= i; ++j) a[i][j] -= a[i][k] * a[j][k];
    __syncthreads();
  }
  if (i < n)
    for (int j = 0; j < n; ++j) Lb[(size_t)i * n + j] = j <= i ? a[i][j] : 0.f;
}

}  // namespace
}  // namespace ardae

using namespace ardae;

extern "C" {

int ardae_relaxed_bernoulli(const float* logit, const float* u, int64_t n, float temperature, float* sample, float* mean,
                            void* stream) {
  ARDAE_CHECK_ARG(logit && n > 0 && (sample || mean) && (!sample || u) && temperature > 0.f, "relaxed_bernoulli: bad arguments");
  hipLaunchKernelGGL(relaxed_bernoulli_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logit, u, n,
                     1.f / temperature, sample, mean);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int ardae_gaussian_sample(const float* mu, const float* logvar, const float* eps, int64_t n, float* sample, void* stream) {
  ARDAE_CHECK_ARG(mu && logvar && eps && sample && n > 0, "gaussian_sample: bad arguments");
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps, n,
                     sample);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int ardae_cholesky_batched(const float* A, int batch, int n, float* L, void* stream) {
  ARDAE_CHECK_ARG(A && L && batch > 0 && n >= 1 && n <= 64, "cholesky_batched: need 1 <= n <= 64 (got n=%d, batch=%d)", n, batch);
  hipLaunchKernelGGL(cholesky_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, A, n, L);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Scalar log channel + static-binarised batch source
// ---------------------------------------------------------------------------------------------------------------------
namespace ardae {
namespace {

struct StepStateView {     // layout of the device step state (elementwise.hip::StepState, ARDAE_STEP_STATE_BYTES)
  uint64_t rng_offset;
  int64_t adam_step;
  float adam_step_size, adam_sqrt_bc2;
};

// One record of ARDAE_LOG_RECORD_FLOATS floats per step into ring slot (iter - 1) % capacity; iter = the step state's Adam t.
__global__ __launch_bounds__(256) void log_scalars_kernel(const float* __restrict__ cdae_loss, const float* __restrict__ model_losses,
                                                          const float* __restrict__ std_b, int B, float beta, float d_lr,
                                                          const StepStateView* __restrict__ state, float* __restrict__ ring, int capacity) {
  __shared__ float s_sum[256], s_max[256], s_min[256];
  const int t = threadIdx.x;
  float sm = 0.f, mx = -INFINITY, mn = INFINITY;
  for (int i = t; i < B; i += 256) {
    const float v = std_b[i];
    sm += v; mx = fmaxf(mx, v); mn = fminf(mn, v);
  }
  s_sum[t] = sm; s_max[t] = mx; s_min[t] = mn;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { s_sum[t] += s_sum[t + o]; s_max[t] = fmaxf(s_max[t], s_max[t + o]); s_min[t] = fminf(s_min[t], s_min[t + o]); }
    __syncthreads();
  }
  if (t != 0) return;
  const int64_t iter = state->adam_step;
  float* r = ring + (size_t)((iter - 1) % capacity) * ARDAE_LOG_RECORD_FLOATS;
  r[0] = __int_as_float((int)(iter & 0x7fffffff));
  r[1] = model_losses[0]; r[2] = model_losses[1]; r[3] = model_losses[2];
  r[4] = beta;
  r[5] = cdae_loss[0];
  r[6] = s_sum[0] / (float)B; r[7] = s_max[0]; r[8] = s_min[0];
  r[9] = d_lr;
  r[10] = __int_as_float((int)(iter >> 31));
}

// out[b] = table[idx[b]]  (rows of D floats)
__global__ void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, int D, float* __restrict__ out) {
  const float* src = table + (size_t)idx[blockIdx.x] * D;
  float* dst = out + (size_t)blockIdx.x * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) dst[c] = src[c];
}

}  // namespace
}  // namespace ardae

extern "C" {

int ardae_log_scalars(const float* cdae_loss, const float* model_losses, const float* std_b, int B, float beta, float d_lr,
                      const void* state, float* ring, int capacity, void* stream) {
  ARDAE_CHECK_ARG(cdae_loss && model_losses && std_b && state && ring && B > 0 && capacity > 0, "log_scalars: bad arguments");
  hipLaunchKernelGGL(log_scalars_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, cdae_loss, model_losses, std_b, B, beta, d_lr,
                     (const StepStateView*)state, ring, capacity);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int ardae_gather_rows(const float* table, const int64_t* idx, int B, int D, float* out, void* stream) {
  ARDAE_CHECK_ARG(table && idx && out && B > 0 && D > 0, "gather_rows: bad arguments");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, table, idx, D, out);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
