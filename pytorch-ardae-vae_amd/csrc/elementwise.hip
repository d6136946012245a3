// Bandwidth-bound helper kernels (gfx950).  See elementwise.h for the operator definitions + reference lines.
#include "elementwise.h"
#include "linear.h"

namespace ardae {
namespace {

// ------------------------------------------------------------------------------------------ Philox4x32-10
struct Philox {
  uint32_t c[4];
  uint32_t k[2];
};
__device__ __forceinline__ void philox_round(Philox& s) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t hi0 = __umulhi(M0, s.c[0]), lo0 = M0 * s.c[0];
  const uint32_t hi1 = __umulhi(M1, s.c[2]), lo1 = M1 * s.c[2];
  const uint32_t n0 = hi1 ^ s.c[1] ^ s.k[0], n1 = lo1, n2 = hi0 ^ s.c[3] ^ s.k[1], n3 = lo0;
  s.c[0] = n0; s.c[1] = n1; s.c[2] = n2; s.c[3] = n3;
  s.k[0] += 0x9E3779B9u;
  s.k[1] += 0xBB67AE85u;
}
__device__ __forceinline__ void philox4(uint64_t seed, uint64_t offset, uint64_t idx, uint32_t out[4]) {
  Philox s;
  s.c[0] = (uint32_t)idx; s.c[1] = (uint32_t)(idx >> 32);
  s.c[2] = (uint32_t)offset; s.c[3] = (uint32_t)(offset >> 32);
  s.k[0] = (uint32_t)seed; s.k[1] = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(s);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = s.c[i];
}
__device__ __forceinline__ float u01_open(uint32_t x) { return ((x >> 8) + 1u) * (1.0f / 16777216.0f); }  // (0,1]
// the four standard normals of Philox counter `idx` (Box-Muller on the two word pairs): the ONE definition every draw uses -
// the stand-alone kernel and the draws fused into their consumers give the same numbers for the same (seed, offset, element)
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t offset, uint64_t idx, float v[4]) {
  uint32_t r[4];
  philox4(seed, offset, idx, r);
  // Box-Muller on the hardware transcendental units (round 4): v_log_f32 is log2, v_sin_f32 / v_cos_f32 take their argument in
  // REVOLUTIONS, so sin(2 pi u) is one instruction with no range reduction.  ~12 vector-ALU instructions per pair instead of the ~150
  // of logf + sincosf (the draws fused into latent_perturb_reg_kernel made that kernel issue-bound: SQ_WAIT_INST_ANY = a third of its
  // wave cycles).  Absolute accuracy ~1e-6 - these are noise samples; what matters is that EVERY draw uses this one definition.
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float rad = __builtin_amdgcn_sqrtf(-1.38629436111989062f * __builtin_amdgcn_logf(u01_open(r[2 * h])));      // sqrt(-2 ln u)
    const float rev = u01_open(r[2 * h + 1]);
    v[2 * h] = rad * __builtin_amdgcn_cosf(rev);
    v[2 * h + 1] = rad * __builtin_amdgcn_sinf(rev);
  }
}

// Device-resident step state (ardae_step_state_advance): lets a captured HIP graph replay the step with fresh noise and
// the right Adam bias correction - kernel arguments are frozen at capture, this block is not.
struct StepState {
  uint64_t rng_offset;     // base offset of this step's Philox draws
  int64_t adam_step;       // t of utils/optim.py:84
  float adam_step_size;    // lr / (1 - beta1^t)
  float adam_sqrt_bc2;     // sqrt(1 - beta2^t)
};

__global__ void step_state_advance_kernel(StepState* s, uint64_t rng_inc, double lr, double beta1, double beta2) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  s->rng_offset += rng_inc;
  const int64_t t = s->adam_step + 1;
  s->adam_step = t;
  const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);   // host formula of launch_adam_ref, in double
  s->adam_step_size = (float)(lr / bc1);
  s->adam_sqrt_bc2 = (float)sqrt(bc2);
}

// q0: first counter of this launch - a rank that owns rows [r0, r1) of a draw generates elements [first, first + n) of the
// GLOBAL draw (first = 4 q0), so the numbers do not depend on how the rows are partitioned over ranks
__global__ void philox_normal_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset, const StepState* state, uint64_t q0) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one counter = 4 normals
  if (q * 4 >= n) return;
  if (state) offset += state->rng_offset;
  float v[4];
  philox_normal4(seed, offset, q0 + (uint64_t)q, v);
  if (q * 4 + 4 <= n && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
    *reinterpret_cast<f32x4*>(out + q * 4) = f32x4{v[0], v[1], v[2], v[3]};
  } else {
    for (int i = 0; i < 4 && q * 4 + i < n; ++i) out[q * 4 + i] = v[i];
  }
}

__global__ void philox_uniform_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q * 4 >= n) return;
  uint32_t r[4];
  philox4(seed, offset, (uint64_t)q, r);
  for (int i = 0; i < 4 && q * 4 + i < n; ++i) out[q * 4 + i] = (r[i] >> 8) * (1.0f / 16777216.0f);   // [0,1)
}

// dynamic binarisation (datasets/mnist.py:36-40): out = Bernoulli(p[col])
__global__ void bernoulli_kernel(const float* __restrict__ p, int64_t rows, int cols, float* __restrict__ out, uint64_t seed,
                                 uint64_t offset) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = rows * cols;
  if (q * 4 >= n) return;
  uint32_t r[4];
  philox4(seed, offset, (uint64_t)q, r);
  for (int i = 0; i < 4 && q * 4 + i < n; ++i) {
    const int64_t e = q * 4 + i;
    out[e] = ((r[i] >> 8) * (1.0f / 16777216.0f)) < p[e % cols] ? 1.f : 0.f;
  }
}

// ------------------------------------------------------------------------------------------ latent statistics
// one workgroup per image; thread (rg, d): d = t % zp, rg = t / zp
__global__ __launch_bounds__(256) void latent_perturb_kernel(const float* __restrict__ latent, const float* __restrict__ z0,
                                                             const float* __restrict__ xi, const float* __restrict__ eps, int nz,
                                                             int nstd, int zd, int zp, float std_scale, float delta,
                                                             float* __restrict__ xbar, float* __restrict__ sigma,
                                                             float* __restrict__ std_b) {
  __shared__ float red[256];
  __shared__ float stat[256];
  const int b = blockIdx.x, t = threadIdx.x;
  const int d = t % zp, rg = t / zp, RG = 256 / zp;
  const float* lat = latent + (size_t)b * nz * zd;
  const float z0d = d < zd ? z0[(size_t)b * zd + d] : 0.f;
  // pass 1: mean of u over the nz samples
  float s = 0.f;
  if (d < zd)
    for (int r = rg; r < nz; r += RG) s += std_scale * (lat[(size_t)r * zd + d] - z0d);
  red[t] = s;
  __syncthreads();
  if (rg == 0) {
    float acc = 0.f;
    for (int g = 0; g < RG; ++g) acc += red[g * zp + d];
    stat[d] = acc / (float)nz;
  }
  __syncthreads();
  const float mean = stat[d];
  __syncthreads();
  // pass 2: unbiased variance
  float ss = 0.f;
  if (d < zd)
    for (int r = rg; r < nz; r += RG) {
      const float u = std_scale * (lat[(size_t)r * zd + d] - z0d) - mean;
      ss += u * u;
    }
  red[t] = ss;
  __syncthreads();
  if (rg == 0) {
    float acc = 0.f;
    for (int g = 0; g < RG; ++g) acc += red[g * zp + d];
    stat[d] = d < zd ? sqrtf(acc / (float)(nz - 1)) : 0.f;
  }
  __syncthreads();
  if (t == 0) {
    float acc = 0.f;
    for (int i = 0; i < zd; ++i) acc += stat[i];
    red[0] = delta * (acc / (float)zd);
  }
  __syncthreads();
  const float sb = red[0];
  if (t == 0) std_b[b] = sb;
  // perturb: every sample row is used nstd times (--train-nstd-cdae, ivae_ardae.py:759-767), each with its own sigma and eps
  const int nrow = nz * nstd;
  for (int e = t; e < nrow * zd; e += 256) {
    const int r = e / zd, dd = e - r * zd;
    const size_t row = (size_t)b * nrow + r;
    const float sg = sb * xi[row];
    const float u = std_scale * (lat[(size_t)(r / nstd) * zd + dd] - z0[(size_t)b * zd + dd]);
    xbar[row * zd + dd] = u + sg * eps[row * zd + dd];
    if (dd == 0) sigma[row] = sg;
  }
}

// Register-resident variant for nz * zd <= 256 * NV with zd a power of two: thread t owns elements t + 256 j (all of the same
// latent dimension d = t % zd, samples r = t / zd + (256 / zd) j - the same assignment and summation order as the kernel
// above, so the results are bit-identical), loaded ONCE with all NV loads in flight instead of three dependent passes.
// DRAW (north star: "fused per-sample Gaussian-perturb + sigma-scaling" with the draws made in the kernel): xi and eps are not read
// but GENERATED here - counter c of the global eps draw covers elements 4c .. 4c + 3, thread t computes counters t, t + 256, ... of
// its image into LDS, from where every thread picks the elements it owns (the same ownership as the injected-noise form, so the
// statistics are bit-identical for equal numbers); eps is also written out (the DAE-loss layer needs it: rho = sigma g + eps).
// Keyed exactly like ardae_philox_normal_at: element i of a rank's shard = element first + i of the global draw (seed, offset).
constexpr int LP_XPAD = 4;            // FWD: the xbar tile's row stride in LDS is z + 4 floats (conflict-free fragment reads)
struct PerturbDraw {
  uint64_t seed, off_xi, off_eps;     // Philox offsets of the two draws (the step state's base offset is added when state != null)
  const StepState* state;
  uint64_t first_row;                 // this rank's first row of the global [rows] / [rows, z] draws
  float* eps_out;
  // FWD: the first layer of the score network's input encoder on the rows just perturbed, a_1 = act(xbar A_1^T + b_1), in the same kernel
  const float* wp1;                   // packed image of A_1 [h, z]
  const float* bias1;                 // [h]
  float* a1_out;                      // [rows, h]
  int h;
};

// FWD > 0 (the activation id): north star "fused per-sample Gaussian-perturb + sigma-scaling + DAE-forward kernel": after the perturbation the
// image's rows xbar [nz, z] are still in the workgroup's LDS; its four waves multiply them by A_1 (FP32 MFMA, 32 x 32 blocks, K = z) and write
// a_1 = act(xbar A_1^T + b_1) [nz, h] (models/graddae/mlp.py:414-434: add_gaussian_noise, then inp_encode's first Linear + activation) - the
// separate K = z N-row launch and its re-read of xbar are gone.  Same MFMA order over k as the stand-alone layer kernels.
template <int NV, bool DRAW, int FWD = 0>
__global__ __launch_bounds__(256) void latent_perturb_reg_kernel(const float* __restrict__ latent, const float* __restrict__ z0,
                                                                 const float* __restrict__ xi, const float* __restrict__ eps, int nz,
                                                                 int zd, float std_scale, float delta, float* __restrict__ xbar,
                                                                 float* __restrict__ sigma, float* __restrict__ std_b, const PerturbDraw dr) {
  __shared__ float red[256];
  __shared__ float stat[256];
  extern __shared__ float nbuf[];    // DRAW: nz * zd normals (eps of this image) + nz (xi)
  const int b = blockIdx.x, t = threadIdx.x;
  const int d = t & (zd - 1), RG = 256 / zd;
  const size_t base = (size_t)b * nz * zd;
  const float* lat = latent + base;
  const float z0d = z0[(size_t)b * zd + d];
  const int per_image = nz * zd;                 // elements beyond it (last pass of a ragged image) are masked out
  float u[NV], ev[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) u[j] = (t + 256 * j < per_image) ? lat[t + 256 * j] : z0d;
  // FWD with few images: gridDim.y workgroups share an image - every one computes the image's statistics (from the latent rows alone, in the
  // same order: identical bits), then draws, perturbs and multiplies only ITS rows, passes j0 <= j < j1 (RG rows each)
  const int j0 = FWD ? (int)blockIdx.y * (per_image >> 8) / (int)gridDim.y : 0;
  const int j1 = FWD ? ((int)blockIdx.y + 1) * (per_image >> 8) / (int)gridDim.y : NV;
  const int RGc = 256 / zd, rows_sl = FWD ? (j1 - j0) * RGc : nz, xi_at = FWD ? rows_sl * (zd + LP_XPAD) : per_image;
  if (DRAW) {
    const uint64_t base_off = dr.state ? dr.state->rng_offset : 0;
    const uint64_t q_eps = (dr.first_row * (uint64_t)zd + base) >> 2;        // first counter of this image's eps block
    const int c_lo = 64 * j0, c_hi = FWD ? 64 * j1 : (per_image + 3) >> 2;   // one pass = 256 elements = 64 counters
    for (int c = c_lo + t; c < c_hi; c += 256) {
      float v[4];
      philox_normal4(dr.seed, dr.off_eps + base_off, q_eps + (uint64_t)c, v);
      *reinterpret_cast<f32x4*>(nbuf + 4 * (c - c_lo)) = f32x4{v[0], v[1], v[2], v[3]};
    }
    const uint64_t q_xi = (dr.first_row + (uint64_t)b * nz) >> 2;
    const int x_lo = (RGc * j0) >> 2, x_hi = FWD ? (RGc * j1) >> 2 : (nz + 3) >> 2;
    for (int c = x_lo + t; c < x_hi; c += 256) {
      float v[4];
      philox_normal4(dr.seed, dr.off_xi + base_off, q_xi + (uint64_t)c, v);
      *reinterpret_cast<f32x4*>(nbuf + xi_at + 4 * (c - x_lo)) = f32x4{v[0], v[1], v[2], v[3]};
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const bool mine = t + 256 * j < per_image && j >= j0 && j < j1;
      ev[j] = mine ? nbuf[t + 256 * (j - j0)] : 0.f;
      if (mine) dr.eps_out[base + t + 256 * j] = ev[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < NV; ++j) ev[j] = (t + 256 * j < per_image) ? eps[base + t + 256 * j] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    u[j] = std_scale * (u[j] - z0d);
    if (t + 256 * j < per_image) s += u[j];
  }
  red[t] = s;
  __syncthreads();
  if (t < zd) {
    float acc = 0.f;
    for (int g = 0; g < RG; ++g) acc += red[g * zd + t];
    stat[t] = acc / (float)nz;
  }
  __syncthreads();
  const float mean = stat[d];
  __syncthreads();
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const float c = u[j] - mean;
    if (t + 256 * j < per_image) ss += c * c;
  }
  red[t] = ss;
  __syncthreads();
  if (t < zd) {
    float acc = 0.f;
    for (int g = 0; g < RG; ++g) acc += red[g * zd + t];
    stat[t] = sqrtf(acc / (float)(nz - 1));
  }
  __syncthreads();
  if (t == 0) {
    float acc = 0.f;
    for (int i = 0; i < zd; ++i) acc += stat[i];
    red[0] = delta * (acc / (float)zd);
  }
  __syncthreads();
  const float sb = red[0];
  if (t == 0 && (!FWD || blockIdx.y == 0)) std_b[b] = sb;
  const int r0 = t / zd;
  if (FWD) __syncthreads();            // everybody has picked its eps values out of nbuf: it now receives the xbar tile
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    if (t + 256 * j < per_image && j >= j0 && j < j1) {
      const size_t row = (size_t)b * nz + r0 + RG * j;
      const float sg = sb * (DRAW ? nbuf[xi_at + r0 + RG * (j - j0)] : xi[row]);
      const float xb = u[j] + sg * ev[j];
      xbar[base + t + 256 * j] = xb;
      if (FWD) nbuf[(r0 + RG * (j - j0)) * (zd + LP_XPAD) + d] = xb;
      if (d == 0) sigma[row] = sg;
    }
  }
  if (FWD) {
    __syncthreads();
    // wave w: column blocks w, w + 4, ... of a_1; the block's weight fragments (<= 8 chunks of 8 k) stay in registers over the row blocks
    const int lane = t & 63, wave = t >> 6, l31 = lane & 31, hh = lane >> 5;
    const int kch = zd >> 3, nrb = rows_sl >> 5, ncb = dr.h >> 5, ld = zd + LP_XPAD;
    for (int cb = wave; cb < ncb; cb += 4) {
      f32x4 bv[8];
      const float* bp = dr.wp1 + (size_t)cb * kch * 256 + lane * 4;
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < kch) bv[c] = *reinterpret_cast<const f32x4*>(bp + (size_t)c * 256);
      const int col = cb * 32 + l31;
      const float bc = dr.bias1[col];
      for (int rb = 0; rb < nrb; ++rb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* arow = nbuf + (rb * 32 + l31) * ld + 4 * hh;
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (c < kch) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 8 * c);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[c][q], acc, 0, 0, 0);
          }
        float* yo = dr.a1_out + ((size_t)b * nz + RG * j0 + rb * 32 + 4 * hh) * dr.h + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) yo[(size_t)((r & 3) + 8 * (r >> 2)) * dr.h] = act_fwd<FWD>(acc[r] + bc);
      }
    }
  }
}

__global__ void center_scale_kernel(const float* __restrict__ latent, const float* __restrict__ z0, int64_t n, int nz, int zd,
                                    float std_scale, float* __restrict__ u) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int64_t row = e / zd;
  const int d = (int)(e - row * zd);
  u[e] = std_scale * (latent[e] - z0[(row / nz) * zd + d]);
}

// grid (groups, ceil(cols/64)); NT threads = NT/64 row lanes x 64 consecutive columns (256-B coalesced rows), eight rows
// in flight per thread.  NT = 256 for many groups (the chip is full anyway); NT = 1024 for the few-groups / many-rows case
// (the per-tile column sums of an N-row launch: one group of 2048 rows took 53 us on four 256-thread workgroups).
template <int NT>
__global__ __launch_bounds__(NT) void segment_sum_kernel(const float* __restrict__ in, int ld, int rows_per_group, int cols,
                                                         float scale, float* __restrict__ out, int ldout) {
  constexpr int RL = NT / 64;
  __shared__ float red[NT];
  const int g = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  if (c < cols) {
    const float* p = in + (size_t)g * rows_per_group * ld + c;
    int r = rl;
    for (; r + 7 * RL < rows_per_group; r += 8 * RL) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += p[(size_t)(r + j * RL) * ld];
    }
    for (; r < rows_per_group; r += RL) s[0] += p[(size_t)r * ld];
  }
  red[threadIdx.x] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (rl == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < RL; ++j) t += red[threadIdx.x + 64 * j];
    out[(size_t)g * ldout + c] = scale * t;
  }
}

__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ in, int n, float scale, float* __restrict__ out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += in[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

__global__ void gather_strided_kernel(const float* __restrict__ src, int stride, int n, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[(size_t)i * stride];
}

// ------------------------------------------------------------------------------------------ optimisers
__global__ void adam_ref_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                float* __restrict__ vmax, int64_t n, float beta1, float beta2, float eps, float step_size,
                                float sqrt_bc2, const StepState* state) {
  if (state) {   // coefficients of the current step from the device-resident state
    step_size = state->adam_step_size;
    sqrt_bc2 = state->adam_sqrt_bc2;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = m[i] * beta1 + (1.f - beta1) * gi;
    float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    if (vmax) {
      vi = fmaxf(vmax[i], vi);
      vmax[i] = vi;
    }
    const float denom = (sqrtf(vi) + eps) / sqrt_bc2;   // eps BEFORE the bias correction (utils/optim.py:102)
    p[i] = p[i] - step_size * (mi / denom);
  }
}

__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, float* __restrict__ buf,
                               int64_t n, float lr, float alpha, float eps, float momentum) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float s = sq[i] * alpha + (1.f - alpha) * gi * gi;
    sq[i] = s;
    const float avg = sqrtf(s) + eps;
    if (momentum > 0.f) {
      const float bi = buf[i] * momentum + gi / avg;
      buf[i] = bi;
      p[i] = p[i] - lr * bi;
    } else {
      p[i] = p[i] - lr * (gi / avg);
    }
  }
}

__global__ void axpy_kernel(const float* __restrict__ x, int64_t n, float alpha, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] += alpha * x[i];
}

// The library issues no hipMemsetAsync / hipMemcpyAsync: as nodes of a captured graph they are not ordered against the
// neighbouring kernel nodes when ROCm 7.2 submits a linear graph as one batch of AQL packets (seen on MI355X: the zero-noise
// block of encode(x, std=0) was read before it had been cleared in ~2 of 3 replays).  Fills and copies are kernels like the rest.
__global__ void fill_kernel(float* __restrict__ y, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = v;
}
__global__ void copy2d_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int64_t rows, int64_t cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i - r * cols;
    y[r * ldy + c] = x[r * ldx + c];
  }
}

__global__ void affine_kernel(const float* __restrict__ x, int64_t n, float alpha, float beta, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = alpha * x[i] + beta;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup per row
template <int KIND>
__global__ __launch_bounds__(256) void vae_loss_kernel(const float* __restrict__ o0, const float* __restrict__ o1,
                                                       const float* __restrict__ x, const float* __restrict__ z, int nz, int D, int zd,
                                                       float beta, int write_grads, float gscale, const float* __restrict__ dz_extra,
                                                       float* __restrict__ rec_row, float* __restrict__ pri_row, float* __restrict__ do0,
                                                       float* __restrict__ do1, float* __restrict__ dzq) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* xr = x + (size_t)(r / nz) * D;
  const size_t base = (size_t)r * D;
  const float LOG2PI = 1.8378770664093453f;
  float acc = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float t = xr[d];
    if (KIND == 0) {
      const float l = o0[base + d];
      const float e = __expf(-fabsf(l));
      acc += fmaxf(l, 0.f) - l * t + (e < 1e-4f ? e * (1.f - 0.5f * e) : __logf(1.f + e));
      if (write_grads) {
        const float sg = l >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        do0[base + d] = gscale * (sg - t);
      }
    } else {
      const float mu = o0[base + d], lv = o1[base + d];
      const float iv = __expf(-lv), df = t - mu;
      acc += 0.5f * (lv + df * df * iv + LOG2PI);
      if (write_grads) {
        do0[base + d] = -gscale * df * iv;
        do1[base + d] = 0.5f * gscale * (1.f - df * df * iv);
      }
    }
  }
  const float rec = block_sum_256(acc, red);
  float pa = 0.f;
  for (int d = threadIdx.x; d < zd; d += 256) {
    const float zv = z[(size_t)r * zd + d];
    pa += 0.5f * (zv * zv + LOG2PI);
    if (write_grads) dzq[(size_t)r * zd + d] = gscale * beta * zv + (dz_extra ? dz_extra[(size_t)r * zd + d] : 0.f);
  }
  const float pri = block_sum_256(pa, red);
  if (threadIdx.x == 0) {
    rec_row[r] = rec;
    pri_row[r] = pri;
  }
}

__global__ __launch_bounds__(256) void vae_loss_finalize_kernel(const float* __restrict__ rec_row, const float* __restrict__ pri_row,
                                                                int rows, float beta, float* __restrict__ losses) {
  __shared__ float red[4];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < rows; i += 256) {
    a += rec_row[i];
    b += pri_row[i];
  }
  const float ra = block_sum_256(a, red);
  const float rb = block_sum_256(b, red);
  if (threadIdx.x == 0) {
    const float inv = 1.f / (float)rows;
    losses[0] = (ra + beta * rb) * inv;
    losses[1] = ra * inv;
    losses[2] = rb * inv;
  }
}

inline int grid_for(int64_t n, int cap = 4096) {
  int64_t g = (n + 255) / 256;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace

// can the draws of xi / eps be made inside the perturbation kernel?  (register-resident form, whole Philox counters per image)
bool latent_perturb_draw_ok(int nz, int nstd, int zd) {
  int zp = 1;
  while (zp < zd) zp <<= 1;
  const int64_t per_image = (int64_t)nz * zd;
  return nstd == 1 && zp == zd && zd >= 4 && zd <= 256 && per_image >= 256 * 4 && per_image <= 256 * 32 && nz % 4 == 0;
}

int launch_latent_perturb_draw(const float* latent, const float* z0, int B, int nz, int zd, float std_scale, float delta, uint64_t seed,
                               uint64_t off_xi, uint64_t off_eps, const void* state, uint64_t first_row, float* xbar, float* sigma,
                               float* eps_out, float* std_b, hipStream_t st) {
  ARDAE_CHECK_ARG(latent && z0 && xbar && sigma && eps_out && std_b && B > 0, "latent_perturb_draw: null pointer");
  ARDAE_CHECK_ARG(latent_perturb_draw_ok(nz, 1, zd), "latent_perturb_draw: shape not eligible (nz=%d z=%d): draw separately", nz, zd);
  ARDAE_CHECK_ARG((first_row & 3) == 0, "latent_perturb_draw: first_row must be a multiple of 4 (one Philox counter = 4 normals)");
  const int64_t per_image = (int64_t)nz * zd;
  const PerturbDraw dr{seed, off_xi, off_eps, (const StepState*)state, first_row, eps_out};
  const size_t lds = (size_t)(per_image + nz) * sizeof(float);     // <= 33 KiB
#define ARDAE_LP_DRAW(NV_)                                                                                                              \
  hipLaunchKernelGGL((latent_perturb_reg_kernel<NV_, true>), dim3(B), dim3(256), lds, st, latent, z0, nullptr, nullptr, nz, zd, std_scale, \
                     delta, xbar, sigma, std_b, dr)
  if (per_image <= 256 * 8) ARDAE_LP_DRAW(8);
  else if (per_image <= 256 * 16) ARDAE_LP_DRAW(16);
  else ARDAE_LP_DRAW(32);
#undef ARDAE_LP_DRAW
  ARDAE_LAUNCH_CHECK();
  return 0;
}

bool latent_perturb_draw_fwd_ok(int nz, int zd, int h, int act) {
  return latent_perturb_draw_ok(nz, 1, zd) && zd % 8 == 0 && zd <= 64 && nz % 32 == 0 && ((int64_t)nz * zd) % 256 == 0 && h % 32 == 0 &&
         (act == ACT_RELU || act == ACT_SOFTPLUS) &&
         ((size_t)nz * (zd + LP_XPAD) + nz) * sizeof(float) <= 64 * 1024;
}

int launch_latent_perturb_draw_fwd(const float* latent, const float* z0, int B, int nz, int zd, float std_scale, float delta, uint64_t seed,
                                   uint64_t off_xi, uint64_t off_eps, const void* state, uint64_t first_row, float* xbar, float* sigma,
                                   float* eps_out, float* std_b, const float* wp1, const float* bias1, int h, int act, float* a1_out,
                                   hipStream_t st) {
  ARDAE_CHECK_ARG(latent && z0 && xbar && sigma && eps_out && std_b && wp1 && bias1 && a1_out && B > 0, "latent_perturb_draw_fwd: null pointer");
  ARDAE_CHECK_ARG(latent_perturb_draw_fwd_ok(nz, zd, h, act), "latent_perturb_draw_fwd: shape not eligible (nz=%d z=%d h=%d act=%d)", nz, zd, h, act);
  ARDAE_CHECK_ARG((first_row & 3) == 0, "latent_perturb_draw_fwd: first_row must be a multiple of 4 (one Philox counter = 4 normals)");
  const int64_t per_image = (int64_t)nz * zd;
  const PerturbDraw dr{seed, off_xi, off_eps, (const StepState*)state, first_row, eps_out, wp1, bias1, a1_out, h};
  // few images: S workgroups per image, rows split among them (a slice is a whole number of 32-row blocks), until the grid fills the chip
  const int rg = 256 / zd, passes = (int)(per_image >> 8);
  static const int target_wgs = debug_knob("ARDAE_LP_WGS") ? atoi(debug_knob("ARDAE_LP_WGS")) : 256;
  int S = 1;
  while (B * S < target_wgs && passes % (2 * S) == 0 && (passes / (2 * S)) * rg % 32 == 0) S *= 2;
  const size_t rows_sl = (size_t)(passes / S) * rg;
  const size_t lds = (rows_sl * (zd + LP_XPAD) + rows_sl) * sizeof(float);
#define ARDAE_LP_FWD(NV_, ACT_)                                                                                                            \
  hipLaunchKernelGGL((latent_perturb_reg_kernel<NV_, true, ACT_>), dim3(B, S), dim3(256), lds, st, latent, z0, nullptr, nullptr, nz, zd,    \
                     std_scale, delta, xbar, sigma, std_b, dr)
#define ARDAE_LP_FWD_NV(ACT_)                                                                                                              \
  do {                                                                                                                                     \
    if (per_image <= 256 * 8) ARDAE_LP_FWD(8, ACT_);                                                                                       \
    else if (per_image <= 256 * 16) ARDAE_LP_FWD(16, ACT_);                                                                                \
    else ARDAE_LP_FWD(32, ACT_);                                                                                                           \
  } while (0)
  if (act == ACT_RELU) ARDAE_LP_FWD_NV(ACT_RELU);
  else ARDAE_LP_FWD_NV(ACT_SOFTPLUS);
#undef ARDAE_LP_FWD_NV
#undef ARDAE_LP_FWD
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_latent_perturb(const float* latent, const float* z0, const float* xi, const float* eps, int B, int nz, int nstd, int zd,
                          float std_scale, float delta, float* xbar, float* sigma, float* std_b, hipStream_t st) {
  ARDAE_CHECK_ARG(latent && z0 && xi && eps && xbar && sigma && std_b, "latent_perturb: null pointer");
  ARDAE_CHECK_ARG(B > 0 && nz >= 2 && nstd >= 1 && zd >= 1 && zd <= 256,
                  "latent_perturb: need B>0, nz>=2 (unbiased std), nstd>=1, 1<=z<=256 (B=%d nz=%d nstd=%d z=%d)", B, nz, nstd, zd);
  int zp = 1;
  while (zp < zd) zp <<= 1;
  const int64_t per_image = (int64_t)nz * zd;
  const bool reg_ok = nstd == 1 && zp == zd && zd <= 256 && per_image >= 256 * 4;
  const PerturbDraw nodraw{};
#define ARDAE_LP_REG(NV_)                                                                                                                      \
  hipLaunchKernelGGL((latent_perturb_reg_kernel<NV_, false>), dim3(B), dim3(256), 0, st, latent, z0, xi, eps, nz, zd, std_scale, delta, xbar, \
                     sigma, std_b, nodraw)
  if (reg_ok && per_image <= 256 * 8) ARDAE_LP_REG(8);
  else if (reg_ok && per_image <= 256 * 16) ARDAE_LP_REG(16);
  else if (reg_ok && per_image <= 256 * 32) ARDAE_LP_REG(32);
  else if (reg_ok && per_image <= 256 * 64) ARDAE_LP_REG(64);
  else if (reg_ok && per_image <= 256 * 96) ARDAE_LP_REG(96);       // nz_cdae 625 of the shipped recipes: 79 values per thread
  else
    hipLaunchKernelGGL(latent_perturb_kernel, dim3(B), dim3(256), 0, st, latent, z0, xi, eps, nz, nstd, zd, zp, std_scale, delta, xbar,
                       sigma, std_b);
#undef ARDAE_LP_REG
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_center_scale(const float* latent, const float* z0, int B, int nz, int zd, float std_scale, float* u, hipStream_t st) {
  ARDAE_CHECK_ARG(latent && z0 && u && B > 0 && nz > 0 && zd > 0, "center_scale: bad arguments");
  const int64_t n = (int64_t)B * nz * zd;
  hipLaunchKernelGGL(center_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, latent, z0, n, nz, zd, std_scale, u);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_segment_sum(const float* in, int ld, int groups, int rows_per_group, int cols, float scale, float* out, int ldout,
                       hipStream_t st) {
  ARDAE_CHECK_ARG(in && out && groups > 0 && rows_per_group > 0 && cols > 0 && ld >= cols && ldout >= cols, "segment_sum: bad arguments");
  if ((int64_t)groups * ceil_div(cols, 64) < 256 && rows_per_group >= 128)
    hipLaunchKernelGGL(segment_sum_kernel<1024>, dim3(groups, ceil_div(cols, 64)), dim3(1024), 0, st, in, ld, rows_per_group, cols, scale, out, ldout);
  else
    hipLaunchKernelGGL(segment_sum_kernel<256>, dim3(groups, ceil_div(cols, 64)), dim3(256), 0, st, in, ld, rows_per_group, cols, scale, out, ldout);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_sum_scale(const float* in, int n, float scale, float* out, hipStream_t st) {
  ARDAE_CHECK_ARG(in && out && n > 0, "sum_scale: bad arguments");
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, in, n, scale, out);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_gather_strided(const float* src, int stride, int n, float* dst, hipStream_t st) {
  ARDAE_CHECK_ARG(src && dst && n > 0 && stride > 0, "gather_strided: bad arguments");
  hipLaunchKernelGGL(gather_strided_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, stride, n, dst);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t st) {
  ARDAE_CHECK_ARG(out && n > 0, "philox_normal: bad arguments");
  const int64_t q = (n + 3) / 4;
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, out, n, seed, offset,
                     (const StepState*)nullptr, (uint64_t)0);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_philox_normal_at(float* out, int64_t n, uint64_t seed, uint64_t offset, const void* state, uint64_t first_element, hipStream_t st) {
  ARDAE_CHECK_ARG(out && n > 0, "philox_normal_at: bad arguments");
  ARDAE_CHECK_ARG((first_element & 3) == 0, "philox_normal_at: first_element must be a multiple of 4 (one Philox counter = 4 normals)");
  const int64_t q = (n + 3) / 4;
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, out, n, seed, offset,
                     (const StepState*)state, first_element >> 2);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_philox_normal_dev(float* out, int64_t n, uint64_t seed, const void* state, uint64_t offset_add, hipStream_t st) {
  ARDAE_CHECK_ARG(state, "philox_normal_dev: bad arguments");
  return launch_philox_normal_at(out, n, seed, offset_add, state, 0, st);
}

int launch_step_state_advance(void* state, uint64_t rng_inc, double lr, double beta1, double beta2, hipStream_t st) {
  ARDAE_CHECK_ARG(state, "step_state_advance: null state");
  hipLaunchKernelGGL(step_state_advance_kernel, dim3(1), dim3(64), 0, st, (StepState*)state, rng_inc, lr, beta1, beta2);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_philox_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t st) {
  ARDAE_CHECK_ARG(out && n > 0, "philox_uniform: bad arguments");
  const int64_t q = (n + 3) / 4;
  hipLaunchKernelGGL(philox_uniform_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, out, n, seed, offset);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_bernoulli(const float* p, int64_t rows, int cols, float* out, uint64_t seed, uint64_t offset, hipStream_t st) {
  ARDAE_CHECK_ARG(p && out && rows > 0 && cols > 0, "bernoulli: bad arguments");
  const int64_t q = (rows * cols + 3) / 4;
  hipLaunchKernelGGL(bernoulli_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, p, rows, cols, out, seed, offset);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_adam_ref(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, double lr, double beta1, double beta2,
                    double eps, int step, hipStream_t st) {
  ARDAE_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam_ref: bad arguments");
  const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
  hipLaunchKernelGGL(adam_ref_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, vmax, n, (float)beta1, (float)beta2,
                     (float)eps, (float)(lr / bc1), (float)sqrt(bc2), (const StepState*)nullptr);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_adam_ref_dev(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, double beta1, double beta2, double eps,
                        const void* state, hipStream_t st) {
  ARDAE_CHECK_ARG(p && g && m && v && n > 0 && state, "adam_ref_dev: bad arguments");
  hipLaunchKernelGGL(adam_ref_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, vmax, n, (float)beta1, (float)beta2,
                     (float)eps, 0.f, 1.f, (const StepState*)state);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_rmsprop(float* p, const float* g, float* sq, float* buf, int64_t n, double lr, double alpha, double eps, double momentum,
                   hipStream_t st) {
  ARDAE_CHECK_ARG(p && g && sq && n > 0 && (momentum <= 0.0 || buf), "rmsprop: bad arguments");
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, g, sq, buf, n, (float)lr, (float)alpha, (float)eps,
                     (float)momentum);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_axpy(const float* x, int64_t n, float alpha, float* y, hipStream_t st) {
  ARDAE_CHECK_ARG(x && y && n > 0, "axpy: bad arguments");
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, n, alpha, y);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_fill(float* y, int64_t n, float v, hipStream_t st) {
  ARDAE_CHECK_ARG(y && n > 0, "fill: bad arguments");
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, st, y, n, v);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_copy2d(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t cols, hipStream_t st) {
  ARDAE_CHECK_ARG(x && y && rows > 0 && cols > 0 && ldx >= cols && ldy >= cols, "copy2d: bad arguments");
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, st, x, ldx, y, ldy, rows, cols);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_copy(const float* x, int64_t n, float* y, hipStream_t st) { return launch_copy2d(x, n, y, n, 1, n, st); }

int launch_affine(const float* x, int64_t n, float alpha, float beta, float* y, hipStream_t st) {
  ARDAE_CHECK_ARG(x && y && n > 0, "affine: bad arguments");
  hipLaunchKernelGGL(affine_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, n, alpha, beta, y);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_vae_loss(int kind, const float* o0, const float* o1, const float* x, const float* z, int rows, int nz, int D, int zd,
                    float beta, int write_grads, float gscale, const float* dz_extra, float* rec_row, float* pri_row, float* do0,
                    float* do1, float* dzq, hipStream_t st) {
  ARDAE_CHECK_ARG(o0 && x && z && rec_row && pri_row && rows > 0 && nz > 0 && D > 0 && zd > 0, "vae_loss: bad arguments");
  ARDAE_CHECK_ARG(kind == 0 || (kind == 1 && o1), "vae_loss: kind 1 needs the logvar head");
  ARDAE_CHECK_ARG(!write_grads || (do0 && dzq && (kind == 0 || do1)), "vae_loss: gradient outputs missing");
  if (kind == 0)
    hipLaunchKernelGGL(vae_loss_kernel<0>, dim3(rows), dim3(256), 0, st, o0, o1, x, z, nz, D, zd, beta, write_grads, gscale, dz_extra,
                       rec_row, pri_row, do0, do1, dzq);
  else
    hipLaunchKernelGGL(vae_loss_kernel<1>, dim3(rows), dim3(256), 0, st, o0, o1, x, z, nz, D, zd, beta, write_grads, gscale, dz_extra,
                       rec_row, pri_row, do0, do1, dzq);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_vae_loss_finalize(const float* rec_row, const float* pri_row, int rows, float beta, float* losses, hipStream_t st) {
  ARDAE_CHECK_ARG(rec_row && pri_row && losses && rows > 0, "vae_loss_finalize: bad arguments");
  hipLaunchKernelGGL(vae_loss_finalize_kernel, dim3(1), dim3(256), 0, st, rec_row, pri_row, rows, beta, losses);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ardae
