// Streaming kernel for the N-row layers that END in the latent space: Y[M, Nout <= 32] = X[M, 256] . Wp^T (+ bias), i.e.
// the sampler's output layer (ivae/mnist.py:151, fc of the 256+100 -> 256 -> z MLP on B*nz rows) and the last product of
// the score pass with the DAE loss fused (g = r_1 A_1, rho = sigma g + eps, gbar, sum rho^2: graddae/mlp.py:437-444).
// 2 GFLOP against 134 MB of activations at config #2: the bound is HBM, and what it takes is bytes in flight.  The generic
// narrow geometry of linear_kernel staged 64-wide K panels through LDS with one panel in flight per workgroup (2.0-2.4
// TB/s); here nothing is staged: a wave owns 32 rows and requests their whole K extent at once, straight into MFMA
// A-fragment registers (32 x global_load_dwordx4 per lane = 32 KB per wave in flight, two waves per SIMD), takes the packed
// weight fragments from L2 eight chunks at a time, and runs its 128 MFMAs as the data lands.  No LDS, no barriers.
#include <stdlib.h>

#include "linear.h"
#include "profile.h"

namespace ardae {
namespace {

constexpr int NK = 32;   // chunks of 8 k: K = 256

template <int EPI>
__global__ __launch_bounds__(256) void linear_narrow_kernel(const LinArgs a) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int row0 = (blockIdx.x * 4 + wave) * 32;

  const float* xr = a.src[0].x + (size_t)(row0 + l31) * a.src[0].ld + 4 * hh;
  f32x4 av[NK];
#pragma unroll
  for (int c = 0; c < NK; ++c) av[c] = *reinterpret_cast<const f32x4*>(xr + 8 * c);

  // epilogue operands: requested now, used last
  const int col = min(l31, a.Nout - 1);
  const bool cok = l31 < a.Nout;
  const float bcol = a.bias ? a.bias[col] : 0.f;
  float sg[16], ev[16];
  if (EPI == EPI_DAE_LOSS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      sg[r] = a.sigma[row];
      ev[r] = a.eps[(size_t)row * a.ldeps + col];
    }
  }

  const float* bp = a.src[0].wp + lane * 4;   // column block 0 of the packed image, kchunks = 32
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 b0[8], b1[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) b0[u] = *reinterpret_cast<const f32x4*>(bp + (size_t)u * 256);
#pragma unroll
  for (int g = 0; g < 4; g += 2) {
#pragma unroll
    for (int u = 0; u < 8; ++u) b1[u] = *reinterpret_cast<const f32x4*>(bp + (size_t)(8 * (g + 1) + u) * 256);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[8 * g + u][q], b0[u][q], acc, 0, 0, 0);
    if (g + 2 < 4) {
#pragma unroll
      for (int u = 0; u < 8; ++u) b0[u] = *reinterpret_cast<const f32x4*>(bp + (size_t)(8 * (g + 2) + u) * 256);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[8 * (g + 1) + u][q], b1[u][q], acc, 0, 0, 0);
  }

  if (EPI == EPI_ACT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (cok) a.Y[(size_t)row * a.ldY + col] = acc[r] + bcol;
    }
  } else {
    float loss_part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float y = acc[r] + bcol;
      const float rho = sg[r] * y + ev[r];
      if (cok) {
        if (a.Y) a.Y[(size_t)row * a.ldY + col] = y;
        if (a.Y2) a.Y2[(size_t)row * a.ldY2 + col] = 2.f * sg[r] * rho * a.scale;
        loss_part += rho * rho;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) loss_part += __shfl_xor(loss_part, off);
    if (lane == 0) red[wave] = loss_part;
    __syncthreads();
    if (tid == 0) a.tile_loss[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

template <int EPI>
int launch_narrow(const LinArgs& a, hipStream_t st) {
  if (g_prof_enabled) {
    char name[64];
    snprintf(name, sizeof(name), "linear_narrow_kernel<%d>", EPI);
    const double tensors = EPI == EPI_DAE_LOSS ? (1.0 + (a.Y ? 1 : 0) + (a.Y2 ? 1 : 0)) : 1.0;
    prof_begin(st, name, 2.0 * a.M * (double)a.Nout * 256.0, 4.0 * ((double)a.M * 256.0 + tensors * a.M * (double)a.Nout + 256.0 * a.Nout));
  }
  hipLaunchKernelGGL((linear_narrow_kernel<EPI>), dim3(a.M / 128), dim3(256), 0, st, a);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// One source of K = 256, Nout <= 32, whole 128-row tiles (the row-tile size the per-tile loss partials are sized for:
// linear_row_tile() of the narrow geometry), vector-aligned rows; plain bias epilogue or the DAE loss.  ARDAE_NARROW=0: off.
bool linear_narrow_eligible(const LinArgs& a, int epi) {
  static const bool on = !(debug_knob("ARDAE_NARROW") && atoi(debug_knob("ARDAE_NARROW")) == 0);
  if (!on || a.nsrc != 1 || a.src[0].K != 256 || a.Nout <= 0 || a.Nout > 32 || a.M <= 0 || (a.M % 128)) return false;
  if ((a.src[0].ld & 3) || (reinterpret_cast<uintptr_t>(a.src[0].x) & 15) || a.colsum != nullptr) return false;
  if (epi == EPI_ACT) return a.act == ACT_NONE && !a.rowbias && !a.rowscale && !a.Y2 && a.Y != nullptr;
  if (epi == EPI_DAE_LOSS) return a.tile_loss != nullptr && a.sigma != nullptr && a.eps != nullptr;
  return false;
}

int launch_linear_narrow(const LinArgs& a, int epi, hipStream_t st) {
  return epi == EPI_ACT ? launch_narrow<EPI_ACT>(a, st) : launch_narrow<EPI_DAE_LOSS>(a, st);
}

}  // namespace ardae
