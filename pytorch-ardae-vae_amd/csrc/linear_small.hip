// Latency-optimised fused linear kernel for the per-image (B-row) problems: the encoder trunk, the context encoder, the
// decoder, the sigma = 0 score pass of the VAE update and their backward chains (models/layers.py:501-515 on a few
// hundred rows).  A step has ~60 of these launches, each a dependent link of a chain, each a few MFLOP: what they cost
// is their own critical path, not throughput.  The 32 x 128 tiling of linear_kernel gives every wave the whole K
// (128 MFMAs = 3.4 us at K = 256) behind an LDS staging round trip; here
//   * one workgroup owns ONE 32 x 32 output block and its four waves split K (32 MFMAs each at K = 256), so a 512 x 256
//     layer runs on 128 CUs instead of 32;
//   * nothing is staged: every wave loads its own A fragments (one float4 per lane per 8-deep chunk, rows are at least
//     32 B contiguous) and packed weight fragments straight into registers, eight chunks of both in flight at once -
//     one memory round trip per 64 k, double buffered for longer K (784-wide first layer, concat inputs);
//   * the epilogue's operands (saved activations, per-image biases) are requested before the K loop;
//   * the four partial accumulators meet in LDS (16 KB) and wave w finishes rows 8w .. 8w+7 of the block, so the epilogue
//     and its stores are spread over the whole workgroup.
// Same operator, same packed-weight image and same epilogue arithmetic (order of the K summation aside) as linear_kernel.
#include <stdlib.h>

#include "linear.h"
#include "profile.h"

namespace ardae {
namespace {

constexpr int SB = 8;   // chunks (of 8 k) per register batch

struct SmallSrc {
  const float* x; const float* wp;
  int ld, K, kchunks;
  bool vec;
};

template <int EPI, int ACT>
__global__ __launch_bounds__(256) void linear_small_kernel(const LinArgs a) {
  __shared__ float red[4][16][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int row0 = blockIdx.x * 32, nb = blockIdx.y;
  const int arow = min(row0 + l31, a.M - 1);

  // ---- epilogue operands first: they are independent of the K loop
  const int col_raw = nb * 32 + l31;
  const bool cok = col_raw < a.Nout;
  const int col = min(col_raw, a.Nout - 1);
  int rowv[4];
  bool ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int raw = row0 + 8 * wave + 4 * hh + i;
    ok[i] = cok && raw < a.M;
    rowv[i] = min(raw, a.M - 1);
  }
  float o1[4], o2[4], bcol = 0.f, wsig = 0.f, wv = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) o1[i] = 0.f, o2[i] = 0.f;
  if (EPI == EPI_ACT) {
    if (a.bias) bcol = a.bias[col];
    if (a.rowbias) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o1[i] = a.rowbias[(size_t)(rowv[i] / a.rows_per_group) * a.rowbias_ld + col];
    }
    if (a.rowscale) {
      wsig = a.rowscale_w[col];
#pragma unroll
      for (int i = 0; i < 4; ++i) o2[i] = a.rowscale[rowv[i]];
    }
    if (a.Y2) wv = a.R[col];
  } else if (EPI == EPI_DACT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o1[i] = a.S[(size_t)rowv[i] * a.ldS + col];
    if (a.Q) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o2[i] = a.Q[(size_t)rowv[i] * a.ldQ + col];
    }
  } else {   // EPI_CHAIN
#pragma unroll
    for (int i = 0; i < 4; ++i) o1[i] = a.S[(size_t)rowv[i] * a.ldS + col];
#pragma unroll
    for (int i = 0; i < 4; ++i) o2[i] = a.R[(size_t)rowv[i] * a.ldR + col];
  }

  // ---- K loop over the flat chunk list of both sources; wave w owns a contiguous quarter
  SmallSrc s0, s1;
  s0.x = a.src[0].x; s0.wp = a.src[0].wp; s0.ld = a.src[0].ld; s0.K = a.src[0].K; s0.kchunks = (s0.K + 7) >> 3;
  s0.vec = ((s0.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(s0.x) & 15) == 0);
  s1 = s0;
  int n1 = 0;
  if (a.nsrc > 1) {
    s1.x = a.src[1].x; s1.wp = a.src[1].wp; s1.ld = a.src[1].ld; s1.K = a.src[1].K; s1.kchunks = (s1.K + 7) >> 3;
    s1.vec = ((s1.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(s1.x) & 15) == 0);
    n1 = s1.kchunks;
  }
  const int n0 = s0.kchunks, T = n0 + n1;
  const int cbeg = (T * wave) >> 2, cend = (T * (wave + 1)) >> 2;

  auto load_chunk = [&](int c, f32x4& av, f32x4& bv) {
    const bool live = c < cend;
    const int cc = live ? c : cend - 1;       // clamped: a dead chunk re-reads a valid address and contributes A = 0
    const bool second = cc >= n0;
    const SmallSrc& s = second ? s1 : s0;
    const int kc = second ? cc - n0 : cc;
    const int k = kc * 8 + 4 * hh;
    bv = *reinterpret_cast<const f32x4*>(s.wp + ((size_t)nb * s.kchunks + kc) * 256 + lane * 4);
    const float* p = s.x + (size_t)arow * s.ld + k;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s.vec && k + 4 <= s.K) {
      v = *reinterpret_cast<const f32x4*>(p);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k + j < s.K) v[j] = p[j];
    }
    if (!live) v = f32x4{0.f, 0.f, 0.f, 0.f};
    av = v;
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f32x4 a0[SB], b0[SB], a1[SB], b1[SB];
  if (cbeg < cend) {
#pragma unroll
    for (int u = 0; u < SB; ++u) load_chunk(cbeg + u, a0[u], b0[u]);
    for (int c = cbeg; c < cend; c += 2 * SB) {
      const bool more1 = c + SB < cend;
      if (more1) {
#pragma unroll
        for (int u = 0; u < SB; ++u) load_chunk(c + SB + u, a1[u], b1[u]);
      }
#pragma unroll
      for (int u = 0; u < SB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][q], b0[u][q], acc, 0, 0, 0);
      if (!more1) break;
      if (c + 2 * SB < cend) {
#pragma unroll
        for (int u = 0; u < SB; ++u) load_chunk(c + 2 * SB + u, a0[u], b0[u]);
      }
#pragma unroll
      for (int u = 0; u < SB; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u][q], b1[u][q], acc, 0, 0, 0);
    }
  }

  // ---- the four partial sums meet in LDS; wave w finishes accumulator registers 4w .. 4w+3
  //      (rows 8w + 4hh + {0..3} of the block, column l31: every store instruction covers two full 128-B lines)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    v[i] = (red[0][4 * wave + i][lane] + red[1][4 * wave + i][lane]) + (red[2][4 * wave + i][lane] + red[3][4 * wave + i][lane]);

  if (EPI == EPI_ACT) {
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = act_fwd<ACT>(v[i] + bcol + o1[i] + o2[i] * wsig);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (ok[i]) a.Y[(size_t)rowv[i] * a.ldY + col] = y[i];
    if (a.Y2) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ok[i]) a.Y2[(size_t)rowv[i] * a.ldY2 + col] = -wv * act_d1<ACT>(y[i]);
    }
  } else if (EPI == EPI_DACT) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (ok[i]) a.Y[(size_t)rowv[i] * a.ldY + col] = v[i] * act_d1<ACT>(o1[i]) + o2[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float em = act_ratio<ACT>(o1[i]);   // s'/s (softplus: 1 - s without cancellation)
      if (ok[i]) {
        a.Y[(size_t)rowv[i] * a.ldY + col] = v[i] * act_d1<ACT>(o1[i]);
        a.Y2[(size_t)rowv[i] * a.ldY2 + col] = v[i] * o2[i] * em;
      }
    }
  }
}

template <int EPI, int ACT>
int launch_small(const LinArgs& a, hipStream_t st) {
  if (g_prof_enabled) {
    char name[64];
    snprintf(name, sizeof(name), "linear_small_kernel<%d, %d>", EPI, ACT);
    double ksum = 0;
    for (int s = 0; s < a.nsrc; ++s) ksum += a.src[s].K;
    const double tensors = 1.0 + (a.Y2 ? 1 : 0) + ((EPI == EPI_DACT || EPI == EPI_CHAIN) ? 1 : 0) + ((EPI == EPI_CHAIN) ? 1 : 0) +
                           ((EPI == EPI_DACT && a.Q) ? 1 : 0);
    prof_begin(st, name, 2.0 * a.M * (double)a.Nout * ksum, 4.0 * ((double)a.M * ksum + tensors * a.M * (double)a.Nout + ksum * a.Nout));
  }
  hipLaunchKernelGGL((linear_small_kernel<EPI, ACT>), dim3(ceil_div(a.M, 32), ceil_div(a.Nout, 32)), dim3(256), 0, st, a);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Per-image problems only: up to SMALL_MAX_TILES workgroups (beyond that the throughput tilings win), no per-tile side
// outputs (column sums, DAE-loss partials: those belong to N-row launches).  ARDAE_SMALL=0 switches the kernel off.
bool linear_small_eligible(const LinArgs& a, int epi) {
  static const bool on = !(debug_knob("ARDAE_SMALL") && atoi(debug_knob("ARDAE_SMALL")) == 0);
  static const int max_tiles = debug_knob("ARDAE_SMALL_MAX_TILES") ? atoi(debug_knob("ARDAE_SMALL_MAX_TILES")) : 512;
  if (!on || epi == EPI_DAE_LOSS || a.colsum != nullptr || a.tile_loss != nullptr) return false;
  if (a.M <= 0 || a.Nout <= 0) return false;
  if (epi == EPI_CHAIN && a.act == ACT_NONE) return false;
  return (int64_t)ceil_div(a.M, 32) * ceil_div(a.Nout, 32) <= max_tiles;
}

int launch_linear_small(const LinArgs& a, int epi, hipStream_t st) {
  switch (epi) {
    case EPI_ACT:
      if (a.act == ACT_NONE) return launch_small<EPI_ACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_small<EPI_ACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_small<EPI_ACT, ACT_SOFTPLUS>(a, st);
      if (a.act == ACT_ELU) return launch_small<EPI_ACT, ACT_ELU>(a, st);
      if (a.act == ACT_TANH) return launch_small<EPI_ACT, ACT_TANH>(a, st);
      if (a.act == ACT_LEAKY) return launch_small<EPI_ACT, ACT_LEAKY>(a, st);
      break;
    case EPI_DACT:
      if (a.act == ACT_NONE) return launch_small<EPI_DACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_small<EPI_DACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_small<EPI_DACT, ACT_SOFTPLUS>(a, st);
      if (a.act == ACT_ELU) return launch_small<EPI_DACT, ACT_ELU>(a, st);
      if (a.act == ACT_TANH) return launch_small<EPI_DACT, ACT_TANH>(a, st);
      if (a.act == ACT_LEAKY) return launch_small<EPI_DACT, ACT_LEAKY>(a, st);
      break;
    case EPI_CHAIN:
      if (a.act == ACT_SOFTPLUS) return launch_small<EPI_CHAIN, ACT_SOFTPLUS>(a, st);
      if (a.act == ACT_RELU) return launch_small<EPI_CHAIN, ACT_RELU>(a, st);
      if (a.act == ACT_ELU) return launch_small<EPI_CHAIN, ACT_ELU>(a, st);
      if (a.act == ACT_TANH) return launch_small<EPI_CHAIN, ACT_TANH>(a, st);
      if (a.act == ACT_LEAKY) return launch_small<EPI_CHAIN, ACT_LEAKY>(a, st);
      break;
  }
  ARDAE_CHECK_ARG(false, "linear_small: unsupported epilogue/activation combination (epi=%d act=%d)", epi, a.act);
  return -1;
}

}  // namespace ardae
