// FP32-MFMA fused linear kernels for gfx950 (MI355X).  See linear.h for the operator definition.
//
// Design (DESIGN.md "K1"):
//  * one workgroup = 4 waves (one per SIMD), two workgroups per CU;
//  * the activation tile X[BM rows][K panel] is staged once in LDS (row stride KPANEL+4 floats, which makes
//    the ds_read_b128 fragment reads bank-conflict free: 16 consecutive rows land on 16 distinct 16-B slots);
//  * weights are consumed straight from L2 in a pre-packed, lane-linear image (one coalesced 1-KiB
//    global_load_dwordx4 per 32x8 block) - no LDS traffic and no transposes in the hot loop; the
//    whole network's weights (~2 MiB at h=256) stay resident in each XCD's 4-MiB L2;
//  * v_mfma_f32_32x32x2_f32: lane (n=l&31, hh=l>>5) holds 4 consecutive k of A/B, so MFMA j of a
//    chunk contracts k = 8*kc + 4*hh + j for both operands (any k->MFMA assignment is legal as long as
//    A and B agree);
//  * the epilogue works on the accumulator layout directly (col = l&31, row = (r&3)+8*(r>>2)+4*hh):
//    every dword load/store instruction touches two full 128-B lines.
#include "linear.h"

namespace ardae {

namespace {

template <int TM, int TN, int WM, int WN, int KPANEL>
struct Geo {
  static constexpr int BM = TM * WM * 32;
  static constexpr int BN = TN * WN * 32;
  static constexpr int LDW = KPANEL + 4;
  static constexpr int LDS_FLOATS = BM * LDW;
};

template <int TM, int TN, int WM, int WN, int KPANEL, int EPI, int ACT>
__global__ __launch_bounds__(256, 2) void linear_kernel(const LinArgs a) {
  using G = Geo<TM, TN, WM, WN, KPANEL>;
  constexpr int BM = G::BM, LDW = G::LDW;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ float lds[G::LDS_FLOATS];
  __shared__ float red[4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hh = lane >> 5;
  const int row0 = blockIdx.x * BM;
  const int nblk_total = (a.Nout + 31) >> 5;
  const int nb0 = blockIdx.y * (G::BN / 32) + wn * TN;
  const bool wave_active = nb0 < nblk_total;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int s = 0; s < a.nsrc; ++s) {
    const float* __restrict__ x = a.src[s].x;
    const float* __restrict__ wp = a.src[s].wp;
    const int ld = a.src[s].ld, K = a.src[s].K;
    const int kchunks = (K + 7) >> 3;
    const bool vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    for (int k0 = 0; k0 < K; k0 += KPANEL) {
      const int kw = min(KPANEL, K - k0);
      const int kw8 = (kw + 7) & ~7;
      const int c4n = kw8 >> 2;
      __syncthreads();
      // ---- stage X[row0:row0+BM, k0:k0+kw8] into LDS (zero fill outside M x K) ----
      for (int idx = tid; idx < BM * c4n; idx += 256) {
        const int r = idx / c4n;
        const int c = (idx - r * c4n) << 2;
        const int grow = row0 + r, gcol = k0 + c;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (grow < a.M) {
          const float* p = x + (size_t)grow * ld + gcol;
          if (vec && gcol + 4 <= K) {
            v = *reinterpret_cast<const f32x4*>(p);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (gcol + j < K) v[j] = p[j];
          }
        }
        *reinterpret_cast<f32x4*>(&lds[r * LDW + c]) = v;
      }
      __syncthreads();
      // ---- MFMA over this panel ----
      if (wave_active) {
        const int nch = kw8 >> 3, kc0 = k0 >> 3;
        const float* arow[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) arow[i] = &lds[((wm * TM + i) * 32 + l31) * LDW + hh * 4];
        const float* bptr[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int nb = min(nb0 + j, nblk_total - 1);   // clamp: out-of-range blocks are masked at the store
          bptr[j] = wp + ((size_t)nb * kchunks + kc0) * 256 + lane * 4;
        }
#pragma unroll 4
        for (int kc = 0; kc < nch; ++kc) {
          f32x4 av[TM], bv[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bptr[j] + (size_t)kc * 256);
#pragma unroll
          for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const f32x4*>(arow[i] + kc * 8);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], bv[j][q], acc[i][j], 0, 0, 0);
        }
      }
    }
  }

  // ------------------------------------------------------------------ epilogue
  float loss_part = 0.f;
  if (wave_active) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = (nb0 + j) * 32 + l31;
      const bool cok = (nb0 + j) < nblk_total && col < a.Nout;
      float csum = 0.f;
      float bcol = 0.f, wsig = 0.f;
      if (EPI == EPI_ACT || EPI == EPI_DAE_LOSS) {
        if (cok && a.bias) bcol = a.bias[col];
        if (cok && a.rowscale_w) wsig = a.rowscale_w[col];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (!(cok && row < a.M)) continue;
          float v = acc[i][j][r];
          float y;
          if (EPI == EPI_ACT) {
            v += bcol;
            if (a.rowbias) v += a.rowbias[(size_t)(row / a.rows_per_group) * a.rowbias_ld + col];
            if (a.rowscale) v += a.rowscale[row] * wsig;
            y = act_fwd<ACT>(v);
            a.Y[(size_t)row * a.ldY + col] = y;
            // optional seed of the score pass: e_L = -w (.) act'(pre)   (R is the [Nout] vector w here)
            if (a.Y2) a.Y2[(size_t)row * a.ldY2 + col] = -a.R[col] * act_d1<ACT>(y);
          } else if (EPI == EPI_DACT) {
            const float sd = act_d1<ACT>(a.S[(size_t)row * a.ldS + col]);
            y = v * sd;
            if (a.Q) y += a.Q[(size_t)row * a.ldQ + col];
            a.Y[(size_t)row * a.ldY + col] = y;
          } else if (EPI == EPI_CHAIN) {
            const float sd = act_d1<ACT>(a.S[(size_t)row * a.ldS + col]);
            y = v * sd;
            a.Y[(size_t)row * a.ldY + col] = y;
            float pb = 0.f;
            if (ACT == ACT_SOFTPLUS) pb = v * a.R[(size_t)row * a.ldR + col] * (1.f - sd);
            a.Y2[(size_t)row * a.ldY2 + col] = pb;
          } else {  // EPI_DAE_LOSS
            const float g = v + bcol;
            const float sg = a.sigma[row];
            const float rho = sg * g + a.eps[(size_t)row * a.ldeps + col];
            if (a.Y) a.Y[(size_t)row * a.ldY + col] = g;
            if (a.Y2) a.Y2[(size_t)row * a.ldY2 + col] = 2.f * sg * rho * a.scale;
            loss_part += rho * rho;
            y = g;
          }
          csum += y;
        }
      }
      if (a.colsum != nullptr && WM == 1) {
        csum += __shfl_xor(csum, 32);
        if (hh == 0 && cok) a.colsum[(size_t)blockIdx.x * a.Nout + col] = csum;
      }
    }
  }
  if (EPI == EPI_DAE_LOSS && a.tile_loss != nullptr) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) loss_part += __shfl_xor(loss_part, off);
    if (lane == 0) red[wave] = loss_part;
    __syncthreads();
    if (tid == 0) a.tile_loss[blockIdx.x * gridDim.y + blockIdx.y] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// M[n][k] -> packed[nb][kc][lane][j], n = nb*32 + (lane&31), k = kc*8 + 4*(lane>>5) + j
__global__ void pack_weight_kernel(const float* __restrict__ W, int ldw, int nout, int K, int transpose,
                                   float* __restrict__ out, int kchunks, size_t total4) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total4) return;
  const int lane = (int)(t & 63);
  const size_t blk = t >> 6;
  const int kc = (int)(blk % kchunks);
  const int nb = (int)(blk / kchunks);
  const int n = nb * 32 + (lane & 31);
  const int kb = kc * 8 + 4 * (lane >> 5);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (n < nout) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kb + j;
      if (k < K) v[j] = transpose ? W[(size_t)k * ldw + n] : W[(size_t)n * ldw + k];
    }
  }
  reinterpret_cast<f32x4*>(out)[t] = v;
}

template <int TM, int TN, int WM, int WN, int KPANEL, int EPI, int ACT>
int launch_geo(const LinArgs& a, hipStream_t st) {
  using G = Geo<TM, TN, WM, WN, KPANEL>;
  dim3 grid(ceil_div(a.M, G::BM), ceil_div(a.Nout, G::BN));
  hipLaunchKernelGGL((linear_kernel<TM, TN, WM, WN, KPANEL, EPI, ACT>), grid, dim3(256), 0, st, a);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

template <int EPI, int ACT>
int launch_epi(const LinArgs& a, hipStream_t st) {
  if (a.Nout <= 32) {
    ARDAE_CHECK_ARG(a.colsum == nullptr, "linear: colsum is not available in the narrow (Nout<=32) geometry");
    return launch_geo<1, 1, 4, 1, 128, EPI, ACT>(a, st);
  }
  return launch_geo<2, 2, 1, 4, 256, EPI, ACT>(a, st);
}

}  // namespace

size_t packed_floats(int nout, int k) { return (size_t)ceil_div(nout, 32) * ceil_div(k, 8) * 256; }
int linear_row_tile(int nout) { return nout <= 32 ? 128 : 64; }
int linear_row_tiles(int M, int nout) { return ceil_div(M, linear_row_tile(nout)); }
int linear_col_panels(int nout) { return nout <= 32 ? 1 : ceil_div(nout, 256); }

int launch_pack_weight(const float* W, int ldw, int nout, int k, bool transpose, float* out, hipStream_t st) {
  ARDAE_CHECK_ARG(W && out && nout > 0 && k > 0 && ldw > 0, "pack_weight: bad arguments");
  const int kchunks = ceil_div(k, 8);
  const size_t total4 = packed_floats(nout, k) / 4;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)ceil_div64(total4, 256)), dim3(256), 0, st, W, ldw, nout, k,
                     transpose ? 1 : 0, out, kchunks, total4);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_linear(const LinArgs& a, int epi, hipStream_t st) {
  ARDAE_CHECK_ARG(a.M > 0 && a.Nout > 0, "linear: empty problem (M=%d, Nout=%d)", a.M, a.Nout);
  ARDAE_CHECK_ARG(a.nsrc >= 1 && a.nsrc <= 2, "linear: nsrc must be 1 or 2");
  for (int s = 0; s < a.nsrc; ++s)
    ARDAE_CHECK_ARG(a.src[s].x && a.src[s].wp && a.src[s].K > 0 && a.src[s].ld >= a.src[s].K,
                    "linear: bad source %d (K=%d ld=%d)", s, a.src[s].K, a.src[s].ld);
  ARDAE_CHECK_ARG(a.Y || epi == EPI_DAE_LOSS, "linear: Y is null");
  switch (epi) {
    case EPI_ACT:
      ARDAE_CHECK_ARG(!a.rowbias || a.rows_per_group > 0, "linear: rows_per_group must be positive");
      if (a.act == ACT_NONE) return launch_epi<EPI_ACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_epi<EPI_ACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_epi<EPI_ACT, ACT_SOFTPLUS>(a, st);
      break;
    case EPI_DACT:
      ARDAE_CHECK_ARG(a.S, "linear: EPI_DACT needs S");
      if (a.act == ACT_NONE) return launch_epi<EPI_DACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_epi<EPI_DACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_epi<EPI_DACT, ACT_SOFTPLUS>(a, st);
      break;
    case EPI_CHAIN:
      ARDAE_CHECK_ARG(a.S && a.R && a.Y2, "linear: EPI_CHAIN needs S, R and Y2");
      if (a.act == ACT_SOFTPLUS) return launch_epi<EPI_CHAIN, ACT_SOFTPLUS>(a, st);
      break;
    case EPI_DAE_LOSS:
      ARDAE_CHECK_ARG(a.sigma && a.eps, "linear: EPI_DAE_LOSS needs sigma and eps");
      return launch_epi<EPI_DAE_LOSS, ACT_NONE>(a, st);
  }
  ARDAE_CHECK_ARG(false, "linear: unsupported epilogue/activation combination (epi=%d act=%d)", epi, a.act);
  return -1;
}

}  // namespace ardae
