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
#include <stdlib.h>

#include "linear.h"
#include "linear_epilogue.h"
#include "profile.h"

namespace ardae {

namespace {

// wide geometry: K panel staged in LDS per pass and the workgroups/CU it is compiled for
#ifndef ARDAE_WIDE_KPANEL
#define ARDAE_WIDE_KPANEL 64
#endif
constexpr int WIDE_KPANEL = ARDAE_WIDE_KPANEL;
#ifndef ARDAE_WIDE_MINB
#define ARDAE_WIDE_MINB 3
#endif
constexpr int WIDE_MINB = ARDAE_WIDE_MINB;
// small-M geometry (per-image layers, a few hundred rows): these launches are pure latency - stage the whole K (up to
// 256) in one panel, so a 256-wide layer pays ONE exposed HBM/L2 round trip instead of four (22 -> ~10 us per launch)
#ifndef ARDAE_SMALLM_KPANEL
#define ARDAE_SMALLM_KPANEL 256
#endif
constexpr int SMALLM_KPANEL = ARDAE_SMALLM_KPANEL;
constexpr int SMALLM_MINB = SMALLM_KPANEL > 64 ? 2 : 4;

template <int TM, int TN, int WM, int WN, int KPANEL>
struct Geo {
  static constexpr int BM = TM * WM * 32;
  static constexpr int BN = TN * WN * 32;
  static constexpr int LDW = KPANEL + 4;
  static constexpr int LDS_FLOATS = 2 * BM * LDW;   // double-buffered K panels
};

// One output tile (bx = row tile, by = column panel of a grid that has nby column panels).  `lds` holds
// Geo::LDS_FLOATS floats, `red` four.  linear_kernel runs one tile per workgroup (
// consecutive layers for a fixed set of rows.
template <int TM, int TN, int WM, int WN, int KPANEL, int EPI, int ACT>
__device__ __forceinline__ void linear_tile(const LinArgs& a, int bx, int by, int nby, float* lds, float* red) {
  using G = Geo<TM, TN, WM, WN, KPANEL>;
  constexpr int BM = G::BM, LDW = G::LDW;
  static_assert(WM * WN == 4, "4 waves per workgroup");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, hh = lane >> 5;
  const int row0 = bx * BM;
  const int nblk_total = (a.Nout + 31) >> 5;
  const int nb0 = by * (G::BN / 32) + wn * TN;
  const bool wave_active = nb0 < nblk_total;
  const bool rows_full = row0 + BM <= a.M;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- K loop: (source, K panel) items, LDS double-buffered.  While the MFMAs of panel t run, the global loads of
  //      panel t+1 are in flight into registers; they are written to the other LDS buffer after the MFMA loop, so
  //      there is ONE barrier per panel and HBM latency hides behind 128 MFMAs per wave.
  constexpr int MAXP = (BM * KPANEL / 4 + 255) / 256;   // float4 per thread per full panel
  struct Panel {
    const float* x; const float* wp;
    int ld, K, k0, kw8, c4n, kchunks;
    bool fast;
  };
  auto make_panel = [&](int s, int k0) {
    Panel p;
    p.x = a.src[s].x; p.wp = a.src[s].wp; p.ld = a.src[s].ld; p.K = a.src[s].K; p.k0 = k0;
    const int kw = min(KPANEL, p.K - k0);
    p.kw8 = (kw + 7) & ~7;
    p.c4n = p.kw8 >> 2;
    p.kchunks = (p.K + 7) >> 3;
    const bool vec = ((p.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
    // fast path: whole float4s and a power-of-two row length (shift/mask indexing).  A ragged last panel (K = 100: 64 + 36)
    // is widened to the next power of two with the float4s beyond K zero-filled, as long as K is a multiple of 4.
    if (p.c4n & (p.c4n - 1)) p.c4n = 1 << (32 - __clz(p.c4n));
    p.fast = vec && rows_full && (kw & 3) == 0 && p.c4n * 4 <= KPANEL && (256 / p.c4n) <= BM;
    return p;
  };
  f32x4 pv[MAXP];
  // fast path: whole float4s, power-of-two row length -> shift/mask indexing
  auto panel_load = [&](const Panel& p) {
    const int sh = 31 - __clz(p.c4n);
    const int rpp = 256 >> sh, passes = BM / rpp;
    const int c = (tid & (p.c4n - 1)) << 2;
    const bool inside = p.k0 + c + 4 <= p.K;
    const float* src = p.x + (size_t)(row0 + (tid >> sh)) * p.ld + p.k0 + (inside ? c : 0);
    const size_t gstep = (size_t)rpp * p.ld;
#pragma unroll
    for (int u = 0; u < MAXP; ++u)
      if (u < passes) {
        pv[u] = *reinterpret_cast<const f32x4*>(src + (size_t)u * gstep);
        if (!inside) pv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };
  auto panel_store = [&](const Panel& p, float* buf) {
    const int sh = 31 - __clz(p.c4n);
    const int rpp = 256 >> sh, passes = BM / rpp;
    float* dst = buf + (tid >> sh) * LDW + ((tid & (p.c4n - 1)) << 2);
#pragma unroll
    for (int u = 0; u < MAXP; ++u)
      if (u < passes) *reinterpret_cast<f32x4*>(dst + u * rpp * LDW) = pv[u];
  };
  // generic path: ragged rows / K, unaligned or tiny rows (zero fill outside M x K)
  auto panel_stage_slow = [&](const Panel& p, float* buf) {
    const bool vec = ((p.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
    for (int idx = tid; idx < BM * p.c4n; idx += 256) {
      const int r = idx / p.c4n;
      const int c = (idx - r * p.c4n) << 2;
      const int grow = row0 + r, gcol = p.k0 + c;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (grow < a.M) {
        const float* q = p.x + (size_t)grow * p.ld + gcol;
        if (vec && gcol + 4 <= p.K) {
          v = *reinterpret_cast<const f32x4*>(q);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (gcol + j < p.K) v[j] = q[j];
        }
      }
      *reinterpret_cast<f32x4*>(&buf[r * LDW + c]) = v;
    }
  };
  auto next_item = [&](int& s, int& k0) {   // -> false when the sequence is exhausted
    k0 += KPANEL;
    if (k0 >= a.src[s].K) { ++s; k0 = 0; }
    return s < a.nsrc;
  };

#ifdef ARDAE_STAMPS
  const unsigned long long T0 = __builtin_amdgcn_s_memtime();
#endif
  int cs = 0, ck0 = 0;
  Panel cur = make_panel(0, 0);
  // Small-M geometry with 256-wide panels (per-image layers): the launch is latency, not throughput - put ALL 32 weight
  // fragments of a full panel in flight at once (128 registers), for the first panel even before the activations are
  // staged, instead of one L2 round trip per 8-deep chunk.
  constexpr bool ALLB = TM == 1 && TN == 1 && KPANEL == 256;
  f32x4 ball[ALLB ? 32 : 1];
  bool ball_valid = false;
  auto load_all_b = [&](const Panel& p) {
    const int nb = min(nb0, nblk_total - 1);
    const float* bp = p.wp + ((size_t)nb * p.kchunks + (p.k0 >> 3)) * 256 + lane * 4;
#pragma unroll
    for (int c = 0; c < (ALLB ? 32 : 1); ++c) ball[c] = *reinterpret_cast<const f32x4*>(bp + (size_t)c * 256);
  };
  if (ALLB && wave_active && (cur.kw8 >> 3) == 32) {
    load_all_b(cur);
    ball_valid = true;
    __builtin_amdgcn_sched_barrier(0);
  }
  if (cur.fast) { panel_load(cur); panel_store(cur, lds); } else panel_stage_slow(cur, lds);
  __syncthreads();
#ifdef ARDAE_STAMPS
  const unsigned long long T1 = __builtin_amdgcn_s_memtime();
#endif
  int bufsel = 0;
  while (true) {
    int ns = cs, nk0 = ck0;
    const bool has_next = next_item(ns, nk0);
    Panel nxt = cur;
    if (has_next) {
      nxt = make_panel(ns, nk0);
#ifndef ARDAE_DBG_NOSTAGE
      if (nxt.fast) panel_load(nxt);          // in flight during the MFMA loop
#endif
    }
    if (wave_active) {
      const float* lbuf = lds + bufsel * (BM * LDW);
      const int nch = cur.kw8 >> 3, kc0 = cur.k0 >> 3;
      const float* arow[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) arow[i] = &lbuf[((wm * TM + i) * 32 + l31) * LDW + hh * 4];
      const float* bptr[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nb = min(nb0 + j, nblk_total - 1);   // clamp: out-of-range blocks are masked at the store
#ifdef ARDAE_DBG_BL1
        bptr[j] = cur.wp + (size_t)(nb & 1) * 256 * 8 + lane * 4;
#else
        bptr[j] = cur.wp + ((size_t)nb * cur.kchunks + kc0) * 256 + lane * 4;
#endif
      }
      // weight fragments: two register sets, the other one is always in flight (L2 latency behind 16 MFMAs)
      f32x4 be[TN], bo[TN], av[TM];
      int kc = 0;
      if (ALLB && nch == 32) {
        if (!ball_valid) {
          load_all_b(cur);
          __builtin_amdgcn_sched_barrier(0);
        }
        ball_valid = false;
        f32x4 a2[2];
        a2[0] = *reinterpret_cast<const f32x4*>(arow[0]);
#pragma unroll
        for (int c = 0; c < (ALLB ? 32 : 1); ++c) {
          if (c + 1 < 32) a2[(c + 1) & 1] = *reinterpret_cast<const f32x4*>(arow[0] + (c + 1) * 8);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[c & 1][q], ball[c][q], acc[0][0], 0, 0, 0);
        }
        kc = nch;   // the generic loops below find nothing left
      }
      if (kc < nch) {
#pragma unroll
        for (int j = 0; j < TN; ++j) be[j] = *reinterpret_cast<const f32x4*>(bptr[j]);
      }
      for (; kc + 1 < nch; kc += 2) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bo[j] = *reinterpret_cast<const f32x4*>(bptr[j] + (size_t)(kc + 1) * 256);
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const f32x4*>(arow[i] + kc * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], be[j][q], acc[i][j], 0, 0, 0);
        const int kn = min(kc + 2, nch - 1);
#pragma unroll
        for (int j = 0; j < TN; ++j) be[j] = *reinterpret_cast<const f32x4*>(bptr[j] + (size_t)kn * 256);
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const f32x4*>(arow[i] + (kc + 1) * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], bo[j][q], acc[i][j], 0, 0, 0);
      }
      if (kc < nch) {   // odd tail (be holds chunk kc)
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const f32x4*>(arow[i] + kc * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], be[j][q], acc[i][j], 0, 0, 0);
      }
    }
    if (!has_next) break;
    float* obuf = lds + (bufsel ^ 1) * (BM * LDW);
#ifndef ARDAE_DBG_NOSTAGE
    if (nxt.fast) panel_store(nxt, obuf); else panel_stage_slow(nxt, obuf);
#endif
#ifndef ARDAE_DBG_NOBAR
    __syncthreads();
#endif
    bufsel ^= 1;
    cur = nxt; cs = ns; ck0 = nk0;
  }

#ifdef ARDAE_STAMPS
  const unsigned long long T2 = __builtin_amdgcn_s_memtime();
#endif
  // ------------------------------------------------------------------ epilogue
  float loss_part = 0.f;
  if (wave_active) {
    const bool full = rows_full && (nb0 + TN) * 32 <= a.Nout;   // wave-uniform: no masks, no clamps
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col_raw = (nb0 + j) * 32 + l31;
      const bool cok = col_raw < a.Nout;
      const int col = min(col_raw, a.Nout - 1);
      const float bcol = ((EPI == EPI_ACT || EPI == EPI_DAE_LOSS) && a.bias) ? a.bias[col] : 0.f;
      const float wsig = (EPI == EPI_ACT && a.rowscale_w) ? a.rowscale_w[col] : 0.f;
      float csum = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int rbase = row0 + (wm * TM + i) * 32 + 4 * hh;
        if (full) epilogue_block<EPI, ACT, true>(a, acc[i][j], rbase, col, true, bcol, wsig, csum, loss_part);
        else epilogue_block<EPI, ACT, false>(a, acc[i][j], rbase, col, cok, bcol, wsig, csum, loss_part);
      }
      if (a.colsum != nullptr && WM == 1) {
        const float c2 = csum + __shfl_xor(csum, 32);
        if (hh == 0 && cok) a.colsum[(size_t)bx * a.Nout + col] = c2;
      }
    }
  }
#ifdef ARDAE_STAMPS
  if (a.tile_loss != nullptr && EPI != EPI_DAE_LOSS && lane == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long T3 = __builtin_amdgcn_s_memtime();
    unsigned long long* o = reinterpret_cast<unsigned long long*>(a.tile_loss) + ((size_t)bx * 4 + wave) * 4;
    o[0] = T0; o[1] = T1; o[2] = T2; o[3] = T3;
  }
#endif
  if (EPI == EPI_DAE_LOSS && a.tile_loss != nullptr) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) loss_part += __shfl_xor(loss_part, off);
    if (lane == 0) red[wave] = loss_part;
    __syncthreads();
    if (tid == 0) a.tile_loss[bx * nby + by] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

template <int TM, int TN, int WM, int WN, int KPANEL, int EPI, int ACT, int MINB>
__global__ __launch_bounds__(256, MINB) void linear_kernel(const LinArgs a) {
  using G = Geo<TM, TN, WM, WN, KPANEL>;
#ifdef ARDAE_DBG_LDSPAD
  __shared__ float lds[G::LDS_FLOATS + ARDAE_DBG_LDSPAD];
#else
  __shared__ float lds[G::LDS_FLOATS];
#endif
  __shared__ float red[4];
  linear_tile<TM, TN, WM, WN, KPANEL, EPI, ACT>(a, blockIdx.x, blockIdx.y, gridDim.y, lds, red);
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

struct PackBatchDev {
  int n;
  PackItem it[PACK_BATCH_MAX];
};

// all matrices of a network in ONE launch (blockIdx.y = matrix); same image as pack_weight_kernel
__global__ void pack_batch_kernel(const PackBatchDev b) {
  const PackItem& P = b.it[blockIdx.y];
  const int kchunks = (P.k + 7) >> 3;
  const size_t total4 = (size_t)((P.nout + 31) >> 5) * kchunks * 64;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total4; t += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(t & 63);
    const size_t blk = t >> 6;
    const int kc = (int)(blk % kchunks);
    const int nb = (int)(blk / kchunks);
    const int n = nb * 32 + (lane & 31);
    const int kb = kc * 8 + 4 * (lane >> 5);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < P.nout) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kb + j;
        if (k < P.k) v[j] = P.transpose ? P.W[(size_t)k * P.ldw + n] : P.W[(size_t)n * P.ldw + k];
      }
    }
    reinterpret_cast<f32x4*>(P.out)[t] = v;
  }
}

template <int TM, int TN, int WM, int WN, int KPANEL, int EPI, int ACT, int MINB>
int launch_geo(const LinArgs& a, hipStream_t st) {
  using G = Geo<TM, TN, WM, WN, KPANEL>;
  dim3 grid(ceil_div(a.M, G::BM), ceil_div(a.Nout, G::BN));
  if (g_prof_enabled) {
    char name[96];
    snprintf(name, sizeof(name), "linear_kernel<%d, %d, %d, %d, %d, %d, %d, %d>", TM, TN, WM, WN, KPANEL, EPI, ACT, MINB);
    double ksum = 0;
    for (int s = 0; s < a.nsrc; ++s) ksum += a.src[s].K;
    // algorithmic bytes: X read once, Y (+Y2) written once, S/R/Q read once, weights once
    double tensors = 1.0 + (a.Y2 ? 1 : 0) + ((EPI == EPI_DACT || EPI == EPI_CHAIN) ? 1 : 0) + ((EPI == EPI_CHAIN) ? 1 : 0) +
                     ((EPI == EPI_DACT && a.Q) ? 1 : 0);
    prof_begin(st, name, 2.0 * a.M * (double)a.Nout * ksum, 4.0 * ((double)a.M * ksum + tensors * a.M * (double)a.Nout + ksum * a.Nout));
  }
  hipLaunchKernelGGL((linear_kernel<TM, TN, WM, WN, KPANEL, EPI, ACT, MINB>), grid, dim3(256), 0, st, a);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

// 0 = narrow (Nout <= 32: 128 rows x 32 cols per workgroup), 1 = small-M (32 x 128: per-image B-row problems, where the
// wide tiling would occupy a handful of CUs and serialise 512 MFMAs per wave), 2 = wide (64 x 256)
int pick_geometry(int M, int nout) {
  if (nout <= 32) return 0;
  if ((int64_t)ceil_div(M, 64) * ceil_div(nout, 256) < 128) return 1;
  return 2;
}

template <int EPI, int ACT>
int launch_epi(const LinArgs& a, hipStream_t st) {
  switch (pick_geometry(a.M, a.Nout)) {
    case 0:
      ARDAE_CHECK_ARG(a.colsum == nullptr, "linear: colsum is not available in the narrow (Nout<=32) geometry");
      return launch_geo<1, 1, 4, 1, 64, EPI, ACT, 2>(a, st);
    case 1:
      return launch_geo<1, 1, 1, 4, SMALLM_KPANEL, EPI, ACT, SMALLM_MINB>(a, st);
    default:
      return launch_geo<2, 2, 1, 4, WIDE_KPANEL, EPI, ACT, WIDE_MINB>(a, st);
  }
}

}  // namespace

size_t packed_floats(int nout, int k) { return (size_t)ceil_div(nout, 32) * ceil_div(k, 8) * 256; }
int linear_row_tile(int M, int nout) {
  const int g = pick_geometry(M, nout);
  return g == 0 ? 128 : g == 1 ? 32 : 64;
}
int linear_row_tiles(int M, int nout) { return ceil_div(M, linear_row_tile(M, nout)); }
int linear_col_panels(int M, int nout) {
  const int g = pick_geometry(M, nout);
  return g == 0 ? 1 : ceil_div(nout, g == 1 ? 128 : 256);
}

int launch_pack_weight(const float* W, int ldw, int nout, int k, bool transpose, float* out, hipStream_t st) {
  ARDAE_CHECK_ARG(W && out && nout > 0 && k > 0 && ldw > 0, "pack_weight: bad arguments");
  const int kchunks = ceil_div(k, 8);
  const size_t total4 = packed_floats(nout, k) / 4;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)ceil_div64(total4, 256)), dim3(256), 0, st, W, ldw, nout, k,
                     transpose ? 1 : 0, out, kchunks, total4);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_pack_batch(const PackItem* items, int n, hipStream_t st) {
  for (int start = 0; start < n; start += PACK_BATCH_MAX) {
    PackBatchDev b;
    b.n = n - start < PACK_BATCH_MAX ? n - start : PACK_BATCH_MAX;
    size_t max4 = 0;
    for (int i = 0; i < b.n; ++i) {
      const PackItem& p = items[start + i];
      ARDAE_CHECK_ARG(p.W && p.out && p.nout > 0 && p.k > 0 && p.ldw > 0, "pack_batch: bad item %d", start + i);
      b.it[i] = p;
      const size_t t4 = packed_floats(p.nout, p.k) / 4;
      if (t4 > max4) max4 = t4;
    }
    const int gx = (int)((max4 + 255) / 256);
    hipLaunchKernelGGL(pack_batch_kernel, dim3(gx < 256 ? gx : 256, b.n), dim3(256), 0, st, b);
    ARDAE_LAUNCH_CHECK();
  }
  return 0;
}

// argument validation of launch_linear
int validate_linear(const LinArgs& a, int epi) {
  ARDAE_CHECK_ARG(a.M > 0 && a.Nout > 0, "linear: empty problem (M=%d, Nout=%d)", a.M, a.Nout);
  ARDAE_CHECK_ARG(a.nsrc >= 1 && a.nsrc <= 2, "linear: nsrc must be 1 or 2");
  for (int s = 0; s < a.nsrc; ++s)
    ARDAE_CHECK_ARG(a.src[s].x && a.src[s].wp && a.src[s].K > 0 && a.src[s].ld >= a.src[s].K,
                    "linear: bad source %d (K=%d ld=%d)", s, a.src[s].K, a.src[s].ld);
  ARDAE_CHECK_ARG(a.Y || epi == EPI_DAE_LOSS, "linear: Y is null");
  switch (epi) {
    case EPI_ACT:
      ARDAE_CHECK_ARG(!a.rowbias || a.rows_per_group > 0, "linear: rows_per_group must be positive");
      ARDAE_CHECK_ARG(a.act >= ACT_NONE && a.act <= ACT_LAST, "linear: unknown activation %d", a.act);
      break;
    case EPI_DACT:
      ARDAE_CHECK_ARG(a.S, "linear: EPI_DACT needs S");
      ARDAE_CHECK_ARG(a.act >= ACT_NONE && a.act <= ACT_LAST, "linear: unknown activation %d", a.act);
      break;
    case EPI_CHAIN:
      ARDAE_CHECK_ARG(a.S && a.R && a.Y2, "linear: EPI_CHAIN needs S, R and Y2");
      ARDAE_CHECK_ARG(a.act > ACT_NONE && a.act <= ACT_LAST, "linear: unsupported epilogue/activation combination (epi=%d act=%d)", epi, a.act);
      break;
    case EPI_DAE_LOSS:
      ARDAE_CHECK_ARG(a.sigma && a.eps, "linear: EPI_DAE_LOSS needs sigma and eps");
      break;
    default:
      ARDAE_CHECK_ARG(false, "linear: unknown epilogue %d", epi);
  }
  return 0;
}

int launch_linear(const LinArgs& a, int epi, hipStream_t st) {
  ARDAE_CHECK_ARG(a.M > 0 && a.Nout > 0, "linear: empty problem (M=%d, Nout=%d)", a.M, a.Nout);
  ARDAE_CHECK_ARG(a.nsrc >= 1 && a.nsrc <= 2, "linear: nsrc must be 1 or 2");
  for (int s = 0; s < a.nsrc; ++s)
    ARDAE_CHECK_ARG(a.src[s].x && a.src[s].wp && a.src[s].K > 0 && a.src[s].ld >= a.src[s].K,
                    "linear: bad source %d (K=%d ld=%d)", s, a.src[s].K, a.src[s].ld);
  ARDAE_CHECK_ARG(a.Y || epi == EPI_DAE_LOSS, "linear: Y is null");
  if (linear_small_eligible(a, epi)) {
    ARDAE_TRY(validate_linear(a, epi));
    return launch_linear_small(a, epi, st);
  }
  if (linear_narrow_eligible(a, epi)) {
    ARDAE_TRY(validate_linear(a, epi));
    return launch_linear_narrow(a, epi, st);
  }
  if (linear_shortk_eligible(a, epi)) {
    ARDAE_TRY(validate_linear(a, epi));
    return launch_linear_shortk(a, epi, st);
  }
  static const bool wide_on = !(debug_knob("ARDAE_WIDE") && atoi(debug_knob("ARDAE_WIDE")) == 0);
  if (wide_on && linear_wide_eligible(a, epi)) {
    if (epi == EPI_ACT) ARDAE_CHECK_ARG(!a.rowbias || a.rows_per_group > 0, "linear: rows_per_group must be positive");
    if (epi == EPI_DACT || epi == EPI_CHAIN) ARDAE_CHECK_ARG(a.S, "linear: EPI_DACT/EPI_CHAIN need S");
    if (epi == EPI_CHAIN) ARDAE_CHECK_ARG(a.R && a.Y2, "linear: EPI_CHAIN needs S, R and Y2");
    return launch_linear_wide(a, epi, st);
  }
  switch (epi) {
    case EPI_ACT:
      ARDAE_CHECK_ARG(!a.rowbias || a.rows_per_group > 0, "linear: rows_per_group must be positive");
      if (a.act == ACT_NONE) return launch_epi<EPI_ACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_epi<EPI_ACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_epi<EPI_ACT, ACT_SOFTPLUS>(a, st);
      if (a.act == ACT_ELU) return launch_epi<EPI_ACT, ACT_ELU>(a, st);
      if (a.act == ACT_TANH) return launch_epi<EPI_ACT, ACT_TANH>(a, st);
      if (a.act == ACT_LEAKY) return launch_epi<EPI_ACT, ACT_LEAKY>(a, st);
      if (a.act == ACT_SWISH) return launch_epi<EPI_ACT, ACT_SWISH>(a, st);
      break;
    case EPI_DACT:
      ARDAE_CHECK_ARG(a.S, "linear: EPI_DACT needs S");
      if (a.act == ACT_NONE) return launch_epi<EPI_DACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_epi<EPI_DACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_epi<EPI_DACT, ACT_SOFTPLUS>(a, st);
      if (a.act == ACT_ELU) return launch_epi<EPI_DACT, ACT_ELU>(a, st);
      if (a.act == ACT_TANH) return launch_epi<EPI_DACT, ACT_TANH>(a, st);
      if (a.act == ACT_LEAKY) return launch_epi<EPI_DACT, ACT_LEAKY>(a, st);
      if (a.act == ACT_SWISH) return launch_epi<EPI_DACT, ACT_SWISH>(a, st);
      break;
    case EPI_CHAIN:
      ARDAE_CHECK_ARG(a.S && a.R && a.Y2, "linear: EPI_CHAIN needs S, R and Y2");
      if (a.act == ACT_SOFTPLUS) return launch_epi<EPI_CHAIN, ACT_SOFTPLUS>(a, st);
      if (a.act == ACT_RELU) return launch_epi<EPI_CHAIN, ACT_RELU>(a, st);
      if (a.act == ACT_ELU) return launch_epi<EPI_CHAIN, ACT_ELU>(a, st);
      if (a.act == ACT_TANH) return launch_epi<EPI_CHAIN, ACT_TANH>(a, st);
      if (a.act == ACT_LEAKY) return launch_epi<EPI_CHAIN, ACT_LEAKY>(a, st);
      if (a.act == ACT_SWISH) return launch_epi<EPI_CHAIN, ACT_SWISH>(a, st);
      break;
    case EPI_DAE_LOSS:
      ARDAE_CHECK_ARG(a.sigma && a.eps, "linear: EPI_DAE_LOSS needs sigma and eps");
      return launch_epi<EPI_DAE_LOSS, ACT_NONE>(a, st);
  }
  ARDAE_CHECK_ARG(false, "linear: unsupported epilogue/activation combination (epi=%d act=%d)", epi, a.act);
  return -1;
}

}  // namespace ardae
