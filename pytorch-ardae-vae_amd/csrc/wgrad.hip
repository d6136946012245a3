// Batched FP32-MFMA weight-gradient kernel for gfx950.  See wgrad.h.
//
// Workgroup tile: 128 (o) x 256 (i) outputs, 4 waves as 2(o) x 2(i), each wave 64 x 128 = 2 x 4 MFMA
// 32x32 tiles (128 accumulator registers).  The reduction runs over ROWS m: a chunk of 32 rows of G and X
// is staged in LDS as stored ([m][col], no transpose needed: with v_mfma_f32_32x32x2_f32 lane (c=l&31,
// kk=l>>5) supplies A[o=c][k=kk] = G[m0+kk][o] and B[k=kk][i=c] = X[m0+kk][i], i.e. both fragment reads are
// 32 consecutive floats per half-wave - conflict-free ds_read_b32).
#include "wgrad.h"
#include "linear.h"
#include "profile.h"

namespace ardae {
namespace {

constexpr int BO = 128, BI = 256, RC = 32;

struct WgradBatchDev {
  int nprob;
  int wg_begin[WGRAD_MAX_PROBLEMS + 1];
  int o_tiles[WGRAD_MAX_PROBLEMS];
  int i_tiles[WGRAD_MAX_PROBLEMS];
  WgradProblem p[WGRAD_MAX_PROBLEMS];
};
static_assert(sizeof(WgradBatchDev) <= 4000, "kernel argument block too large");

__device__ __forceinline__ f32x4 load4_guard(const float* base, int ld, int row, int col, int row_end, int ncols,
                                             bool vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (row < row_end && col < ncols) {
    const float* p = base + (size_t)row * ld + col;
    if (vec && col + 4 <= ncols) {
      v = *reinterpret_cast<const f32x4*>(p);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (col + j < ncols) v[j] = p[j];
    }
  }
  return v;
}

__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradBatchDev batch) {
  __shared__ float Gs[RC * BO];
  __shared__ float Xs[RC * BI];

  int pi = 0;
  while (pi + 1 < batch.nprob && (int)blockIdx.x >= batch.wg_begin[pi + 1]) ++pi;
  const WgradProblem& P = batch.p[pi];
  const int local = blockIdx.x - batch.wg_begin[pi];
  const int split = local % P.splits;
  const int t2 = local / P.splits;
  const int it = t2 % batch.i_tiles[pi];
  const int ot = t2 / batch.i_tiles[pi];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wo = wave >> 1, wi = wave & 1;

  const int rows_per_split = ((P.M + P.splits - 1) / P.splits + RC - 1) / RC * RC;
  const int m_begin = split * rows_per_split;
  const int m_end = min(P.M, m_begin + rows_per_split);

  const int o_base = ot * BO, i_base = it * BI;
  const int nib = max(0, min(4, (P.I - (i_base + wi * 128) + 31) >> 5));   // valid 32-col blocks of this wave
  const int nob = max(0, min(2, (P.O - (o_base + wo * 64) + 31) >> 5));

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  f32x4 bsum = {0.f, 0.f, 0.f, 0.f}, rsum = {0.f, 0.f, 0.f, 0.f};
  const bool want_vec = (P.bias_pair >= 0) && (it == 0) && (P.partial_vec != nullptr);

  // Software pipeline over (pair, row chunk): the global loads of chunk t+1 are issued into registers right before the
  // MFMA loop of chunk t and written to LDS after it, so HBM latency hides behind 128 MFMAs per wave.
  const int nchunks = (m_end - m_begin + RC - 1) / RC;
  const int total = nchunks > 0 ? P.npairs * nchunks : 0;
  const int gc = (tid & 31) << 2, gr = tid >> 5;   // G chunk: 32 rows x 128 cols, 4 float4 per thread
  const int xc = (tid & 63) << 2, xr = tid >> 6;   // X chunk: 32 rows x 256 cols, 8 float4 per thread
  f32x4 gv[4], xv[8];
  float rsv[4];
  auto fetch = [&](int t) {
    const int pr = t / nchunks, m0 = m_begin + (t - pr * nchunks) * RC;
    const float* __restrict__ G = P.G[pr];
    const float* __restrict__ X = P.X[pr];
    const int ldG = P.ldG[pr], ldX = P.ldX[pr];
    const bool vecG = ((ldG & 3) == 0) && ((reinterpret_cast<uintptr_t>(G) & 15) == 0);
    const bool vecX = ((ldX & 3) == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) gv[j] = load4_guard(G, ldG, m0 + gr + 8 * j, o_base + gc, m_end, P.O, vecG);
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = load4_guard(X, ldX, m0 + xr + 4 * j, i_base + xc, m_end, P.I, vecX);
    if (want_vec && pr == P.bias_pair && P.rowscale) {
#pragma unroll
      for (int j = 0; j < 4; ++j) rsv[j] = (m0 + gr + 8 * j < m_end) ? P.rowscale[m0 + gr + 8 * j] : 0.f;
    }
  };
  if (total > 0) fetch(0);
  for (int t = 0; t < total; ++t) {
    const int pr = t / nchunks;
    __syncthreads();   // every wave is done reading the previous chunk
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&Gs[(gr + 8 * j) * BO + gc]) = gv[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(&Xs[(xr + 4 * j) * BI + xc]) = xv[j];
    if (want_vec && pr == P.bias_pair) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bsum += gv[j];
        if (P.rowscale) rsum += gv[j] * rsv[j];
      }
    }
    __syncthreads();
    if (t + 1 < total) fetch(t + 1);   // in flight during the MFMA loop below
    if (nib > 0 && nob > 0) {
      const float* ga = &Gs[hh * BO + wo * 64 + l31];
      const float* xb = &Xs[hh * BI + wi * 128 + l31];
#pragma unroll 4
      for (int ks = 0; ks < RC / 2; ++ks) {
        float av[2], bv[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) av[a] = ga[ks * 2 * BO + a * 32];
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = xb[ks * 2 * BI + b * 32];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (b < nib) {
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- partial tile store: partial[split][o][i]
  float* __restrict__ part = P.partial + (size_t)split * P.O * P.I;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = i_base + wi * 128 + b * 32 + l31;
      if (a < nob && b < nib && i < P.I) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o_base + wo * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (o < P.O) part[(size_t)o * P.I + i] = acc[a][b][r];
        }
      }
    }
  }
  // ---- column sums of G (bias gradient) and sigma-weighted column sums (the sigma column of W1)
  if (want_vec) {
    __syncthreads();
    const int c = (tid & 31) << 2, rg = tid >> 5;
    *reinterpret_cast<f32x4*>(&Gs[rg * BO + c]) = bsum;
    *reinterpret_cast<f32x4*>(&Xs[rg * BO + c]) = rsum;
    __syncthreads();
    if (tid < BO && o_base + tid < P.O) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        s0 += Gs[g * BO + tid];
        s1 += Xs[g * BO + tid];
      }
      P.partial_vec[((size_t)split * 2 + 0) * P.O + o_base + tid] = s0;
      P.partial_vec[((size_t)split * 2 + 1) * P.O + o_base + tid] = s1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The per-image problems (M = images of the shard: context encoder, encoder trunk, decoder - a few MFLOP each) in the 128 x 256
// tiling above are ONE or two workgroups per matrix walking all rows with eight accumulators per wave: 19-22 us per launch, on the
// critical path of both updates, at any batch size.  Here every WAVE owns one 32 x 32 tile of dW (a workgroup = 2 x 2 of them), reads
// its G and X columns straight from memory (128 contiguous bytes per half-wave and row, sixteen rows in flight, the next sixteen
// requested before the MFMAs) and a 256 x 256 matrix spreads over 16 workgroups.  Writes the same partial / partial_vec layout
// (split s of `splits`), so wgrad_reduce_kernel finishes these problems like all others.
constexpr int SM_ROWS = 16;      // rows per register batch (8 MFMAs)

__global__ __launch_bounds__(256) void wgrad_small_kernel(const WgradBatchDev batch) {
  int pi = 0;
  while (pi + 1 < batch.nprob && (int)blockIdx.x >= batch.wg_begin[pi + 1]) ++pi;
  const WgradProblem& P = batch.p[pi];
  const int local = blockIdx.x - batch.wg_begin[pi];
  const int split = local % P.splits;
  const int t2 = local / P.splits;
  const int it = t2 % batch.i_tiles[pi], ot = t2 / batch.i_tiles[pi];       // 64 x 64 workgroup tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int o0 = ot * 64 + (wave >> 1) * 32, i0 = it * 64 + (wave & 1) * 32;
  if (o0 >= P.O || i0 >= P.I) return;                                         // (uniform per wave; no barriers below)
  const int rows_per_split = ((P.M + P.splits - 1) / P.splits + 1) / 2 * 2;
  const int m_begin = split * rows_per_split, m_end = min(P.M, m_begin + rows_per_split);
  const int oc = min(o0 + l31, P.O - 1), ic = min(i0 + l31, P.I - 1);        // clamped columns: their products are never stored
  const bool want_vec = P.bias_pair >= 0 && P.partial_vec != nullptr && i0 == 0;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f, rsum = 0.f;
  for (int pr = 0; pr < P.npairs; ++pr) {
    const float* __restrict__ G = P.G[pr] + oc;
    const float* __restrict__ X = P.X[pr] + ic;
    const size_t ldG = P.ldG[pr], ldX = P.ldX[pr];
    const bool vec = want_vec && pr == P.bias_pair;
    float g0[SM_ROWS / 2], x0[SM_ROWS / 2], g1[SM_ROWS / 2], x1[SM_ROWS / 2], rs0[SM_ROWS / 2], rs1[SM_ROWS / 2];
    auto fetch = [&](int m0, float (&g)[SM_ROWS / 2], float (&x)[SM_ROWS / 2], float (&rs)[SM_ROWS / 2]) {
#pragma unroll
      for (int u = 0; u < SM_ROWS / 2; ++u) {
        const int m = m0 + 2 * u + hh;
        const bool live = m < m_end;
        const int mc = live ? m : m_end - 1;
        const float gv = G[(size_t)mc * ldG], xv = X[(size_t)mc * ldX];
        g[u] = live ? gv : 0.f;
        x[u] = xv;                                                          // a dead row has g = 0
        if (vec && P.rowscale) rs[u] = P.rowscale[mc];
      }
    };
    auto mac = [&](const float (&g)[SM_ROWS / 2], const float (&x)[SM_ROWS / 2], const float (&rs)[SM_ROWS / 2]) {
#pragma unroll
      for (int u = 0; u < SM_ROWS / 2; ++u) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(g[u], x[u], acc, 0, 0, 0);
        if (vec) {
          bsum += g[u];
          if (P.rowscale) rsum += g[u] * rs[u];
        }
      }
    };
    if (m_begin < m_end) {
      fetch(m_begin, g0, x0, rs0);
      for (int m0 = m_begin; m0 < m_end; m0 += 2 * SM_ROWS) {
        if (m0 + SM_ROWS < m_end) fetch(m0 + SM_ROWS, g1, x1, rs1);
        mac(g0, x0, rs0);
        if (m0 + SM_ROWS >= m_end) break;
        if (m0 + 2 * SM_ROWS < m_end) fetch(m0 + 2 * SM_ROWS, g0, x0, rs0);
        mac(g1, x1, rs1);
      }
    }
  }
  float* __restrict__ part = P.partial + (size_t)split * P.O * P.I;
  if (i0 + l31 < P.I) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (o < P.O) part[(size_t)o * P.I + i0 + l31] = acc[r];
    }
  }
  if (want_vec) {      // the two half-waves hold the sums over even / odd rows
    bsum += __shfl_xor(bsum, 32);
    rsum += __shfl_xor(rsum, 32);
    if (hh == 0 && o0 + l31 < P.O) {
      P.partial_vec[((size_t)split * 2 + 0) * P.O + o0 + l31] = bsum;
      P.partial_vec[((size_t)split * 2 + 1) * P.O + o0 + l31] = rsum;
    }
  }
}

// out = beta*out + sum_s partial[s]  (fixed order -> reproducible).  The partial tiles are read exactly once (64 MB per
// step at config #2), so what matters is memory-level parallelism: four elements per thread as one float4 and four splits
// per iteration keep 16 values in flight (the one-element, one-split loop took 42 us for 13 us of traffic).
__device__ __forceinline__ void wgrad_reduce_store(const WgradProblem& P, size_t e, float s) {
  const int o = (int)(e / P.I), i = (int)(e - (size_t)o * P.I);
  float* dst = P.out + (size_t)o * P.ldout + i;
  *dst = (P.beta != 0.f ? P.beta * *dst : 0.f) + s;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradBatchDev batch) {
  // 256 threads = 64 consecutive float4 of the matrix x 4 split groups: group g sums splits g, g+4, ... (four of its loads
  // in flight), the groups meet in LDS in a fixed order.  One thread per element left each thread a chain of ~50 dependent
  // round trips (43 us for 64 MB); this way it is ~4.
  __shared__ f32x4 red[4][64];
  const WgradProblem& P = batch.p[blockIdx.y];
  const size_t n_mat = (size_t)P.O * P.I;
  const size_t n_vec = (P.bias_pair >= 0 && P.partial_vec) ? 2 * (size_t)P.O : 0;
  const int S = P.splits;
  const int lane64 = threadIdx.x & 63, sg = threadIdx.x >> 6;
  if ((n_mat & 3) == 0 && (reinterpret_cast<uintptr_t>(P.partial) & 15) == 0) {
    const size_t n4 = n_mat >> 2;
    for (size_t q0 = (size_t)blockIdx.x * 64; q0 < n4; q0 += (size_t)gridDim.x * 64) {   // uniform per workgroup
      const size_t q = q0 + lane64;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (q < n4) {
        const f32x4* src = reinterpret_cast<const f32x4*>(P.partial) + q;
        int sp = sg;
        for (; sp + 12 < S; sp += 16) {
          const f32x4 v0 = src[(size_t)sp * n4], v1 = src[(size_t)(sp + 4) * n4], v2 = src[(size_t)(sp + 8) * n4], v3 = src[(size_t)(sp + 12) * n4];
          acc += (v0 + v1) + (v2 + v3);
        }
        for (; sp < S; sp += 4) acc += src[(size_t)sp * n4];
      }
      red[sg][lane64] = acc;
      __syncthreads();
      if (sg == 0 && q < n4) {
        const f32x4 t = (red[0][lane64] + red[1][lane64]) + (red[2][lane64] + red[3][lane64]);
#pragma unroll
        for (int j = 0; j < 4; ++j) wgrad_reduce_store(P, 4 * q + j, t[j]);
      }
      __syncthreads();
    }
  } else {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    for (size_t e = tid; e < n_mat; e += nthr) {
      float s = 0.f;
      int sp = 0;
      for (; sp + 4 <= S; sp += 4)
        s += (P.partial[(size_t)sp * n_mat + e] + P.partial[(size_t)(sp + 1) * n_mat + e]) +
             (P.partial[(size_t)(sp + 2) * n_mat + e] + P.partial[(size_t)(sp + 3) * n_mat + e]);
      for (; sp < S; ++sp) s += P.partial[(size_t)sp * n_mat + e];
      wgrad_reduce_store(P, e, s);
    }
  }
  // bias / sigma-column sums: the same four split groups per output
  for (size_t v0 = (size_t)blockIdx.x * 64; v0 < n_vec; v0 += (size_t)gridDim.x * 64) {
    const size_t v = v0 + lane64;
    float s = 0.f;
    float* dst = nullptr;
    if (v < n_vec) {
      const int which = (int)(v / P.O), o = (int)(v % P.O);
      dst = which == 0 ? (P.out_bias ? P.out_bias + o : nullptr) : (P.out_rowscale ? P.out_rowscale + (size_t)o * P.ld_rowscale : nullptr);
      if (dst) {
        const float* pv = P.partial_vec + (size_t)which * P.O + o;
        const size_t st2 = 2 * (size_t)P.O;
        int sp = sg;
        for (; sp + 12 < S; sp += 16) s += (pv[sp * st2] + pv[(sp + 4) * st2]) + (pv[(sp + 8) * st2] + pv[(sp + 12) * st2]);
        for (; sp < S; sp += 4) s += pv[sp * st2];
      }
    }
    red[sg][lane64][0] = s;
    __syncthreads();
    if (sg == 0 && dst) {
      const float t = (red[0][lane64][0] + red[1][lane64][0]) + (red[2][lane64][0] + red[3][lane64][0]);
      *dst = (P.beta != 0.f ? P.beta * *dst : 0.f) + t;
    }
    __syncthreads();
  }
}

}  // namespace

int wgrad_splits(int M, int O, int I, int nproblems_hint) {
  // first-layer shapes (I = z_dim): the 256 x 32 geometry of wgrad_wide.hip streams G once, one workgroup per CU, and is
  // latency-bound per workgroup - give it the whole chip
  if (O % 256 == 0 && I % 32 == 0 && I <= 64 && M % RC == 0 && M >= 64 * RC) return 256 / ((O / 256) * (I / 32));
  // 256 x 256 tiles of the same kernel: a problem with many tiles (h_dim 512 / 1024: 4 / 16) must still be allowed the
  // splits that fill 256 CUs when it shares a launch with few others (the launch lowers the count to 256 / tiles in it)
  if (O % 256 == 0 && I % 256 == 0 && M % RC == 0 && M >= 64 * RC && (O / 256) * (I / 256) > 1) {
    const int t = (O / 256) * (I / 256);
    const int s_sq = t >= 32 ? 8 : 256 / (2 * t);          // at least two such problems per launch
    const int hinted = 1024 / (ceil_div(O, BO) * ceil_div(I, BI) * (nproblems_hint > 0 ? nproblems_hint : 1));
    return s_sq > hinted ? s_sq : (hinted < 1 ? 1 : hinted);
  }
  const int tiles = ceil_div(O, BO) * ceil_div(I, BI);
  const int target = 1024;   // ~2 resident workgroups per CU x 256 CUs x 2 waves of work
  int s = target / (tiles * (nproblems_hint > 0 ? nproblems_hint : 1));
  // N-row problems whose shape keeps them off the software-pipelined kernels (I = z_dim = 2 of the toy model, ragged conv
  // shapes) share THIS launch only with a few per-image problems - the regular ones of the batch run in wgrad_wide.hip - so the
  // hint over-divides: 102 workgroups streamed 268 MB in 279 us at config #1.  Give such a problem the chip.
  const bool wide_shape = O % 256 == 0 && (I % 256 == 0 || (I % 32 == 0 && I <= 64)) && M % RC == 0;
  if (!wide_shape && M >= 16384 && s < 512 / tiles) s = 512 / tiles;
  const int max_s = ceil_div(M, 2 * RC);   // at least 2 chunks of rows per split (per-image problems are latency: more, shorter workgroups)
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}

int launch_wgrad_batch(const WgradProblem* probs, int nprob, hipStream_t st) {
  ARDAE_CHECK_ARG(probs && nprob >= 1 && nprob <= WGRAD_MAX_PROBLEMS, "wgrad: 1..%d problems per batch (got %d)",
                  WGRAD_MAX_PROBLEMS, nprob);
  // the big regular problems go to the software-pipelined kernel (which may lower their split count), the rest to
  // wgrad_kernel; one reduction for all
  WgradProblem local[WGRAD_MAX_PROBLEMS];
  int wide_idx[WGRAD_MAX_PROBLEMS], nwide = 0, wide_geo[WGRAD_MAX_PROBLEMS] = {0}, wide_ntiles[WGRAD_MAX_PROBLEMS] = {0};
  size_t max_elems = 0;
  for (int i = 0; i < nprob; ++i) {
    const WgradProblem& p = probs[i];
    ARDAE_CHECK_ARG(p.M > 0 && p.O > 0 && p.I > 0, "wgrad[%d]: empty problem", i);
    ARDAE_CHECK_ARG(p.npairs >= 1 && p.npairs <= 2, "wgrad[%d]: npairs must be 1 or 2", i);
    for (int k = 0; k < p.npairs; ++k)
      ARDAE_CHECK_ARG(p.G[k] && p.X[k] && p.ldG[k] >= p.O && p.ldX[k] >= p.I, "wgrad[%d]: bad pair %d", i, k);
    ARDAE_CHECK_ARG(p.splits >= 1 && p.partial && p.out && p.ldout >= p.I, "wgrad[%d]: bad output/scratch", i);
    ARDAE_CHECK_ARG(p.bias_pair < p.npairs, "wgrad[%d]: bias_pair out of range", i);
    ARDAE_CHECK_ARG(p.bias_pair < 0 || p.partial_vec, "wgrad[%d]: bias_pair needs partial_vec", i);
    local[i] = p;
    const size_t el = (size_t)p.O * p.I + 2 * (size_t)p.O;
    if (el > max_elems) max_elems = el;
    const int geo = wgrad_wide_geometry(p);
    const int tiles = geo == 1 ? (p.O / 256) * (p.I / 256) : geo == 2 ? (p.O / 256) * (p.I / 32) : 0;
    if (geo != 0 && tiles <= 32) {
      wide_idx[nwide++] = i;
      wide_geo[i] = geo;
      wide_ntiles[i] = tiles;
    }
  }
  // one software-pipelined launch holds up to 32 tiles of a geometry (its argument block): h_dim 1024 has 16 tiles per
  // matrix, so its eleven N-row problems go out as six launches of two - each launch fills the chip through its row splits
  for (int geo = 1; geo <= 2; ++geo) {
    int group[WGRAD_MAX_PROBLEMS], ng = 0, gt = 0;
    for (int w = 0; w <= nwide; ++w) {
      const bool end = w == nwide;
      const int i = end ? -1 : wide_idx[w];
      if (!end && wide_geo[i] != geo) continue;
      if (end || gt + wide_ntiles[i] > 32) {
        if (ng > 0) {
          const int rc = launch_wgrad_wide(local, group, ng, st);
          if (rc < 0) return rc;
        }
        ng = 0; gt = 0;
      }
      if (!end) { group[ng++] = i; gt += wide_ntiles[i]; }
    }
  }
  // per-image problems (few rows): one 32 x 32 tile per wave (wgrad_small_kernel).  ARDAE_WGRAD_SMALL=0: off
  static const bool small_on = !(debug_knob("ARDAE_WGRAD_SMALL") && atoi(debug_knob("ARDAE_WGRAD_SMALL")) == 0);
  bool is_small[WGRAD_MAX_PROBLEMS] = {false};
  WgradBatchDev b;
  {
    memset(&b, 0, sizeof(b));
    int total = 0, ns = 0;
    double fl = 0, by = 0;
    for (int i = 0, w = 0; i < nprob; ++i) {
      if (w < nwide && wide_idx[w] == i) { ++w; continue; }
      const WgradProblem& p = local[i];
      if (!small_on || p.M > 1024) continue;
      is_small[i] = true;
      b.p[ns] = p;
      b.o_tiles[ns] = ceil_div(p.O, 64);
      b.i_tiles[ns] = ceil_div(p.I, 64);
      b.wg_begin[ns] = total;
      total += b.o_tiles[ns] * b.i_tiles[ns] * p.splits;
      fl += 2.0 * p.npairs * (double)p.M * p.O * p.I;
      by += 4.0 * (p.npairs * (double)p.M * (p.O + p.I) + (double)p.splits * p.O * p.I);
      ++ns;
    }
    b.nprob = ns;
    b.wg_begin[ns] = total;
    if (ns > 0) {
      if (g_prof_enabled) prof_begin(st, "wgrad_small_kernel", fl, by);
      hipLaunchKernelGGL(wgrad_small_kernel, dim3(total), dim3(256), 0, st, b);
      prof_end(st);
      ARDAE_LAUNCH_CHECK();
    }
  }
  // the remaining problems for wgrad_kernel
  memset(&b, 0, sizeof(b));
  int total = 0, nrest = 0;
  double fl = 0, by = 0;
  for (int i = 0, w = 0; i < nprob; ++i) {
    if (w < nwide && wide_idx[w] == i) { ++w; continue; }
    if (is_small[i]) continue;
    const WgradProblem& p = local[i];
    b.p[nrest] = p;
    b.o_tiles[nrest] = ceil_div(p.O, BO);
    b.i_tiles[nrest] = ceil_div(p.I, BI);
    b.wg_begin[nrest] = total;
    total += b.o_tiles[nrest] * b.i_tiles[nrest] * p.splits;
    fl += 2.0 * p.npairs * (double)p.M * p.O * p.I;
    by += 4.0 * (p.npairs * (double)p.M * (p.O + p.I) + (double)p.splits * p.O * p.I);
    ++nrest;
  }
  b.nprob = nrest;
  b.wg_begin[nrest] = total;
  if (nrest > 0) {
    if (g_prof_enabled) prof_begin(st, "wgrad_kernel", fl, by);
    hipLaunchKernelGGL(wgrad_kernel, dim3(total), dim3(256), 0, st, b);
    prof_end(st);
    ARDAE_LAUNCH_CHECK();
  }
  // reduction over every problem, with the split counts actually used
  memset(&b, 0, sizeof(b));
  b.nprob = nprob;
  for (int i = 0; i < nprob; ++i) b.p[i] = local[i];
  const int rb = (int)ceil_div64((int64_t)max_elems, 256);   // 256 elements = 64 float4 per workgroup pass
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb < 1024 ? rb : 1024, nprob), dim3(256), 0, st, b);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

}  // namespace ardae
