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

#include <algorithm>

#include "elementwise.h"
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
__device__ __forceinline__ void small_block(const LinArgs& a, int bx, int by, float (*red)[16][64]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int row0 = bx * 32, nb = by;
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

  // ---- K loop: source by source (concat inputs: [hidden | noise]); every source's STEPS of 16 k (chunk pairs) are split over the four
  //      waves and multiplied in the canonical order of SmallFrag::mac - the same sequence of products per accumulator element as in
  //      small_block_fast and small_block16, so that a row's bits do not depend on which block shape its layer ran on
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int si = 0; si < 2; ++si) {
    if (si >= a.nsrc) continue;
    SmallSrc s;
    s.x = a.src[si].x; s.wp = a.src[si].wp; s.ld = a.src[si].ld; s.K = a.src[si].K; s.kchunks = (s.K + 7) >> 3;
    s.vec = ((s.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(s.x) & 15) == 0);
    const int G = (s.kchunks + 1) >> 1;
    const int cbeg = 2 * ((G * wave) >> 2), cend = min(2 * ((G * (wave + 1)) >> 2), s.kchunks);
    if (cbeg >= cend) continue;

    auto load_chunk = [&](int c, f32x4& av, f32x4& bv) {
      const bool live = c < cend;
      const int kc = live ? c : cend - 1;       // clamped: a dead chunk re-reads a valid address and contributes A = 0
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
    auto mac = [&](const f32x4 (&av)[SB], const f32x4 (&bv)[SB]) {
#pragma unroll
      for (int u = 0; u < SB; u += 2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][q], bv[u][q], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u + 1][q], bv[u + 1][q], acc, 0, 0, 0);
        }
    };

    f32x4 a0[SB], b0[SB], a1[SB], b1[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) load_chunk(cbeg + u, a0[u], b0[u]);
    for (int c = cbeg; c < cend; c += 2 * SB) {
      const bool more1 = c + SB < cend;
      if (more1) {
#pragma unroll
        for (int u = 0; u < SB; ++u) load_chunk(c + SB + u, a1[u], b1[u]);
      }
      mac(a0, b0);
      if (!more1) break;
      if (c + 2 * SB < cend) {
#pragma unroll
        for (int u = 0; u < SB; ++u) load_chunk(c + 2 * SB + u, a0[u], b0[u]);
      }
      mac(a1, b1);
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

// The REGULAR case of small_block - one source, K % 8 == 0, 16-byte aligned rows, M % 32 == 0, Nout % 32 == 0 (every per-image layer of the
// mlp configurations except those touching the 784 / 100-wide data and noise) - without the generic block's clamps, per-lane source
// selects and scalar-tail loads: those made it ~3000 instructions (190 exec-mask branches) per wave for 32 MFMAs, i.e. the per-image
// launches were ISSUE-bound (3.7 us of work per 256 x 256 level, measured inside the chain kernel with the hand-over switched off),
// not latency-bound.  Same chunk ranges per wave, same MFMA order, same reduction tree: bit-identical to small_block.
// COH: the rows read were written by OTHER workgroups of this launch (linear_small_chain_kernel) - every load of them bypasses the CU's
// L1 (`nt`; a CU's L1 is never refreshed by another CU's stores), the weights and biases are launch constants and load normally.
template <bool COH, typename T>
__device__ __forceinline__ T ld_row(const T* p) {
  if (COH) return __builtin_nontemporal_load(p);
  return *p;
}

template <int N>
struct IntTag { static constexpr int value = N; };

template <int NB>
struct SmallFrag {            // NB chunks (8 k each) of A and packed-B fragments
  f32x4 a[NB], b[NB];
  template <bool COH>
  __device__ __forceinline__ void load(const f32x4*& ap, const f32x4*& bp) {
#pragma unroll
    for (int u = 0; u < NB; ++u) { a[u] = ld_row<COH>(ap + 2 * u); b[u] = bp[64 * u]; }
    ap += 2 * NB; bp += 64 * NB;
  }
  // CANONICAL k ORDER (round 4; see the note above small_block16): a step of 16 k = the chunk pair (2 g, 2 g + 1), and inside it MFMA j of
  // chunk 2 g, then MFMA j of chunk 2 g + 1, j = 0 .. 3 - the products enter the accumulator in the order k = 16 g + j + {0, 4, 8, 12},
  // which is the order ONE v_mfma_f32_16x16x4_f32 of small_block16 adds them in.  A batch starts at an even chunk; a last odd chunk
  // (K % 16 == 8) runs alone, where the 16 x 16 block multiplies zeros.
  __device__ __forceinline__ void mac(f32x16& acc) const {
    static_assert(NB == 1 || NB % 2 == 0, "chunk pairs");
    if (NB == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][q], b[0][q], acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u + 1 < NB; u += 2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][q], b[u][q], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u + 1][q], b[u + 1][q], acc, 0, 0, 0);
        }
    }
  }
};

template <int EPI, int ACT, bool COH, bool STAMP = false>
__device__ __forceinline__ void small_block_fast(const LinArgs& a, int bx, int by, float (*red)[16][64], unsigned long long* stamp = nullptr) {
  auto mark = [&](int i) {      // timing experiment (scratch/exp_small_chain.py, ARDAE_SC_DEBUG=8): wave 0 of one workgroup at one level
    if (STAMP) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) stamp[i] = __builtin_readcyclecounter();
    }
  };
  mark(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // uniform: the chunk range below is scalar
  const int l31 = lane & 31, hh = lane >> 5;
  const int col = by * 32 + l31, rowb = bx * 32 + 8 * wave + 4 * hh;      // this lane finishes rows rowb .. rowb + 3 of column col
  float o1[4], o2[4], bcol = 0.f, wsig = 0.f, wv = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) o1[i] = 0.f, o2[i] = 0.f;
  if (EPI == EPI_ACT) {
    if (a.bias) bcol = a.bias[col];
    if (a.rowbias) {
      const int rpg = a.rows_per_group, g0 = rpg == 1 ? rowb : rowb / rpg, rem = rowb - g0 * rpg;      // one division for the four rows
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int g = rpg == 1 ? rowb + i : (rpg >= 4 ? g0 + (rem + i >= rpg ? 1 : 0) : (rowb + i) / rpg);
        o1[i] = ld_row<COH>(a.rowbias + (size_t)g * a.rowbias_ld + col);
      }
    }
    if (a.rowscale) {
      wsig = a.rowscale_w[col];
#pragma unroll
      for (int i = 0; i < 4; ++i) o2[i] = ld_row<COH>(a.rowscale + rowb + i);
    }
    if (a.Y2) wv = a.R[col];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) o1[i] = ld_row<COH>(a.S + (size_t)(rowb + i) * a.ldS + col);
    if (EPI == EPI_CHAIN) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o2[i] = ld_row<COH>(a.R + (size_t)(rowb + i) * a.ldR + col);
    } else if (a.Q) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o2[i] = ld_row<COH>(a.Q + (size_t)(rowb + i) * a.ldQ + col);
    }
  }

  const int K = a.src[0].K, T = K >> 3, G = (T + 1) >> 1;     // chunks of 8 k; steps of 16 k (the last one may hold one chunk)
  const int cbeg = 2 * ((G * wave) >> 2), cend = min(2 * ((G * (wave + 1)) >> 2), T);      // the wave's quarter of the STEPS, as in small_block16
  const f32x4* ap = reinterpret_cast<const f32x4*>(a.src[0].x + (size_t)(bx * 32 + l31) * a.src[0].ld + 4 * hh) + 2 * cbeg;   // + 2 per chunk
  const f32x4* bp = reinterpret_cast<const f32x4*>(a.src[0].wp) + ((size_t)by * T + cbeg) * 64 + lane;                             // + 64 per chunk
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  mark(1);      // epilogue operands have landed
  // whole batches of SB chunks, the next batch's loads in flight behind this one's MFMAs; then the remainder as static batches of 4 / 2 / 1
  int left = cend - cbeg;
  if (left >= SB) {
    SmallFrag<SB> f0, f1;
    f0.template load<COH>(ap, bp);
    mark(2);    // first batch of fragments has landed
    left -= SB;
    while (left >= 2 * SB) {
      f1.template load<COH>(ap, bp);
      f0.mac(acc);
      f0.template load<COH>(ap, bp);
      f1.mac(acc);
      left -= 2 * SB;
    }
    if (left >= SB) {
      f1.template load<COH>(ap, bp);
      f0.mac(acc);
      f1.mac(acc);
      left -= SB;
    } else {
      f0.mac(acc);
    }
  }
  if (left & 4) { SmallFrag<4> f; f.template load<COH>(ap, bp); f.mac(acc); }
  if (left & 2) { SmallFrag<2> f; f.template load<COH>(ap, bp); f.mac(acc); }
  if (left & 1) { SmallFrag<1> f; f.template load<COH>(ap, bp); f.mac(acc); }

  if (STAMP) {      // behind the last MFMA's 16 passes
    asm volatile("s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\ns_nop 7" ::: "memory");
    if (threadIdx.x == 0) stamp[3] = __builtin_readcyclecounter() + (unsigned long long)(acc[0] == 123.456f);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    v[i] = (red[0][4 * wave + i][lane] + red[1][4 * wave + i][lane]) + (red[2][4 * wave + i][lane] + red[3][4 * wave + i][lane]);
  mark(4);      // partial sums reduced
  float* yo = a.Y + (size_t)rowb * a.ldY + col;
  if (EPI == EPI_ACT) {
    float y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = act_fwd<ACT>(v[i] + bcol + o1[i] + o2[i] * wsig);
#pragma unroll
    for (int i = 0; i < 4; ++i) yo[(size_t)i * a.ldY] = y[i];
    if (a.Y2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a.Y2[(size_t)(rowb + i) * a.ldY2 + col] = -wv * act_d1<ACT>(y[i]);
    }
  } else if (EPI == EPI_DACT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) yo[(size_t)i * a.ldY] = v[i] * act_d1<ACT>(o1[i]) + o2[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float em = act_ratio<ACT>(o1[i]);
      yo[(size_t)i * a.ldY] = v[i] * act_d1<ACT>(o1[i]);
      a.Y2[(size_t)(rowb + i) * a.ldY2 + col] = v[i] * o2[i] * em;
    }
  }
  mark(5);      // results stored and acknowledged
}

// FEW rows (at most 64 blocks of 32 x 32 in the layer - the 64 / 128-image shards of the 8 / 4-GPU runs): a 32 x 32 block at K = 256 pulls
// 64 KiB through ONE CU's 64 B / clk vector-memory path - ~1.7 k of the block's 8.4 k cycles (in-kernel stamps, scratch/exp_small_chain.py) -
// and runs a chain of 32 dependent 64-cycle MFMAs (another 3 k) while most of the chip idles.  Here a workgroup owns a 16 x 16 block:
// four times the workgroups, a quarter of the bytes and of the matrix time each (v_mfma_f32_16x16x4_f32, 16 per wave at K = 256).
// Lane (c = lane % 16, q = lane / 16) of wave w, step g (16 k): ONE float4 of A (row c, k = 16 g + 4 q ..+3: the four q-lanes of a row read
// 64 contiguous bytes) and ONE float4 of the packed weights (chunk 2 g + q / 2, half q % 2: the same four k), then four MFMAs - MFMA j
// multiplies k = 16 g + 4 q + j over q = 0..3.  An FP32 MFMA is a chain of fused multiply-adds over its k in ascending order, for both
// shapes (scratch/mfma/order.hip on an MI355X: 204800 / 204800 and 51200 / 51200 elements equal to that model bit for bit, and one
// 16 x 16 x 4 == two 32 x 32 x 2 on k pairs (0, 1), (2, 3)), so the 32 x 32 blocks can - and since round 4 do - add the SAME sequence of
// products per output element: same split of a source's 16-k steps over the four waves, inside a step k = 16 g + j + {0, 4, 8, 12} for
// j = 0 .. 3 (SmallFrag::mac), same (w0 + w1) + (w2 + w3) tree over the waves' partial sums.  A row's bits therefore do not depend on the
// block shape, i.e. not on how many rows (images per rank) the layer has: tests/test_linear_gpu.py::test_linear_per_image_blocks_bit_identical.
template <int EPI, int ACT, bool COH>
__device__ __forceinline__ void small_block16(const LinArgs& a, int bx, int by, float (*red)[4][64]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int col_raw = by * 16 + c16, row = bx * 16 + 4 * q + wave;   // this lane finishes (row, col): accumulator register `wave` of lane
  const bool cok = col_raw < a.Nout;                                  // ragged column count (h = 300 of the shipped aux recipe, the 100 noise columns)
  const int col = cok ? col_raw : a.Nout - 1;
  float o1 = 0.f, o2 = 0.f, bcol = 0.f, wsig = 0.f, wv = 0.f;
  if (EPI == EPI_ACT) {
    if (a.bias) bcol = a.bias[col];
    if (a.rowbias) o1 = ld_row<COH>(a.rowbias + (size_t)(a.rows_per_group == 1 ? row : row / a.rows_per_group) * a.rowbias_ld + col);
    if (a.rowscale) { wsig = a.rowscale_w[col]; o2 = ld_row<COH>(a.rowscale + row); }
    if (a.Y2) wv = a.R[col];
  } else {
    o1 = ld_row<COH>(a.S + (size_t)row * a.ldS + col);
    if (EPI == EPI_CHAIN) o2 = ld_row<COH>(a.R + (size_t)row * a.ldR + col);
    else if (a.Q) o2 = ld_row<COH>(a.Q + (size_t)row * a.ldQ + col);
  }

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // one or two sources (concat inputs: [hidden | noise]); every source's steps of 16 k are split over the four waves.  K % 4 == 0: the
  // source's last step may be partial - its lanes beyond K load nothing (neither the row, which ends there, nor weight chunks the packed
  // image does not have) and multiply zeros
#pragma unroll
  for (int s = 0; s < 2; ++s) {       // (static indices: the chain kernel's argument struct must stay in registers)
    if (s >= a.nsrc) continue;
    const int K = a.src[s].K, T = (K + 7) >> 3, Gf = K >> 4, tail = K & 15;     // packed chunks of 8 k; full steps of 16 k; k of a partial last step
    const int G = Gf + (tail ? 1 : 0);
    const int gbeg = (G * wave) >> 2, gend = (G * (wave + 1)) >> 2;
    const f32x4* ap = reinterpret_cast<const f32x4*>(a.src[s].x + (size_t)(bx * 16 + c16) * a.src[s].ld + 4 * q) + 4 * gbeg;          // + 4 per step
    const f32x4* bp = reinterpret_cast<const f32x4*>(a.src[s].wp) + ((size_t)(by >> 1) * T + 2 * gbeg + (q >> 1)) * 64 + (by & 1) * 16 + c16 + 32 * (q & 1);   // + 128 per step
    auto run = [&](auto nb) {
      constexpr int NB = decltype(nb)::value;
      f32x4 av[NB], bv[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) { av[u] = ld_row<COH>(ap + 4 * u); bv[u] = bp[128 * u]; }
      ap += 4 * NB; bp += 128 * NB;
#pragma unroll
      for (int u = 0; u < NB; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][j], bv[u][j], acc, 0, 0, 0);
    };
    const bool has_tail = tail != 0 && gend == G && gend > gbeg;      // (uniform per wave) the wave that owns the source's last step
    int left = gend - gbeg - (has_tail ? 1 : 0);
    for (; left >= 8; left -= 8) run(IntTag<8>{});
    if (left & 4) run(IntTag<4>{});
    if (left & 2) run(IntTag<2>{});
    if (left & 1) run(IntTag<1>{});
    if (has_tail) {
      f32x4 av = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
      if (4 * q < tail) { av = ld_row<COH>(ap); bv = bp[0]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
    }
  }

  // partial sums of the four waves meet in LDS; wave w finishes accumulator register w of every lane: rows 4 q + w
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  const float v = (red[0][wave][lane] + red[1][wave][lane]) + (red[2][wave][lane] + red[3][wave][lane]);
  if (!cok) return;
  if (EPI == EPI_ACT) {
    const float y = act_fwd<ACT>(v + bcol + o1 + o2 * wsig);
    a.Y[(size_t)row * a.ldY + col] = y;
    if (a.Y2) a.Y2[(size_t)row * a.ldY2 + col] = -wv * act_d1<ACT>(y);
  } else if (EPI == EPI_DACT) {
    a.Y[(size_t)row * a.ldY + col] = v * act_d1<ACT>(o1) + o2;
  } else {
    const float em = act_ratio<ACT>(o1);
    a.Y[(size_t)row * a.ldY + col] = v * act_d1<ACT>(o1);
    a.Y2[(size_t)row * a.ldY2 + col] = v * o2 * em;
  }
}

// what small_block_fast assumes (host side)
inline bool small_fast_on() {     // ARDAE_SMALL_FAST=0: the generic block everywhere (A/B)
  static const bool on = !(debug_knob("ARDAE_SMALL_FAST") && atoi(debug_knob("ARDAE_SMALL_FAST")) == 0);
  return on;
}
inline bool small_regular(const LinArgs& a) {
  return a.nsrc == 1 && a.src[0].K >= 8 && a.src[0].K % 8 == 0 && a.src[0].ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a.src[0].x) & 15) == 0 &&
         a.M % 32 == 0 && a.Nout % 32 == 0;
}

// what small_block16 assumes, and when it pays: the layer has few 32 x 32 blocks (ARDAE_SMALL16_MAX_BLOCKS, default 64)
inline bool small_regular16(const LinArgs& a) {
  static const int max_blocks = debug_knob("ARDAE_SMALL16_MAX_BLOCKS") ? atoi(debug_knob("ARDAE_SMALL16_MAX_BLOCKS")) : 64;
  if (a.nsrc < 1 || a.nsrc > 2 || a.M % 16 || a.Nout < 1 || (int64_t)ceil_div(a.M, 32) * ceil_div(a.Nout, 32) > max_blocks) return false;
  for (int s = 0; s < a.nsrc; ++s)      // float4 fragments: rows 16-byte aligned, K a multiple of 4 (a partial last step of 16 k is fine)
    if (a.src[s].K < 4 || a.src[s].K % 4 || a.src[s].ld % 4 || (reinterpret_cast<uintptr_t>(a.src[s].x) & 15)) return false;
  return true;
}

template <int EPI, int ACT>
__global__ __launch_bounds__(256) void linear_small16_kernel(const LinArgs a) {
  __shared__ float red[4][4][64];
  small_block16<EPI, ACT, false>(a, blockIdx.x, blockIdx.y, red);
}
template <int EPI, int ACT>
__global__ __launch_bounds__(256) void linear_small16_pair_kernel(const LinArgs a0, const LinArgs a1) {
  const LinArgs& a = blockIdx.z ? a1 : a0;
  if ((int)blockIdx.x * 16 >= a.M || (int)blockIdx.y * 16 >= a.Nout) return;
  __shared__ float red[4][4][64];
  small_block16<EPI, ACT, false>(a, blockIdx.x, blockIdx.y, red);
}

template <int EPI, int ACT>
__global__ __launch_bounds__(256) void linear_small_fast_kernel(const LinArgs a) {
  __shared__ float red[4][16][64];
  small_block_fast<EPI, ACT, false>(a, blockIdx.x, blockIdx.y, red);
}
template <int EPI, int ACT>
__global__ __launch_bounds__(256) void linear_small_fast_pair_kernel(const LinArgs a0, const LinArgs a1) {
  const LinArgs& a = blockIdx.z ? a1 : a0;
  if ((int)blockIdx.x * 32 >= a.M || (int)blockIdx.y * 32 >= a.Nout) return;
  __shared__ float red[4][16][64];
  small_block_fast<EPI, ACT, false>(a, blockIdx.x, blockIdx.y, red);
}

template <int EPI, int ACT>
__global__ __launch_bounds__(256) void linear_small_kernel(const LinArgs a) {
  __shared__ float red[4][16][64];
  small_block<EPI, ACT>(a, blockIdx.x, blockIdx.y, red);
}
// TWO independent problems of one epilogue kind in one launch (blockIdx.z picks): links of two dependent chains that do not depend
// on each other - the context encoder and the input encoder of the sigma = 0 score pass (models/graddae/mlp.py:426-431) - cost one
// launch latency per level instead of two
template <int EPI, int ACT>
__global__ __launch_bounds__(256) void linear_small_pair_kernel(const LinArgs a0, const LinArgs a1) {
  const LinArgs& a = blockIdx.z ? a1 : a0;
  if ((int)blockIdx.x * 32 >= a.M || (int)blockIdx.y * 32 >= a.Nout) return;      // the grid covers the larger problem
  __shared__ float red[4][16][64];
  small_block<EPI, ACT>(a, blockIdx.x, blockIdx.y, red);
}

// ---------------------------------------------------------------------------------------------------------------------
// A whole dependent CHAIN of per-image layers in ONE launch.  The per-image (B-row) passes of a step - the sigma = 0 score pass of
// the VAE update, encoder trunk, context encoder and their backward chains - are ~40 launches on the step's critical path, each
// 1 us of MFMA inside 5.5 us (8.5 us at 512 rows) of launch-to-launch latency.  Here the workgroups of a launch walk the LEVELS of a
// chain: a level holds one or two independent problems, workgroup (rb, j) computes 32 x 32 block j of row block rb at every level and
// then waits until all blocks of ITS ROW BLOCK at that level are done (a layer is row-local in its input: block (rb, *) of level l + 1
// reads rows rb of level l only) - a counter per row block, released / acquired at agent scope, instead of a kernel boundary.
// Row block rb's workgroups are placed on ONE XCD (block index i runs on XCD i % 8 - scratch/mfma/hwid.hip - so i = (rb / 8) * 8 nj +
// j * 8 + rb % 8 puts all nj blocks of row block rb on XCD rb % 8): their hand-over stays inside that XCD's L2.
// Deadlock freedom: a workgroup waits only for workgroups of its OWN launch, so the launch must fit the device as a whole - every
// workgroup resident at once, whatever order the dispatcher picks them in.  The host admits a chain launch only when its grid is at most
// (workgroups of this kernel one CU holds, hipOccupancyMaxActiveBlocksPerMultiprocessor) x (CUs of the device), queried once
// (sc_resident_cap); larger problems go out one launch per problem.  Other kernels on the device finish without this one, so their CUs
// free up and every workgroup of the grid is dispatched eventually.
constexpr int SC_CNT_STRIDE = LINEAR_SMALL_CHAIN_COUNTER_WORDS;     // counters of different row blocks never share a cache line
constexpr int SC_MAXLEV = 16, SC_MAXPROB = 20;     // the argument block stays under the 4 KiB kernel-argument limit
struct ScProblem {            // what small_block reads of ardae_linear_args, 104 bytes
  const float *x0, *wp0, *x1, *wp1, *bias, *rowbias, *rowscale, *rowscale_w, *S, *Q, *R;
  float *Y, *Y2;
  int ld0, K0, ld1, K1, nsrc, Nout, act, epi, rowbias_ld, rows_per_group, ldS, ldQ, ldR, ldY, ldY2;
};
struct ScLevel { int np, nblk0, nblk; };        // problems, 32-column blocks of problem 0, of both
struct ScArgs {
  int nlev, M, nj, fast, nrb; // nj: workgroups per row block (max blocks of a level); fast: XCD-local hand-over allowed; nrb: row blocks
  int blk;                    // block edge: 32 (small_block_fast) or 16 (small_block16)
  unsigned* cnt;              // [nrb][32] zeroed before the launch: a 128-byte line per row block (level counter, XCC mask, arrivals)
  ScLevel lv[SC_MAXLEV];
  ScProblem pr[SC_MAXPROB];
  int first[SC_MAXLEV];       // index of level l's first problem in pr
};

__device__ __forceinline__ LinArgs sc_args(const ScProblem& q, int M) {
  LinArgs a;
  a.M = M; a.Nout = q.Nout; a.nsrc = q.nsrc; a.act = q.act;
  a.src[0].x = q.x0; a.src[0].ld = q.ld0; a.src[0].K = q.K0; a.src[0].wp = q.wp0;
  a.src[1].x = q.x1; a.src[1].ld = q.ld1; a.src[1].K = q.K1; a.src[1].wp = q.wp1;
  a.bias = q.bias; a.rowbias = q.rowbias; a.rowbias_ld = q.rowbias_ld; a.rows_per_group = q.rows_per_group;
  a.rowscale = q.rowscale; a.rowscale_w = q.rowscale_w;
  a.S = q.S; a.ldS = q.ldS; a.R = q.R; a.ldR = q.ldR; a.Q = q.Q; a.ldQ = q.ldQ;
  a.sigma = nullptr; a.eps = nullptr; a.ldeps = 0; a.scale = 0.f;
  a.Y = q.Y; a.ldY = q.ldY; a.Y2 = q.Y2; a.ldY2 = q.ldY2; a.colsum = nullptr; a.tile_loss = nullptr;
  return a;
}

__global__ __launch_bounds__(256) void linear_small_chain_kernel(const ScArgs c) {
  __shared__ float red[4][16][64];
  const int i = (int)blockIdx.x, nj = c.nj;
  const int rb = (i / (8 * nj)) * 8 + (i & 7), j = (i / 8) % nj;     // see above: row block rb's workgroups share an XCD
  if (rb * c.blk >= c.M) return;
  // Is the placement what the fast hand-over assumes?  Every workgroup of the row block reports its XCC id (atomics only, no data
  // involved) and waits for the others: one XCC -> the hand-over may stay inside that XCD's L2 (vmcnt + L1 invalidate); anything else
  // (another dispatch order, a partitioned device) -> agent-scope release / acquire (L2 write-back + invalidate: correct, slow).
  // the levels' descriptors are read from the kernel-argument segment level by level: touch every line of it now (one round trip for all)
  // instead of one scalar-cache miss chain per level.  ONE asm statement, loads and their wait together: the destination is a scratch
  // SGPR the compiler must not hand to anything else while a load is still in flight.
  {
    static_assert(sizeof(ScArgs) >= 56 * 64, "prefetch range below");
    unsigned t;
#define SC_PF(o) "s_load_dword %0, %1, " #o "\n"
#define SC_PF8(a, b, c, d, e, f, g, h) SC_PF(a) SC_PF(b) SC_PF(c) SC_PF(d) SC_PF(e) SC_PF(f) SC_PF(g) SC_PF(h)
    asm volatile(SC_PF8(0x000, 0x040, 0x080, 0x0c0, 0x100, 0x140, 0x180, 0x1c0) SC_PF8(0x200, 0x240, 0x280, 0x2c0, 0x300, 0x340, 0x380, 0x3c0)
                 SC_PF8(0x400, 0x440, 0x480, 0x4c0, 0x500, 0x540, 0x580, 0x5c0) SC_PF8(0x600, 0x640, 0x680, 0x6c0, 0x700, 0x740, 0x780, 0x7c0)
                 SC_PF8(0x800, 0x840, 0x880, 0x8c0, 0x900, 0x940, 0x980, 0x9c0) SC_PF8(0xa00, 0xa40, 0xa80, 0xac0, 0xb00, 0xb40, 0xb80, 0xbc0)
                 SC_PF8(0xc00, 0xc40, 0xc80, 0xcc0, 0xd00, 0xd40, 0xd80, 0xdc0) "s_waitcnt lgkmcnt(0)"
                 : "=&s"(t)
                 : "s"(__builtin_amdgcn_kernarg_segment_ptr())
                 : "memory");
#undef SC_PF8
#undef SC_PF
  }
  bool fast = false;
  {
    __shared__ unsigned s_mask;
    if (threadIdx.x == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      // the mask bit is published BEFORE the arrival (release), and the mask is read AFTER the last arrival has been seen (acquire): a
      // reader that counts nj arrivals sees all nj mask bits, so every workgroup of the row block takes the same decision (once per launch)
      __hip_atomic_fetch_or(c.cnt + SC_CNT_STRIDE * rb + 1, 1u << (xcc & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(c.cnt + SC_CNT_STRIDE * rb + 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(c.cnt + SC_CNT_STRIDE * rb + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nj) __builtin_amdgcn_s_sleep(1);
      s_mask = __hip_atomic_load(c.cnt + SC_CNT_STRIDE * rb + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    fast = (c.fast & 1) && __builtin_popcount(s_mask) == 1;
  }
  unsigned target = 0;
  for (int l = 0; l < c.nlev; ++l) {
    if ((c.fast & 8) && l == 5 && i == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(c.cnt + 8)[7] = __builtin_readcyclecounter();
    const ScLevel lv = c.lv[l];
    if (j < lv.nblk && !(c.fast & 2)) {
      const bool second = j >= lv.nblk0;
      const LinArgs a = sc_args(c.pr[c.first[l] + (second ? 1 : 0)], c.M);
      const int cb = second ? j - lv.nblk0 : j;
      const int key = a.act * 4 + (int)(c.pr[c.first[l] + (second ? 1 : 0)].epi);
      if (c.blk == 16) {     // few rows: 16 x 16 blocks (small_block16)
        float (*red16)[4][64] = reinterpret_cast<float (*)[4][64]>(&red[0][0][0]);
        switch (key) {
          case ACT_NONE * 4 + EPI_ACT: small_block16<EPI_ACT, ACT_NONE, true>(a, rb, cb, red16); break;
          case ACT_RELU * 4 + EPI_ACT: small_block16<EPI_ACT, ACT_RELU, true>(a, rb, cb, red16); break;
          case ACT_SOFTPLUS * 4 + EPI_ACT: small_block16<EPI_ACT, ACT_SOFTPLUS, true>(a, rb, cb, red16); break;
          case ACT_NONE * 4 + EPI_DACT: small_block16<EPI_DACT, ACT_NONE, true>(a, rb, cb, red16); break;
          case ACT_RELU * 4 + EPI_DACT: small_block16<EPI_DACT, ACT_RELU, true>(a, rb, cb, red16); break;
          default: small_block16<EPI_DACT, ACT_SOFTPLUS, true>(a, rb, cb, red16); break;
        }
      } else
      switch (key) {         // regular problems only (launch_linear_small_chain): the lean block, its row loads bypassing the L1
        case ACT_NONE * 4 + EPI_ACT: small_block_fast<EPI_ACT, ACT_NONE, true>(a, rb, cb, red); break;
        case ACT_RELU * 4 + EPI_ACT: small_block_fast<EPI_ACT, ACT_RELU, true>(a, rb, cb, red); break;
        case ACT_SOFTPLUS * 4 + EPI_ACT:
          if ((c.fast & 8) && l == 5 && i == 0) {
            unsigned long long* stamp = reinterpret_cast<unsigned long long*>(c.cnt + 8);     // words 8 .. 23 of row block 0's counter line
            if (c.fast & 16) small_block_fast<EPI_ACT, ACT_SOFTPLUS, false, true>(a, rb, cb, red, stamp);
            else small_block_fast<EPI_ACT, ACT_SOFTPLUS, true, true>(a, rb, cb, red, stamp);
            if (threadIdx.x == 0) stamp[6] = __builtin_readcyclecounter();
          } else {
            small_block_fast<EPI_ACT, ACT_SOFTPLUS, true>(a, rb, cb, red);
          }
          break;
        case ACT_NONE * 4 + EPI_DACT: small_block_fast<EPI_DACT, ACT_NONE, true>(a, rb, cb, red); break;
        case ACT_RELU * 4 + EPI_DACT: small_block_fast<EPI_DACT, ACT_RELU, true>(a, rb, cb, red); break;
        default: small_block_fast<EPI_DACT, ACT_SOFTPLUS, true>(a, rb, cb, red); break;
      }
    }
    if (l + 1 == c.nlev) break;                     // the kernel boundary orders the last level
    target += (unsigned)lv.nblk;
    // release: this workgroup's stores are visible at agent scope before its arrival is; acquire: nothing of the next level is read
    // before every block of this row block has arrived
    // (every wave releases its OWN stores - a workgroup barrier does not wait for another wave's stores to reach the L2)
    if (c.fast & 4) {              // timing experiment only (scratch/exp_small_chain.py): no hand-over at all
      __syncthreads();
    } else if (fast) {
      // all workgroups of the row block share an XCD (verified above), so the hand-over only has to reach that XCD's L2: the L1 is
      // write-through (a store is complete when the L2 has it: vmcnt), and the consumer's loads of handed-over rows bypass its CU's L1
      // (small_block_fast<.., COH = true>; a workgroup-scope fence / `buffer_inv sc0` would invalidate nothing another CU wrote)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        if (j < lv.nblk) __hip_atomic_fetch_add(c.cnt + SC_CNT_STRIDE * rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(c.cnt + SC_CNT_STRIDE * rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      if (threadIdx.x == 0) {
        if (j < lv.nblk) __hip_atomic_fetch_add(c.cnt + SC_CNT_STRIDE * rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(c.cnt + SC_CNT_STRIDE * rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
  }
}

template <int EPI, int ACT>
int launch_small_pair(const LinArgs& a0, const LinArgs& a1, hipStream_t st) {
  if (g_prof_enabled) {
    char name[64];
    snprintf(name, sizeof(name), "linear_small_pair_kernel<%d, %d>", EPI, ACT);
    double fl = 0, by = 0;
    for (const LinArgs* a : {&a0, &a1}) {
      double ksum = 0;
      for (int s = 0; s < a->nsrc; ++s) ksum += a->src[s].K;
      fl += 2.0 * a->M * (double)a->Nout * ksum;
      by += 4.0 * ((double)a->M * ksum + 2.0 * a->M * (double)a->Nout + ksum * a->Nout);
    }
    prof_begin(st, name, fl, by);
  }
  const dim3 grid(std::max(ceil_div(a0.M, 32), ceil_div(a1.M, 32)), std::max(ceil_div(a0.Nout, 32), ceil_div(a1.Nout, 32)), 2);
  if (small_fast_on() && small_regular16(a0) && small_regular16(a1))
    hipLaunchKernelGGL((linear_small16_pair_kernel<EPI, ACT>), dim3(std::max(a0.M, a1.M) / 16, ceil_div(std::max(a0.Nout, a1.Nout), 16), 2), dim3(256), 0, st, a0, a1);
  else if (small_fast_on() && small_regular(a0) && small_regular(a1))
    hipLaunchKernelGGL((linear_small_fast_pair_kernel<EPI, ACT>), grid, dim3(256), 0, st, a0, a1);
  else
    hipLaunchKernelGGL((linear_small_pair_kernel<EPI, ACT>), grid, dim3(256), 0, st, a0, a1);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
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
  if (small_fast_on() && small_regular16(a))
    hipLaunchKernelGGL((linear_small16_kernel<EPI, ACT>), dim3(a.M / 16, ceil_div(a.Nout, 16)), dim3(256), 0, st, a);
  else if (small_fast_on() && small_regular(a))
    hipLaunchKernelGGL((linear_small_fast_kernel<EPI, ACT>), dim3(a.M / 32, a.Nout / 32), dim3(256), 0, st, a);
  else
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

namespace {
// How many workgroups of linear_small_chain_kernel the device holds AT ONCE (see "Deadlock freedom" above).  ARDAE_SC_CAP (debug knob):
// another value, for the test of the fallback.
int sc_resident_cap() {
  static const int cap = [] {
    if (debug_knob("ARDAE_SC_CAP")) return atoi(debug_knob("ARDAE_SC_CAP"));
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, linear_small_chain_kernel, 256, 0) != hipSuccess)
      return 0;      // unknown: never chain
    return cus * per_cu;
  }();
  return cap;
}

// [p, p + bytes) of two buffers overlap
inline bool sc_overlap(const float* a, size_t na, const float* b, size_t nb) {
  if (!a || !b || na == 0 || nb == 0) return false;
  const uintptr_t a0 = reinterpret_cast<uintptr_t>(a), b0 = reinterpret_cast<uintptr_t>(b);
  return a0 < b0 + nb * sizeof(float) && b0 < a0 + na * sizeof(float);
}
// The hand-over inside one XCD relies on every chained buffer being written exactly ONCE before any read of it in the launch (the
// consumers' `nt` loads must find nothing stale): a problem's outputs may alias nothing an EARLIER or the same level reads or writes.
bool sc_write_once(const LinArgs* probs, const int* level_of, int nprob) {
  auto extent = [](const LinArgs& a, int ld, int cols) { return (size_t)(a.M - 1) * (size_t)ld + (size_t)cols; };
  for (int i = 0; i < nprob; ++i) {
    const LinArgs& w = probs[i];
    const float* outs[2] = {w.Y, w.Y2};
    const size_t outn[2] = {extent(w, w.ldY, w.Nout), w.Y2 ? extent(w, w.ldY2, w.Nout) : 0};
    for (int j = 0; j < nprob; ++j) {
      if (level_of[j] > level_of[i]) continue;             // later levels READ what this one writes: that is the chain
      const LinArgs& r = probs[j];
      for (int o = 0; o < 2; ++o) {
        if (!outs[o]) continue;
        if (j != i && (sc_overlap(outs[o], outn[o], r.Y, extent(r, r.ldY, r.Nout)) || (r.Y2 && sc_overlap(outs[o], outn[o], r.Y2, extent(r, r.ldY2, r.Nout)))))
          return false;
        for (int s = 0; s < r.nsrc; ++s)
          if (sc_overlap(outs[o], outn[o], r.src[s].x, extent(r, r.src[s].ld, r.src[s].K))) return false;
        if (sc_overlap(outs[o], outn[o], r.S, r.S ? extent(r, r.ldS, r.Nout) : 0)) return false;
        if (sc_overlap(outs[o], outn[o], r.Q, r.Q ? extent(r, r.ldQ, r.Nout) : 0)) return false;
        if (r.rowbias && sc_overlap(outs[o], outn[o], r.rowbias, (size_t)((r.M - 1) / std::max(r.rows_per_group, 1)) * r.rowbias_ld + r.Nout)) return false;
        if (r.rowscale && sc_overlap(outs[o], outn[o], r.rowscale, (size_t)r.M)) return false;
      }
      if (j == i && w.Y2 && sc_overlap(w.Y, outn[0], w.Y2, outn[1])) return false;
    }
  }
  return true;
}

bool sc_fill(const LinArgs& a, int epi, ScProblem& q) {
  if (!linear_small_eligible(a, epi) || a.nsrc < 1 || a.nsrc > 2 || !(epi == EPI_ACT || epi == EPI_DACT)) return false;
  if (a.act != ACT_NONE && a.act != ACT_RELU && a.act != ACT_SOFTPLUS) return false;
  q.x0 = a.src[0].x; q.wp0 = a.src[0].wp; q.ld0 = a.src[0].ld; q.K0 = a.src[0].K;
  q.x1 = a.nsrc > 1 ? a.src[1].x : a.src[0].x; q.wp1 = a.nsrc > 1 ? a.src[1].wp : a.src[0].wp; q.ld1 = a.nsrc > 1 ? a.src[1].ld : 0; q.K1 = a.nsrc > 1 ? a.src[1].K : 0;
  q.nsrc = a.nsrc; q.Nout = a.Nout; q.act = a.act; q.epi = epi;
  q.bias = a.bias; q.rowbias = a.rowbias; q.rowbias_ld = a.rowbias_ld; q.rows_per_group = a.rows_per_group; q.rowscale = a.rowscale; q.rowscale_w = a.rowscale_w;
  q.S = a.S; q.ldS = a.ldS; q.Q = a.Q; q.ldQ = a.ldQ; q.R = a.R; q.ldR = a.ldR; q.Y = a.Y; q.ldY = a.ldY; q.Y2 = a.Y2; q.ldY2 = a.ldY2;
  return true;
}
}  // namespace

// A chain of per-image levels (each one or two independent problems on the same M rows; level l + 1 reads outputs of levels <= l of
// its own rows only) in one launch.  counters: LINEAR_SMALL_CHAIN_COUNTER_WORDS x ceil(M / 16) unsigned ints of scratch (cleared here with a fill launch).  Falls back to
// one launch per problem when a problem does not qualify for the split-K kernel.
int launch_linear_small_chain(const LinArgs* probs, const int* epis, const int* level_of, int nprob, float* counters, hipStream_t st) {
  ARDAE_CHECK_ARG(probs && epis && level_of && nprob >= 1 && counters, "linear_small_chain: bad arguments");
  static const bool on = !(debug_knob("ARDAE_SMALL_CHAIN") && atoi(debug_knob("ARDAE_SMALL_CHAIN")) == 0);
  ScArgs c;
  memset(&c, 0, sizeof(c));
  static_assert(sizeof(ScArgs) <= 4096, "kernel arguments");
  bool ok = on && nprob <= SC_MAXPROB;
  const int M = probs[0].M;
  int nlev = 0;
  bool all16 = small_fast_on();
  for (int i = 0; all16 && i < nprob; ++i) all16 = small_regular16(probs[i]) || (probs[i].Nout <= 32 && probs[i].Nout % 16 == 0 && probs[i].M % 16 == 0 && probs[i].src[0].K % 16 == 0 && probs[i].nsrc == 1);
  const int blk = all16 ? 16 : 32;
  for (int i = 0; ok && i < nprob; ++i) {
    const int l = level_of[i];
    ok = probs[i].M == M && l >= 0 && l < SC_MAXLEV && (i == 0 ? l == 0 : (l == level_of[i - 1] || l == level_of[i - 1] + 1)) &&
         (all16 || small_regular(probs[i])) && sc_fill(probs[i], epis[i], c.pr[i]);
    if (!ok) break;
    if (l + 1 > nlev) { nlev = l + 1; c.first[l] = i; c.lv[l].np = 0; }
    ok = c.lv[l].np < 2;
    if (!ok) break;
    const int nb = ceil_div(probs[i].Nout, blk);
    if (c.lv[l].np == 0) c.lv[l].nblk0 = nb;
    c.lv[l].nblk += nb;
    c.lv[l].np += 1;
    if (c.lv[l].nblk > c.nj) c.nj = c.lv[l].nblk;
  }
  const int nrb = ceil_div(M, blk);
  // the whole grid resident at once (deadlock freedom), nothing written twice or over an input (XCD-local hand-over)
  ok = ok && nlev >= 2 && (int64_t)ceil_div(nrb, 8) * 8 * c.nj <= sc_resident_cap() && sc_write_once(probs, level_of, nprob);
  if (!ok) {
    for (int i = 0; i < nprob; ++i) ARDAE_TRY(launch_linear(probs[i], epis[i], st));
    return 0;
  }
  c.nlev = nlev; c.M = M; c.cnt = reinterpret_cast<unsigned*>(counters);
  static const bool fast = !(debug_knob("ARDAE_SMALL_CHAIN_FAST") && atoi(debug_knob("ARDAE_SMALL_CHAIN_FAST")) == 0);
  static const int dbg = debug_knob("ARDAE_SC_DEBUG") ? atoi(debug_knob("ARDAE_SC_DEBUG")) : 0;     // 2: no compute, 4: no hand-over, 8: cycle stamps of one block (timing only)
  c.fast = (fast ? 1 : 0) | (dbg & 30);
  c.nrb = nrb;
  c.blk = blk;
  ARDAE_TRY(launch_fill(counters, (size_t)SC_CNT_STRIDE * nrb, 0.f, st));
  if (g_prof_enabled) {
    double fl = 0, by = 0;
    for (int i = 0; i < nprob; ++i) {
      double ksum = 0;
      for (int s = 0; s < probs[i].nsrc; ++s) ksum += probs[i].src[s].K;
      fl += 2.0 * M * (double)probs[i].Nout * ksum;
      by += 4.0 * ((double)M * ksum + 2.0 * M * (double)probs[i].Nout + ksum * probs[i].Nout);
    }
    char name[64];
    snprintf(name, sizeof(name), "linear_small_chain_kernel x%d", nlev);
    prof_begin(st, name, fl, by);
  }
  const int groups = ceil_div(nrb, 8);
  hipLaunchKernelGGL(linear_small_chain_kernel, dim3(groups * 8 * c.nj), dim3(256), 0, st, c);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

// two independent per-image problems: one launch when both qualify for the split-K kernel with the same epilogue / activation
// (instantiated for what the score pass needs: forward layers), two launches of launch_linear otherwise
int launch_linear_pair(const LinArgs& a0, const LinArgs& a1, int epi, hipStream_t st) {
  if (epi == EPI_ACT && a0.act == a1.act && linear_small_eligible(a0, epi) && linear_small_eligible(a1, epi) && !a0.Y2 && !a1.Y2) {
    if (a0.act == ACT_SOFTPLUS) return launch_small_pair<EPI_ACT, ACT_SOFTPLUS>(a0, a1, st);
    if (a0.act == ACT_RELU) return launch_small_pair<EPI_ACT, ACT_RELU>(a0, a1, st);
  }
  ARDAE_TRY(launch_linear(a0, epi, st));
  return launch_linear(a1, epi, st);
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
      if (a.act == ACT_SWISH) return launch_small<EPI_ACT, ACT_SWISH>(a, st);
      break;
    case EPI_DACT:
      if (a.act == ACT_NONE) return launch_small<EPI_DACT, ACT_NONE>(a, st);
      if (a.act == ACT_RELU) return launch_small<EPI_DACT, ACT_RELU>(a, st);
      if (a.act == ACT_SOFTPLUS) return launch_small<EPI_DACT, ACT_SOFTPLUS>(a, st);
      if (a.act == ACT_ELU) return launch_small<EPI_DACT, ACT_ELU>(a, st);
      if (a.act == ACT_TANH) return launch_small<EPI_DACT, ACT_TANH>(a, st);
      if (a.act == ACT_LEAKY) return launch_small<EPI_DACT, ACT_LEAKY>(a, st);
      if (a.act == ACT_SWISH) return launch_small<EPI_DACT, ACT_SWISH>(a, st);
      break;
    case EPI_CHAIN:
      if (a.act == ACT_SOFTPLUS) return launch_small<EPI_CHAIN, ACT_SOFTPLUS>(a, st);
      if (a.act == ACT_RELU) return launch_small<EPI_CHAIN, ACT_RELU>(a, st);
      if (a.act == ACT_ELU) return launch_small<EPI_CHAIN, ACT_ELU>(a, st);
      if (a.act == ACT_TANH) return launch_small<EPI_CHAIN, ACT_TANH>(a, st);
      if (a.act == ACT_LEAKY) return launch_small<EPI_CHAIN, ACT_LEAKY>(a, st);
      if (a.act == ACT_SWISH) return launch_small<EPI_CHAIN, ACT_SWISH>(a, st);
      break;
  }
  ARDAE_CHECK_ARG(false, "linear_small: unsupported epilogue/activation combination (epi=%d act=%d)", epi, a.act);
  return -1;
}

}  // namespace ardae
