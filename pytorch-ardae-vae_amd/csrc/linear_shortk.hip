// N-row layers with a SHORT reduction: Y[M, Nout] = act(X[M, K <= 128] . Wp^T + bias + per-image row bias) - the first
// sampler layer on the B*nz Monte-Carlo rows, whose only N-row input is the noise (K = noise_dim = 100; the image part of
// the concat enters as the per-image row bias: ivae/mnist.py:123-165, models/layers.py:681-724).  7 GFLOP, 52 MB in,
// 134 MB out at config #2.  K = 100 fits none of the panelled kernels (64-wide LDS panels: 64 + a ragged 36), and two
// panels per tile amortise nothing; so here a wave keeps the whole K extent of its 32 rows in registers as MFMA A-fragments
// (<= 16 float4 per lane, read from HBM exactly once, zero beyond K) and walks the column blocks with the packed weight
// fragments of the next block in flight from L2 behind the MFMAs of the current one.  No LDS, no barriers.
#include <stdlib.h>

#include <algorithm>

#include "linear.h"
#include "profile.h"

namespace ardae {
namespace {

constexpr int SK_MAXK = 128;

// SK_MAXCH: compile-time bound of the chunk count (8 k each) - it sizes the three fragment arrays, i.e. the occupancy
template <int ACT, int SK_MAXCH>
__global__ __launch_bounds__(256, (SK_MAXCH > 13 ? 1 : 2)) void linear_shortk_kernel(const LinArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int row0 = (blockIdx.x * 4 + wave) * 32;          // wave-uniform: row bases below live in SGPRs
  const int K = a.src[0].K, nch = (K + 7) >> 3;

  const float* xr = a.src[0].x + (size_t)(row0 + l31) * a.src[0].ld + 4 * hh;
  f32x4 av[SK_MAXCH];
#pragma unroll
  for (int c = 0; c < SK_MAXCH; ++c) {
    av[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nch && 8 * c + 4 * hh + 4 <= K) av[c] = *reinterpret_cast<const f32x4*>(xr + 8 * c);
  }

  const int nblk = a.Nout >> 5;
  const float* bp = a.src[0].wp + lane * 4;
  const size_t bstride = (size_t)nch * 256;          // floats per column block of the packed image
  // Groups of >= 32 rows: the 32 rows of a wave belong to at most two images - rows [0, split) to the first (split >= 32: all of
  // them), the rest to the next one: two row-bias values per column and a compare per element instead of a division per element
  // (nz_cdae 625 of the shipped recipes: the per-element form made this kernel 3x slower than at nz_cdae 256).
  const bool rb_two = a.rowbias && a.rows_per_group >= 32;
  const int g0 = a.rowbias ? row0 / a.rows_per_group : 0;
  const int split = rb_two ? (g0 + 1) * a.rows_per_group - row0 : 32;
  const float* rbrow = a.rowbias ? a.rowbias + (size_t)g0 * a.rowbias_ld : nullptr;
  const float* rbrow1 = (rb_two && split < 32) ? rbrow + a.rowbias_ld : rbrow;

  f32x4 b0[SK_MAXCH], b1[SK_MAXCH];
  auto load_b = [&](f32x4 (&b)[SK_MAXCH], int nb) {
#pragma unroll
    for (int c = 0; c < SK_MAXCH; ++c)
      if (c < nch) b[c] = *reinterpret_cast<const f32x4*>(bp + (size_t)nb * bstride + (size_t)c * 256);
  };
  auto block = [&](const f32x4 (&b)[SK_MAXCH], int nb) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < SK_MAXCH; ++c)
      if (c < nch) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][q], b[c][q], acc, 0, 0, 0);
      }
    const int col = nb * 32 + l31;
    const float bcolv = a.bias ? a.bias[col] : 0.f;
    const float pre = bcolv + (rb_two ? rbrow[col] : 0.f), pre1 = bcolv + (rb_two ? rbrow1[col] : 0.f);
    // stores in the scalar-base form: the row base is uniform (SALU arithmetic), the lane adds one 32-bit byte offset
    const unsigned voff = (unsigned)((4 * hh * a.ldY + l31) * 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rowu = row0 + (r & 3) + 8 * (r >> 2);       // + 4 hh is in voff
      float v = acc[r] + (((r & 3) + 8 * (r >> 2) + 4 * hh) < split ? pre : pre1);
      if (a.rowbias && !rb_two) v += a.rowbias[(size_t)((rowu + 4 * hh) / a.rows_per_group) * a.rowbias_ld + col];
      char* sb = reinterpret_cast<char*>(a.Y + (size_t)rowu * a.ldY + nb * 32);
      *reinterpret_cast<float*>(sb + voff) = act_fwd<ACT>(v);
    }
  };

  load_b(b0, 0);
  for (int nb = 0; nb < nblk; nb += 2) {
    if (nb + 1 < nblk) load_b(b1, nb + 1);
    block(b0, nb);
    if (nb + 1 >= nblk) break;
    if (nb + 2 < nblk) load_b(b0, nb + 2);
    block(b1, nb + 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same layer FUSED with the latent-space layer behind it:  Z[M, N2 <= 32] = act(X W1^T + rowbias (+ b1)) W2^T + b2 - the whole
// N-row sampler of the mnist-concat model when nobody needs its hidden rows (encode() / forward_hidden() of the cDAE phase run under
// no_grad: ivae_ardae.py:734-751).  The hidden block of 32 columns a wave has just finished goes through a wave-private LDS tile
// (accumulator layout -> A-fragment layout) straight into 16 more MFMAs against the matching K slice of W2; the [M, h] hidden tensor
// (134 MB written + read at config #2) never exists.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ST_LD = 36;   // floats per row of the transpose tile (16-byte aligned rows, 4-bank skew)

// NOUT2 = 2: two heads of <= 32 columns each behind the same hidden layer (mean | logvar of the hierarchical conv sampler,
// ivae/auxconv.py) - the second set of accumulators and weight fragments takes the kernel to one workgroup per CU.
struct TailOut { const float* wp; const float* bias; float* Z; int ldz; };

template <int ACT, int SK_MAXCH, int NOUT2>
__global__ __launch_bounds__(256, ((SK_MAXCH > 13 || NOUT2 > 1) ? 1 : 2)) void sampler_tail_kernel(const LinArgs a, const TailOut o0, const TailOut o1, int n2, int cs) {
  // cs (1, 2, 4): few rows - the hidden layer's column blocks of a 32-row block are split over cs waves (each a chain of 544 / cs MFMAs
  // instead of 544), whose partial latent-space sums meet in LDS; the workgroup then holds 4 / cs row blocks.
  // The latent-space sum over the hidden columns has ONE order for every cs (round 4: a sample's z must not depend on how many rows the
  // launch has, i.e. on the images per rank): the column blocks form FOUR groups, group g = blocks [g nblk / 4, (g + 1) nblk / 4), each
  // summed block by block from zero, and z = ((p0 + p1) + p2) + p3 + bias - in one wave's registers (cs = 1), or with the groups of
  // the other waves handed over through LDS (cs = 2: wave 1 hands p2 and p3 over separately; cs = 4: one group per wave).
  __shared__ float tile[4][32 * ST_LD];
  __shared__ float part[4][NOUT2][2][32 * ST_LD];      // a wave's finished groups for the wave that adds them up (cs > 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int cpart = wave & (cs - 1), rbi = wave / cs;
  const int row0 = (blockIdx.x * (4 / cs) + rbi) * 32;
  const int K = a.src[0].K, nch = (K + 7) >> 3;
  const int h = a.Nout, nblk = h >> 5, nch2 = h >> 3;

  const float* xr = a.src[0].x + (size_t)(row0 + l31) * a.src[0].ld + 4 * hh;
  f32x4 av[SK_MAXCH];
#pragma unroll
  for (int c = 0; c < SK_MAXCH; ++c) {
    av[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nch && 8 * c + 4 * hh + 4 <= K) av[c] = *reinterpret_cast<const f32x4*>(xr + 8 * c);
  }
  const float* bp = a.src[0].wp + lane * 4;
  const float* bp2[2] = {o0.wp + lane * 4, (NOUT2 > 1 ? o1.wp : o0.wp) + lane * 4};
  const size_t bstride = (size_t)nch * 256;
  const bool rb_two = a.rowbias && a.rows_per_group >= 32;      // see linear_shortk_kernel: at most two images per wave
  const int g0 = a.rowbias ? row0 / a.rows_per_group : 0;
  const int split = rb_two ? (g0 + 1) * a.rows_per_group - row0 : 32;
  const float* rbrow = a.rowbias ? a.rowbias + (size_t)g0 * a.rowbias_ld : nullptr;
  const float* rbrow1 = (rb_two && split < 32) ? rbrow + a.rowbias_ld : rbrow;
  float* T = tile[wave];

  f32x16 zacc[NOUT2];                    // the current group's sum (the sum of the wave's finished groups waits in LDS: part[wave][o][0])
#pragma unroll
  for (int o = 0; o < NOUT2; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) zacc[o][r] = 0.f;
  const bool adder = cs == 1 || cpart == 0;
  const int gpw = 4 / cs;                // groups per wave
  int grp = cpart * gpw;                 // the group the wave is summing; its last block is ((grp + 1) nblk) / 4 - 1
  f32x4 b0[SK_MAXCH], b1[SK_MAXCH];
  auto load_b = [&](f32x4 (&b)[SK_MAXCH], int nb) {
#pragma unroll
    for (int c = 0; c < SK_MAXCH; ++c)
      if (c < nch) b[c] = *reinterpret_cast<const f32x4*>(bp + (size_t)nb * bstride + (size_t)c * 256);
  };
  auto block = [&](const f32x4 (&b)[SK_MAXCH], int nb) {
    f32x4 w2[NOUT2][4];
#pragma unroll
    for (int o = 0; o < NOUT2; ++o)
#pragma unroll
      for (int c2 = 0; c2 < 4; ++c2) w2[o][c2] = *reinterpret_cast<const f32x4*>(bp2[o] + (size_t)(4 * nb + c2) * 256);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < SK_MAXCH; ++c)
      if (c < nch) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][q], b[c][q], acc, 0, 0, 0);
      }
    const int col = nb * 32 + l31;
    const float bcolv = a.bias ? a.bias[col] : 0.f;
    const float pre = bcolv + (rb_two ? rbrow[col] : 0.f), pre1 = bcolv + (rb_two ? rbrow1[col] : 0.f);
    // hidden block: accumulator layout (lane = column, registers = rows) -> tile[row][column]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh;
      float v = acc[r] + (rl < split ? pre : pre1);
      if (a.rowbias && !rb_two) v += a.rowbias[(size_t)((row0 + rl) / a.rows_per_group) * a.rowbias_ld + col];
      T[rl * ST_LD + l31] = act_fwd<ACT>(v);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ... and back as A fragments (lane = row, four consecutive k): 16 MFMAs against W2[:, 32 nb .. 32 nb + 31]
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
      const f32x4 ha = *reinterpret_cast<const f32x4*>(T + l31 * ST_LD + 8 * c2 + 4 * hh);
#pragma unroll
      for (int o = 0; o < NOUT2; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) zacc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(ha[q], w2[o][c2][q], zacc[o], 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();     // the tile is rewritten by the next block
    // group boundaries (wave-uniform): fold the finished group(s) - empty groups (nblk < 4) add an exact zero
    while (grp < (cpart + 1) * gpw && nb + 1 == ((grp + 1) * nblk) / 4) {
      const int gl = grp - cpart * gpw;            // index of the group inside the wave
#pragma unroll
      for (int o = 0; o < NOUT2; ++o) {
        // the adding wave keeps (p0 + p1 ..) in its slot 0; the others hand every group over as it is
        float* P = part[wave][o][adder ? 0 : gl];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int e = ((r & 3) + 8 * (r >> 2) + 4 * hh) * ST_LD + l31;
          float v = zacc[o][r];
          if (adder && gl > 0) v = P[e] + v;
          P[e] = v;
          zacc[o][r] = 0.f;
        }
      }
      ++grp;
    }
  };
  (void)nch2;
  const int nb_lo = (cpart * gpw * nblk) / 4, nb_hi = ((cpart + 1) * gpw * nblk) / 4;
  // groups that end before the wave's first block (empty leading groups when nblk < 4): nothing to add, move on
  while (grp < (cpart + 1) * gpw && ((grp + 1) * nblk) / 4 <= nb_lo) {
#pragma unroll
    for (int o = 0; o < NOUT2; ++o)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[wave][o][adder ? 0 : grp - cpart * gpw][((r & 3) + 8 * (r >> 2) + 4 * hh) * ST_LD + l31] = 0.f;
    ++grp;
  }
  load_b(b0, nb_lo);
  for (int nb = nb_lo; nb < nb_hi; nb += 2) {
    if (nb + 1 < nb_hi) load_b(b1, nb + 1);
    block(b0, nb);
    if (nb + 1 >= nb_hi) break;
    if (nb + 2 < nb_hi) load_b(b0, nb + 2);
    block(b1, nb + 1);
  }
  if (cs > 1) __syncthreads();
  if (adder) {       // its own groups' sum, then (cs > 1) the groups of waves (rbi, 1 .. cs - 1) in group order
#pragma unroll
    for (int o = 0; o < NOUT2; ++o) {
#pragma unroll
      for (int r = 0; r < 16; ++r) zacc[o][r] = part[wave][o][0][((r & 3) + 8 * (r >> 2) + 4 * hh) * ST_LD + l31];
      for (int c = 1; c < cs; ++c)
        for (int gl = 0; gl < gpw; ++gl) {
          const float* P = part[wave + c][o][gl];
#pragma unroll
          for (int r = 0; r < 16; ++r) zacc[o][r] += P[((r & 3) + 8 * (r >> 2) + 4 * hh) * ST_LD + l31];
        }
    }
  }
  if (l31 < n2 && cpart == 0) {
#pragma unroll
    for (int o = 0; o < NOUT2; ++o) {
      const TailOut& out = o == 0 ? o0 : o1;
      const float bz = out.bias ? out.bias[l31] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowu = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        out.Z[(size_t)rowu * out.ldz + l31] = zacc[o][r] + bz;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same fused sampler tail, WEIGHT-STATIONARY, for the shape every MLP configuration of the reference has (h = 256, i.e. 8 hidden column
// blocks; K <= 104; one head): round 4.  The general kernel above re-reads the packed W1 / W2 fragments for every 32-row block (13 + 4
// float4 per lane per column block, double buffered: 104 registers of staging) and runs ONE accumulator chain of 52 + 16 dependent MFMAs
// per column block - 123 us at config #2 against ~66 us of matrix time (0.46 of the FP32 MFMA peak; rocprofv3, round 3).  Here
//   * a workgroup is persistent over 32-row blocks (row block i, i + grid, ...); wave w owns hidden column blocks 2 w and 2 w + 1 for
//     EVERY row block, and its slices of the packed weights - W1: 2 x 13 fragments, W2: 2 x 4 - are loaded once, into registers;
//   * the two column blocks' first-layer products are two INDEPENDENT accumulator chains, interleaved MFMA by MFMA (a dependent
//     32 x 32 x 2 MFMA issues ~92 cycles after its predecessor, an independent one after 64: scratch/mfma/dep.hip);
//   * the next row block's noise rows are requested before this block's products start (one whole row block of loads in flight);
//   * the four waves' partial latent-space sums meet in LDS (double buffered over row blocks: one barrier per row block) and wave w
//     finishes accumulator registers 4 w .. 4 w + 3.
// Same order of every sum as sampler_tail_kernel (wave w's two column blocks ARE its column group w; z = ((p0 + p1) + p2) + p3 + bias):
// bit-identical results - tests/test_engine_gpu.py::test_sampler_bits_do_not_depend_on_images_per_launch covers both kernels' row counts.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TW_CH = 13;      // chunks of 8 k of the first layer (K <= 104)
template <int ACT>
__global__ __launch_bounds__(256, 1) void sampler_tail_ws_kernel(const LinArgs a, const TailOut o0, int n2, int nrb) {
  __shared__ float tile[4][2][32 * ST_LD];      // wave-private: the wave's two hidden blocks of the current row block (accumulator -> A layout)
  __shared__ float part[2][4][16][64];          // [row-block parity][wave][accumulator register][lane]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int K = a.src[0].K, nch = (K + 7) >> 3;

  // ---- stationary operands
  f32x4 bw[2][TW_CH], w2[2][4];
  float bcol[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nb = 2 * wave + j;
    const float* bp = a.src[0].wp + (size_t)nb * nch * 256 + lane * 4;
#pragma unroll
    for (int c = 0; c < TW_CH; ++c) {
      bw[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nch) bw[j][c] = *reinterpret_cast<const f32x4*>(bp + (size_t)c * 256);
    }
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) w2[j][c2] = *reinterpret_cast<const f32x4*>(o0.wp + (size_t)(4 * nb + c2) * 256 + lane * 4);
    bcol[j] = a.bias ? a.bias[nb * 32 + l31] : 0.f;
  }
  const float bz = (o0.bias && l31 < n2) ? o0.bias[l31] : 0.f;
  const bool rb_two = a.rowbias != nullptr;       // (host: rows_per_group >= 32 whenever there is a row bias)

  auto load_rows = [&](int rb, f32x4 (&av)[TW_CH]) {
    const float* xr = a.src[0].x + (size_t)(rb * 32 + l31) * a.src[0].ld + 4 * hh;
#pragma unroll
    for (int c = 0; c < TW_CH; ++c) {
      av[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nch && 8 * c + 4 * hh + 4 <= K) av[c] = *reinterpret_cast<const f32x4*>(xr + 8 * c);
    }
  };

  f32x4 avA[TW_CH], avB[TW_CH];
  int rb = blockIdx.x, par = 0;
  if (rb < nrb) load_rows(rb, avA);
  // one row block: `av` holds its rows, `avn` receives the next one's (the two register sets swap roles from trip to trip: no copies)
  auto row_block = [&](const f32x4 (&av)[TW_CH], f32x4 (&avn)[TW_CH]) {
    const int row0 = rb * 32, rbn = rb + (int)gridDim.x;
    if (rbn < nrb) load_rows(rbn, avn);            // the next row block's rows: in flight behind this block's products
    // per-image row bias: the 32 rows belong to at most two images (see linear_shortk_kernel)
    const int g0 = rb_two ? row0 / a.rows_per_group : 0;
    const int split = rb_two ? (g0 + 1) * a.rows_per_group - row0 : 32;
    float pre[2], pre1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (2 * wave + j) * 32 + l31;
      const float* rbrow = rb_two ? a.rowbias + (size_t)g0 * a.rowbias_ld : nullptr;
      const float* rbrow1 = (rb_two && split < 32) ? rbrow + a.rowbias_ld : rbrow;
      pre[j] = bcol[j] + (rb_two ? rbrow[col] : 0.f);
      pre1[j] = bcol[j] + (rb_two ? rbrow1[col] : 0.f);
    }
    // ---- first layer: two independent chains
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
#pragma unroll
    for (int c = 0; c < TW_CH; ++c)
      if (c < nch) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][q], bw[0][c][q], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][q], bw[1][c][q], acc1, 0, 0, 0);
        }
      }
    // ---- hidden blocks: accumulator layout (lane = column, registers = rows) -> tile[row][column]
    float* T0 = tile[wave][0];
    float* T1 = tile[wave][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = (r & 3) + 8 * (r >> 2) + 4 * hh;
      T0[rl * ST_LD + l31] = act_fwd<ACT>(acc0[r] + (rl < split ? pre[0] : pre1[0]));
      T1[rl * ST_LD + l31] = act_fwd<ACT>(acc1[r] + (rl < split ? pre[1] : pre1[1]));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- latent-space layer: this wave's column group (block 2 w, then block 2 w + 1), one chain from zero
    f32x16 zacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) zacc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float* T = j ? T1 : T0;
#pragma unroll
      for (int c2 = 0; c2 < 4; ++c2) {
        const f32x4 ha = *reinterpret_cast<const f32x4*>(T + l31 * ST_LD + 8 * c2 + 4 * hh);
#pragma unroll
        for (int q = 0; q < 4; ++q) zacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ha[q], w2[j][c2][q], zacc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[par][wave][r][lane] = zacc[r];
    __syncthreads();        // (the tiles are wave-private; part is double buffered over row blocks: one barrier per row block)
    // wave w finishes accumulator registers 4 w .. 4 w + 3: rows 8 w + 4 hh + {0 .. 3}, column l31
    if (l31 < n2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * wave + i;
        const float v = ((part[par][0][r][lane] + part[par][1][r][lane]) + part[par][2][r][lane]) + part[par][3][r][lane];
        o0.Z[(size_t)(row0 + 8 * wave + 4 * hh + i) * o0.ldz + l31] = v + bz;
      }
    }
    rb = rbn;
    par ^= 1;
  };
  while (rb < nrb) {
    row_block(avA, avB);
    if (rb >= nrb) break;
    row_block(avB, avA);
  }
}

inline bool tail_ws_eligible(const LinArgs& a, const TailOut& o1) {
  static const bool on = !(debug_knob("ARDAE_TAIL_WS") && atoi(debug_knob("ARDAE_TAIL_WS")) == 0);
  return on && !o1.wp && a.Nout == 256 && ((a.src[0].K + 7) >> 3) <= TW_CH && (!a.rowbias || a.rows_per_group >= 32);
}

template <int ACT>
int launch_tail_ws(const LinArgs& a, const TailOut& o0, int n2, hipStream_t st) {
  if (g_prof_enabled) {
    char name[64];
    snprintf(name, sizeof(name), "sampler_tail_ws_kernel<%d>", ACT);
    const double K = a.src[0].K, nc = (double)n2;
    prof_begin(st, name, 2.0 * a.M * ((double)a.Nout * K + (double)a.Nout * nc), 4.0 * ((double)a.M * K + (double)a.M * nc + K * a.Nout + (double)a.Nout * nc));
  }
  const int nrb = a.M / 32;
  hipLaunchKernelGGL((sampler_tail_ws_kernel<ACT>), dim3(std::min(nrb, 256)), dim3(256), 0, st, a, o0, n2, nrb);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

template <int ACT, int MAXCH, int NOUT2>
int launch_tail_ch(const LinArgs& a, const TailOut& o0, const TailOut& o1, int n2, hipStream_t st) {
  if (g_prof_enabled) {
    char name[64];
    snprintf(name, sizeof(name), "sampler_tail_kernel<%d, %d, %d>", ACT, MAXCH, NOUT2);
    const double K = a.src[0].K, nc = (double)n2 * NOUT2;
    prof_begin(st, name, 2.0 * a.M * ((double)a.Nout * K + (double)a.Nout * nc), 4.0 * ((double)a.M * K + (double)a.M * nc + K * a.Nout + (double)a.Nout * nc));
  }
  // few rows: split the hidden columns over 2 / 4 waves until the grid fills the chip (ARDAE_TAIL_SPLIT=1: never)
  static const int max_cs = debug_knob("ARDAE_TAIL_SPLIT") ? atoi(debug_knob("ARDAE_TAIL_SPLIT")) : 4;
  int cs = 1;
  while (cs < max_cs && cs < 4 && (a.M / 128) * (2 * cs) <= 256 && (a.Nout / 32) % (2 * cs) == 0) cs *= 2;
  hipLaunchKernelGGL((sampler_tail_kernel<ACT, MAXCH, NOUT2>), dim3(a.M / 128 * cs), dim3(256), 0, st, a, o0, o1, n2, cs);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

template <int ACT>
int launch_tail(const LinArgs& a, const TailOut& o0, const TailOut& o1, int n2, hipStream_t st) {
  const int nch = (a.src[0].K + 7) >> 3;
  if (tail_ws_eligible(a, o1)) return launch_tail_ws<ACT>(a, o0, n2, st);
  if (o1.wp) {
    if (nch <= 13) return launch_tail_ch<ACT, 13, 2>(a, o0, o1, n2, st);
    return launch_tail_ch<ACT, 16, 2>(a, o0, o1, n2, st);
  }
  if (nch <= 8) return launch_tail_ch<ACT, 8, 1>(a, o0, o1, n2, st);
  if (nch <= 13) return launch_tail_ch<ACT, 13, 1>(a, o0, o1, n2, st);
  return launch_tail_ch<ACT, 16, 1>(a, o0, o1, n2, st);
}

template <int ACT, int MAXCH>
int launch_shortk_ch(const LinArgs& a, hipStream_t st) {
  if (g_prof_enabled) {
    char name[64];
    snprintf(name, sizeof(name), "linear_shortk_kernel<%d, %d>", ACT, MAXCH);
    const double K = a.src[0].K;
    prof_begin(st, name, 2.0 * a.M * (double)a.Nout * K, 4.0 * ((double)a.M * K + (double)a.M * a.Nout + K * a.Nout));
  }
  hipLaunchKernelGGL((linear_shortk_kernel<ACT, MAXCH>), dim3(a.M / 128), dim3(256), 0, st, a);
  prof_end(st);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

template <int ACT>
int launch_shortk(const LinArgs& a, hipStream_t st) {
  const int nch = (a.src[0].K + 7) >> 3;
  if (nch <= 8) return launch_shortk_ch<ACT, 8>(a, st);
  if (nch <= 13) return launch_shortk_ch<ACT, 13>(a, st);
  return launch_shortk_ch<ACT, 16>(a, st);
}

}  // namespace

// One source, K <= 128 and a multiple of 4 but NOT one of the panelled kernels' shapes, whole 128-row tiles and 32-column
// blocks, >= 8192 rows (64 workgroups: from there on it beats the generic kernel - at the 64-image shard of the 8-GPU run, 16384
// rows, the fused sampler is 1.7 % of the step); forward epilogue (bias, per-image row bias, activation) only.  ARDAE_SHORTK=0: off.
bool linear_shortk_eligible(const LinArgs& a, int epi) {
  static const bool on = !(debug_knob("ARDAE_SHORTK") && atoi(debug_knob("ARDAE_SHORTK")) == 0);
  if (!on || epi != EPI_ACT || a.nsrc != 1) return false;
  const int K = a.src[0].K;
  if (K <= 0 || K > SK_MAXK || (K & 3) || K % 32 == 0) return false;
  if (a.M < 128 * 64 || (a.M % 128) || a.Nout <= 0 || (a.Nout % 32)) return false;
  if ((a.src[0].ld & 3) || (reinterpret_cast<uintptr_t>(a.src[0].x) & 15)) return false;
  if (a.rowscale || a.Y2 || a.colsum || !a.Y) return false;
  if (a.rowbias && a.rows_per_group <= 0) return false;
  return a.act == ACT_NONE || a.act == ACT_RELU || a.act == ACT_SOFTPLUS;
}

// `first` is the short-K layer as launch_linear would get it (its Y is ignored), followed by Z = hidden . W2^T + b2 with N2 <= 32
// columns.  ARDAE_SAMPLER_TAIL=0: off.
bool sampler_tail_eligible(const LinArgs& first, int n2) {
  static const bool on = !(debug_knob("ARDAE_SAMPLER_TAIL") && atoi(debug_knob("ARDAE_SAMPLER_TAIL")) == 0);
  if (!on || n2 < 1 || n2 > 32) return false;
  LinArgs a = first;
  if (!a.Y) a.Y = const_cast<float*>(a.src[0].x);   // the single-layer rule wants an output pointer; unused here
  return linear_shortk_eligible(a, EPI_ACT);
}

// wp2b / bias2b / Zb: an optional second head of n2 columns behind the same hidden layer (nullptr: one head)
int launch_sampler_tail(const LinArgs& first, const float* wp2, const float* bias2, float* Z, int ldz, int n2, hipStream_t st,
                        const float* wp2b, const float* bias2b, float* Zb, int ldzb) {
  ARDAE_CHECK_ARG(sampler_tail_eligible(first, n2) && wp2 && Z && ldz >= n2 && (!wp2b || (Zb && ldzb >= n2)), "sampler tail: shape not eligible");
  const TailOut o0{wp2, bias2, Z, ldz}, o1{wp2b, bias2b, Zb, ldzb};
  if (first.act == ACT_RELU) return launch_tail<ACT_RELU>(first, o0, o1, n2, st);
  if (first.act == ACT_SOFTPLUS) return launch_tail<ACT_SOFTPLUS>(first, o0, o1, n2, st);
  return launch_tail<ACT_NONE>(first, o0, o1, n2, st);
}

int launch_linear_shortk(const LinArgs& a, int epi, hipStream_t st) {
  (void)epi;
  if (a.act == ACT_RELU) return launch_shortk<ACT_RELU>(a, st);
  if (a.act == ACT_SOFTPLUS) return launch_shortk<ACT_SOFTPLUS>(a, st);
  return launch_shortk<ACT_NONE>(a, st);
}

}  // namespace ardae
