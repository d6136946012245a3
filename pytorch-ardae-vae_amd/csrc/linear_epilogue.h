// Row-major-accumulator epilogues shared by the fused linear kernels (linear.hip, linear_wide.hip).
#pragma once
#include "linear.h"

namespace ardae {

// Epilogue of one 32x32 accumulator block (16 registers per lane).  All operand loads of the block are issued
// first, into registers, and only then the math + stores run: outputs may alias inputs (Y == Q in place), so the
// compiler cannot hoist loads over stores by itself and an element-at-a-time epilogue serialises on HBM latency.
template <int EPI, int ACT, bool FULL, int R0>
__device__ __forceinline__ void epilogue_half(const LinArgs& a, const f32x16& acc16, int rbase, int col, bool cok, float bcol,
                                              float wsig, float& csum, float& loss_part) {
  // registers R0..R0+7 of the 32x32 block: rows rbase + {0..3} + 8*(R0/4 + {0,1})
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = acc16[R0 + r];
  int rowv[8];
  bool ok[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int raw = rbase + ((R0 + r) & 3) + 8 * ((R0 + r) >> 2);
    ok[r] = FULL ? true : (cok && raw < a.M);
    rowv[r] = FULL ? raw : min(raw, a.M - 1);
  }
  float y[8];
  if (EPI == EPI_ACT) {
    float rb[8], rs[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) rb[r] = 0.f, rs[r] = 0.f;
    if (a.rowbias) {
      if (a.rows_per_group >= 32) {
        // a 32-row block meets at most two images: ONE division per block, then a compare per row (an integer division per
        // element - ~35 vector-ALU instructions each - cost the nz_cdae 625 recipes a quarter of this kernel's time)
        const int blk0 = rbase & ~31;
        const int g0 = blk0 / a.rows_per_group;
        const int split = (g0 + 1) * a.rows_per_group;                       // first row of the next image
        const float rbA = a.rowbias[(size_t)g0 * a.rowbias_ld + col];
        const float rbB = (split < blk0 + 32 && split < a.M) ? a.rowbias[(size_t)(g0 + 1) * a.rowbias_ld + col] : rbA;
#pragma unroll
        for (int r = 0; r < 8; ++r) rb[r] = rowv[r] < split ? rbA : rbB;
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) rb[r] = a.rowbias[(size_t)(rowv[r] / a.rows_per_group) * a.rowbias_ld + col];
      }
    }
    if (a.rowscale) {
#pragma unroll
      for (int r = 0; r < 8; ++r) rs[r] = a.rowscale[rowv[r]];
    }
    const float wv = a.Y2 ? a.R[col] : 0.f;   // R is the [Nout] vector w here
#pragma unroll
    for (int r = 0; r < 8; ++r) y[r] = act_fwd<ACT>(acc[r] + bcol + rb[r] + rs[r] * wsig);
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (ok[r]) a.Y[(size_t)rowv[r] * a.ldY + col] = y[r];
    if (a.Y2) {   // seed of the score pass: e_L = -w (.) act'(pre)
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (ok[r]) a.Y2[(size_t)rowv[r] * a.ldY2 + col] = -wv * act_d1<ACT>(y[r]);
    }
  } else if (EPI == EPI_DACT) {
    float sv[8], qv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) sv[r] = a.S[(size_t)rowv[r] * a.ldS + col];
    if (a.Q) {
#pragma unroll
      for (int r = 0; r < 8; ++r) qv[r] = a.Q[(size_t)rowv[r] * a.ldQ + col];
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) qv[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) y[r] = acc[r] * act_d1<ACT>(sv[r]) + qv[r];
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (ok[r]) a.Y[(size_t)rowv[r] * a.ldY + col] = y[r];
  } else if (EPI == EPI_CHAIN) {
    float sv[8], rv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) sv[r] = a.S[(size_t)rowv[r] * a.ldS + col];
#pragma unroll
    for (int r = 0; r < 8; ++r) rv[r] = a.R[(size_t)rowv[r] * a.ldR + col];
    float y2[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      // s'/s from the saved output (softplus: s = 1 - exp(-a) and 1 - s = exp(-a) are both formed without cancellation)
      const float em = act_ratio<ACT>(sv[r]);
      y[r] = acc[r] * act_d1<ACT>(sv[r]);
      y2[r] = acc[r] * rv[r] * em;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (ok[r]) {
        a.Y[(size_t)rowv[r] * a.ldY + col] = y[r];
        a.Y2[(size_t)rowv[r] * a.ldY2 + col] = y2[r];
      }
  } else {  // EPI_DAE_LOSS
    float sg[8], ev[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) sg[r] = a.sigma[rowv[r]];
#pragma unroll
    for (int r = 0; r < 8; ++r) ev[r] = a.eps[(size_t)rowv[r] * a.ldeps + col];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      y[r] = acc[r] + bcol;
      const float rho = sg[r] * y[r] + ev[r];
      if (ok[r]) {
        if (a.Y) a.Y[(size_t)rowv[r] * a.ldY + col] = y[r];
        if (a.Y2) a.Y2[(size_t)rowv[r] * a.ldY2 + col] = 2.f * sg[r] * rho * a.scale;
        loss_part += rho * rho;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) csum += ok[r] ? y[r] : 0.f;
}

template <int EPI, int ACT, bool FULL>
__device__ __forceinline__ void epilogue_block(const LinArgs& a, const f32x16& acc, int rbase, int col, bool cok, float bcol,
                                               float wsig, float& csum, float& loss_part) {
  epilogue_half<EPI, ACT, FULL, 0>(a, acc, rbase, col, cok, bcol, wsig, csum, loss_part);
  epilogue_half<EPI, ACT, FULL, 8>(a, acc, rbase, col, cok, bcol, wsig, csum, loss_part);
}

}  // namespace ardae
