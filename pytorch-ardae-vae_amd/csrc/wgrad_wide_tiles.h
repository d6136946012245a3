// Tile list of a batched weight-gradient launch: shared by wgrad_wide.hip (FP32 MFMA, 256 x 32 geometry) and wgrad_x9.hip (256 x 256 tiles,
// products on the BF16 matrix cores).
#pragma once
#include "wgrad.h"

namespace ardae {

constexpr int WW_MAX_TILES = 32;

struct WwTile {
  const float* G[2];
  const float* X[2];
  const float* rowscale;   // sigma per row (pair `bias_pair` only) or null
  float* partial;          // [splits][O][I]
  float* partial_vec;      // [splits][2][O] or null
  int ldG[2], ldX[2];
  int M, npairs, O, I, o0, i0, bias_pair, want_vec;
};

struct WwBatchDev {
  int ntiles, splits;
  WwTile t[WW_MAX_TILES];
};
static_assert(sizeof(WwBatchDev) <= 4000, "kernel argument block too large");

// wgrad_x9.hip
bool wgrad_x9_eligible(const WgradProblem& p);
int launch_wgrad_x9(const WwBatchDev& b, hipStream_t st);

}  // namespace ardae
