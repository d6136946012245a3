// Batched weight-gradient kernel:  dW[o][i] = sum_pairs sum_m G[m][o] * X[m][i]   (FP32 MFMA, gfx950)
//
// Replaces the `grad_weight = grad_output^T @ input` products autograd runs for every nn.Linear during
// `cdae_loss.backward()` / `model_loss.backward()` (reference ivae_ardae.py:771,804,834).  All problems of
// one update are deferred and executed by ONE launch (grid = sum over problems of o_tiles*i_tiles*splits),
// each workgroup reducing its row range in registers and writing one partial tile; a second, tiny launch
// sums the partials in a fixed order (bitwise reproducible - no atomics).
#pragma once
#include "ardae_hip.h"
#include "common.h"

namespace ardae {

using WgradProblem = ardae_wgrad_problem;
constexpr int WGRAD_MAX_PROBLEMS = ARDAE_WGRAD_MAX_PROBLEMS;

// Number of row splits the launcher will use for a problem (workspace sizing: partial needs
// splits*O*I floats, bias/rowscale partials splits*O floats each).
int wgrad_splits(int M, int O, int I, int nproblems_hint);

// runs every problem and reduces into p[i].out / out_bias / out_rowscale ( = beta*out + sum )
int launch_wgrad_batch(const WgradProblem* probs, int nprob, hipStream_t st);

// software-pipelined 256 x 256 kernel for the big, regular problems (wgrad_wide.hip)
bool wgrad_wide_eligible(const WgradProblem& p);
int wgrad_wide_geometry(const WgradProblem& p);   // 0 none, 1: 256 x 256 tiles, 2: 256 x 32 tiles
int launch_wgrad_wide(WgradProblem* probs, const int* idx, int n, hipStream_t st);

}  // namespace ardae
