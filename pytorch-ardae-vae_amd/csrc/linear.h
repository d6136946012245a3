// Fused "linear" kernel family:  Y = epilogue( sum_s X_s[M,K_s] * Wp_s^T )  on FP32 MFMA (gfx950).
//
// Replaces the reference's `MLP.forward` Linear->act chain (models/layers.py:501-515) and, with the
// derivative epilogues, the autograd passes PyTorch builds for it (first backward, the
// create_graph=True score pass of models/graddae/mlp.py:35-36,437 and its double backward).
#pragma once
#include "ardae_hip.h"
#include "common.h"

namespace ardae {

using LinSrc = ardae_lin_src;
using LinArgs = ardae_linear_args;   // value-initialise: `LinArgs a{};`

enum Epi : int {
  EPI_ACT = ARDAE_EPI_ACT,
  EPI_DACT = ARDAE_EPI_DACT,
  EPI_CHAIN = ARDAE_EPI_CHAIN,
  EPI_DAE_LOSS = ARDAE_EPI_DAE_LOSS
};

// number of floats of the packed image of an [nout, k] matrix
size_t packed_floats(int nout, int k);
// rows per row-tile of the geometry launch_linear() picks for this Nout (colsum / tile_loss sizing)
int linear_row_tile(int M, int nout);
int linear_row_tiles(int M, int nout);
int linear_col_panels(int M, int nout);

// M[n][k] = transpose ? W[k*ldw + n] : W[n*ldw + k]
int launch_pack_weight(const float* W, int ldw, int nout, int k, bool transpose, float* out, hipStream_t st);
int launch_linear(const LinArgs& a, int epi, hipStream_t st);
// pack many matrices with one launch (the per-step refresh of a network's weight images)
struct PackItem { const float* W; int ldw, nout, k, transpose; float* out; };
constexpr int PACK_BATCH_MAX = 48;
int launch_pack_batch(const PackItem* items, int n, hipStream_t st);
// split-K 32 x 32 kernel for the per-image (B-row) problems (linear_small.hip): latency, not throughput
bool linear_small_eligible(const LinArgs& a, int epi);
int launch_linear_small(const LinArgs& a, int epi, hipStream_t st);
// two INDEPENDENT per-image problems of one epilogue kind in one launch where both run on the split-K kernel (else two launches)
int launch_linear_pair(const LinArgs& a0, const LinArgs& a1, int epi, hipStream_t st);
// a dependent CHAIN of per-image levels (one or two independent problems each, probs[i] on level level_of[i], levels ascending) in
// ONE launch with a row-block counter hand-over between the levels (linear_small.hip); counters: 3 ceil(M / 32) floats of scratch
// a run of row-local K = Nout = 256 layers as one layer-major launch of the weight-stationary kernel (linear_wide.hip)
bool linear_wide_layers_eligible(const LinArgs* L, int nl, int epi);
int launch_linear_wide_layers(const LinArgs* L, int nl, int epi, hipStream_t st);
constexpr int LINEAR_SMALL_CHAIN_COUNTER_WORDS = 32;     // scratch words per 16-row block (one 128-byte line): counters = 32 x ceil(M / 16) words
int launch_linear_small_chain(const LinArgs* probs, const int* epis, const int* level_of, int nprob, float* counters, hipStream_t st);
// streaming kernel for N-row layers with Nout <= 32 and K = 256 (linear_narrow.hip): HBM-bound
bool linear_narrow_eligible(const LinArgs& a, int epi);
int launch_linear_narrow(const LinArgs& a, int epi, hipStream_t st);
// N-row forward layers with K <= 128 that is not a multiple of 32 (linear_shortk.hip): A fragments stay in registers
bool linear_shortk_eligible(const LinArgs& a, int epi);
int launch_linear_shortk(const LinArgs& a, int epi, hipStream_t st);
// that layer fused with the latent-space layer behind it (N2 <= 32 columns): the hidden rows never reach memory
bool sampler_tail_eligible(const LinArgs& first, int n2);
int launch_sampler_tail(const LinArgs& first, const float* wp2, const float* bias2, float* Z, int ldz, int n2, hipStream_t st,
                        const float* wp2b = nullptr, const float* bias2b = nullptr, float* Zb = nullptr, int ldzb = 0);   // second head (mean | logvar)
// software-pipelined 64 x 256 kernel (linear_wide.hip): full tiles, K % 32 == 0
bool linear_wide_eligible(const LinArgs& a, int epi);
int launch_linear_wide(const LinArgs& a, int epi, hipStream_t st);

// row-local CHAINS of K = Nout = 256 layers in one launch (linear_chain.hip): the small-shard regime (few tiles per workgroup)
bool linear_chain_eligible(const LinArgs* layers, int nl, int epi);
int launch_linear_chain(const LinArgs* layers, int nl, int epi, hipStream_t st);

}  // namespace ardae
