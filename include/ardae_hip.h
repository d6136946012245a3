/* ardae_hip.h  --  C ABI of libardae_hip.so: the MI355X (gfx950) engine for the AR-DAE-VAE inner training loop.
 *
 * The reference (lim0606/pytorch-ardae-vae) has no FFI of its own: its hot path is PyTorch autograd over
 * Python modules.  This header is the boundary a maintainer binds instead (ctypes stub: INTEGRATION.md);
 * each entry point names the reference code (file:line under the reference root) it replaces.
 *
 * Conventions (all entry points):
 *   - plain pointers are DEVICE pointers to fp32 unless the name says `host_`; the caller owns all memory,
 *     nothing is retained past the call, nothing is allocated or freed by the library;
 *   - `stream` is a hipStream_t (pass PyTorch's current stream); calls are asynchronous and stream-ordered,
 *     no hidden device synchronisation;
 *   - return value: 0 ok, <0 invalid argument, >0 a hipError_t; `ardae_last_error()` has the message;
 *   - matrices are row-major; `nn.Linear` weights are [out, in] exactly as the reference stores them.
 */
#ifndef ARDAE_HIP_H
#define ARDAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARDAE_ABI_VERSION 1

/* activations: reference utils/models.py:14-32 (F.relu, F.softplus beta=1 threshold=20) */
enum { ARDAE_ACT_NONE = 0, ARDAE_ACT_RELU = 1, ARDAE_ACT_SOFTPLUS = 2 };

/* epilogues of ardae_linear */
enum {
  ARDAE_EPI_ACT = 0,      /* Y = act(V + bias[c] + rowbias[r / rows_per_group][c] + rowscale[r] * rowscale_w[c])        */
  ARDAE_EPI_DACT = 1,     /* Y = V * act'(S) (+ Q)          : one back-prop step through Linear->act                    */
  ARDAE_EPI_CHAIN = 2,    /* Y = V * act'(S); Y2 = V * R * (1 - act'(S)) : forward-mode step of the double backward    */
  ARDAE_EPI_DAE_LOSS = 3  /* g = V + bias; rho = sigma[r]*g + eps; Y = g; Y2 = 2*sigma*rho*scale; tile_loss += rho^2   */
};

typedef struct ardae_lin_src {
  const float* x;  /* activations [M, K], row stride ld                                   */
  int ld;
  int K;
  const float* wp; /* weight matrix [Nout, K] in the packed image written by ardae_pack_weight */
} ardae_lin_src;

typedef struct ardae_linear_args {
  int M, Nout;
  int nsrc;              /* 1 or 2 operand pairs accumulated into the same output (concat inputs)       */
  ardae_lin_src src[2];
  int act;
  const float* bias;     /* [Nout] or NULL                                                               */
  const float* rowbias;  /* [M / rows_per_group, rowbias_ld] per-image term or NULL                      */
  int rowbias_ld;
  int rows_per_group;
  const float* rowscale;   /* [M] (sigma) or NULL                                                        */
  const float* rowscale_w; /* [Nout]                                                                     */
  const float* S; int ldS; /* saved post-activation                                                      */
  const float* R; int ldR;
  const float* Q; int ldQ;
  const float* sigma;      /* [M]   (EPI_DAE_LOSS)                                                       */
  const float* eps; int ldeps;
  float scale;
  float* Y;  int ldY;
  float* Y2; int ldY2;
  float* colsum;     /* optional [ardae_linear_row_tiles(M,Nout), Nout] per-tile column sums of Y       */
  float* tile_loss;  /* [row_tiles * col_panels] partial sums (EPI_DAE_LOSS)                             */
} ardae_linear_args;

const char* ardae_last_error(void);
int ardae_abi_version(void);

/* ---- K1: fused Linear(+concat)(+act / derivative epilogue) on FP32 MFMA -----------------------------------
 * replaces models/layers.py:501-515 (MLP.forward) and the autograd passes over it.                          */
size_t ardae_packed_floats(int nout, int k);
int ardae_linear_row_tiles(int M, int nout);
int ardae_linear_col_panels(int nout);
/* M[n][k] = transpose ? W[k*ldw + n] : W[n*ldw + k]  ->  MFMA-lane-linear image (out: ardae_packed_floats) */
int ardae_pack_weight(const float* W, int ldw, int nout, int k, int transpose, float* out, void* stream);
int ardae_linear(const ardae_linear_args* args, int epilogue, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ARDAE_HIP_H */
