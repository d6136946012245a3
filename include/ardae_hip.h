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
/* get_nonlinear_func (reference utils/models.py:14-32): relu, softplus ('csoftplus' = log(exp(x) + 1) is the same function; it is evaluated in softplus' overflow- and cancellation-free form), elu (alpha 1), tanh, leaky_relu
 * (slope 0.2), swish (x sigmoid(x), utils/models.py:8-10).  Derivatives are rebuilt from saved OUTPUTS; swish is not monotonic, so its
 * forward records the branch (x below / above the minimum at -1.27846) in the lowest mantissa bit of the stored output (<= 1 ulp) and
 * the derivative epilogues invert it by bisection (csrc/common.h::swish_f).  The software-pipelined N-row kernels exist for
 * NONE / RELU / SOFTPLUS (every shipped recipe); ELU / TANH / LEAKY / SWISH layers run on the generic kernels. */
enum { ARDAE_ACT_NONE = 0, ARDAE_ACT_RELU = 1, ARDAE_ACT_SOFTPLUS = 2, ARDAE_ACT_ELU = 3, ARDAE_ACT_TANH = 4, ARDAE_ACT_LEAKY_RELU = 5, ARDAE_ACT_SWISH = 6 };

/* epilogues of ardae_linear */
enum {
  ARDAE_EPI_ACT = 0,      /* Y = act(V + bias[c] + rowbias[r / rows_per_group][c] + rowscale[r] * rowscale_w[c]);
                             if Y2: Y2 = -R[c] * act'(Y)  (R is an [Nout] vector: seed of the score pass)               */
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
int ardae_linear_col_panels(int M, int nout);
/* M[n][k] = transpose ? W[k*ldw + n] : W[n*ldw + k]  ->  MFMA-lane-linear image (out: ardae_packed_floats) */
int ardae_pack_weight(const float* W, int ldw, int nout, int k, int transpose, float* out, void* stream);
int ardae_linear(const ardae_linear_args* args, int epilogue, void* stream);
/* A row-local CHAIN of nl layers (layer l's input = layer l - 1's Y) in ONE launch: the workgroup keeps its 64-row tile in LDS
 * between layers and reloads the next layer's weight slab behind the current layer's MFMAs - what the N-row passes of the cDAE
 * update (models/graddae/mlp.py:400-444: forward, score, forward-mode and backward runs of h x h layers) use when a rank's shard
 * is only a few tiles per CU (data-parallel runs).  Results are bit-identical to nl calls of ardae_linear.
 * ardae_linear_chain_eligible: 1 if the shapes qualify (K = Nout = 256, M % 64 == 0, few row tiles, one epilogue kind). */
int ardae_linear_chain_eligible(const ardae_linear_args* layers, int nl, int epilogue);
int ardae_linear_chain(const ardae_linear_args* layers, int nl, int epilogue, void* stream);
/* The same run LAYER-major for MANY tiles per workgroup (the full-size shard): one launch of the weight-stationary N-row kernel walks all
 * its row tiles for layer l (the wave's weight slab loaded once per layer, as in a launch of its own), then layer l + 1 on the rows it has
 * written itself - no hand-over between workgroups, no dispatch ramp / drain between the layers.  Bit-identical to nl calls of ardae_linear.
 * Eligible: K = Nout = 256, every layer reading its predecessor's Y, one epilogue kind / activation (relu, softplus) / set of operands
 * (forward layers with a bias only; DACT all with Q or all without; CHAIN). */
int ardae_linear_wide_layers_eligible(const ardae_linear_args* layers, int nl, int epilogue);
int ardae_linear_wide_layers(const ardae_linear_args* layers, int nl, int epilogue, void* stream);


/* ---- K6w: batched weight gradients  dW[o][i] = sum_pairs sum_m G[m][o] X[m][i]  -------------------------------
 * replaces the grad_weight / grad_bias products of autograd's Linear backward executed by
 * cdae_loss.backward() (ivae_ardae.py:771) and model_loss.backward() / latent.backward(grad) (:804,:834).      */
#define ARDAE_WGRAD_MAX_PROBLEMS 20
typedef struct ardae_wgrad_problem {
  int M, O, I;
  int npairs;                 /* 1 or 2 (G,X) pairs summed into the same dW                                  */
  const float* G[2]; int ldG[2]; /* [M, O] output-side factors                                               */
  const float* X[2]; int ldX[2]; /* [M, I] input-side factors                                                */
  int bias_pair;              /* pair whose G is column-summed into out_bias / out_rowscale, or -1           */
  const float* rowscale;      /* optional [M]: out_rowscale[o] = sum_m rowscale[m] * G[bias_pair][m][o]      */
  int splits;                 /* row splits, from ardae_wgrad_splits                                         */
  float* partial;             /* scratch [splits, O, I]                                                      */
  float* partial_vec;         /* scratch [splits, 2, O] (needed when bias_pair >= 0)                         */
  float* out; int ldout;      /* [O, I] view of the gradient buffer (row stride ldout)                       */
  float* out_bias;            /* [O] or NULL                                                                 */
  float* out_rowscale; int ld_rowscale; /* strided [O] (one column of an [O, *] matrix) or NULL              */
  float beta;                 /* out = beta*out + sum                                                        */
} ardae_wgrad_problem;
int ardae_wgrad_splits(int M, int O, int I, int nproblems_hint);
int ardae_wgrad_batch(const ardae_wgrad_problem* problems, int nproblems, void* stream);


/* ---- helper kernels ---------------------------------------------------------------------------------------- */
/* ivae_ardae.py:753-761 + models/graddae/mlp.py:21-23:  u = s(z - z0[b]); std_b = delta*mean_d(std_nz(u)) (unbiased);
 * sigma[b,i] = std_b*xi[b,i]; xbar = u + sigma*eps.   latent [B,nz,z], z0 [B,z], xi [B*nz], eps [B*nz,z]            */
int ardae_latent_perturb(const float* latent, const float* z0, const float* xi, const float* eps, int B, int nz, int z,
                         float std_scale, float delta, float* xbar, float* sigma, float* std_b, void* stream);
/* --train-nstd-cdae > 1 (ivae_ardae.py:759-767): every one of the nz sample rows is used nstd times, each with its own sigma and eps;
 * the statistics are those of the nz samples.  xi [B*nz*nstd], eps / xbar [B*nz*nstd, z], sigma [B*nz*nstd]; row (b, i, j) <- latent[b, i] */
int ardae_latent_perturb_nstd(const float* latent, const float* z0, const float* xi, const float* eps, int B, int nz, int nstd, int z,
                              float std_scale, float delta, float* xbar, float* sigma, float* std_b, void* stream);
/* The same perturbation with its two draws made INSIDE the kernel (ivae_ardae.py:761 `torch.randn_like(std)` and
 * models/graddae/mlp.py:21-23 `add_gaussian_noise`): no xi / eps inputs; the kernel generates the Philox counters of its image's rows,
 * keyed exactly like ardae_philox_normal_at (element i of this rank's rows = element first_row [* z] + i of the global draw with
 * (seed, offset_xi / offset_eps [+ the step state's base offset])), and writes the eps rows to eps_out for the DAE loss.  Same
 * numbers as two ardae_philox_normal_at calls followed by ardae_latent_perturb, two launches and 2 x N (z + 1) floats of HBM traffic
 * less.  ardae_latent_perturb_draw_ok: 1 if (nz, nstd, z) qualifies (nstd == 1, z a power of two, nz % 4 == 0, nz z <= 8192). */
int ardae_latent_perturb_draw_ok(int nz, int nstd, int z);
int ardae_latent_perturb_draw(const float* latent, const float* z0, int B, int nz, int z, float std_scale, float delta, uint64_t seed,
                              uint64_t offset_xi, uint64_t offset_eps, const void* state, uint64_t first_row, float* xbar, float* sigma,
                              float* eps_out, float* std_b, void* stream);
/* u = s (z - z0[b])  (ivae_ardae.py:827) */
int ardae_center_scale(const float* latent, const float* z0, int B, int nz, int z, float std_scale, float* u, void* stream);
/* Philox4x32-10 counter RNG (replaces torch.randn at ivae/mnist.py:73, ivae_ardae.py:761, graddae/mlp.py:22) */
int ardae_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
int ardae_philox_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* dynamic binarisation x ~ Bernoulli(p[col]) (datasets/mnist.py:36-40) */
int ardae_bernoulli(const float* p, int64_t rows, int cols, float* out, uint64_t seed, uint64_t offset, void* stream);
/* utils/optim.py:49-108 (vendored Adam, eps before the bias correction); vmax non-NULL = amsgrad; step >= 1 */
int ardae_adam_ref_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, int64_t n,
                        double lr, double beta1, double beta2, double eps, int step, void* stream);
/* Device-resident step state, so that a whole train step (ivae_ardae.py:707-846) can be captured once in a HIP graph and
 * replayed: kernel arguments are frozen at capture, the 32-byte state block is not.  Layout: u64 rng_offset, i64 adam_step,
 * f32 lr/(1-beta1^t), f32 sqrt(1-beta2^t); zero-initialise it.  `advance` (first node of the step) adds rng_inc to the
 * Philox base offset, increments t and refreshes the two Adam coefficients (utils/optim.py:84-104, computed in double);
 * the _dev variants read the state instead of taking offset / step by value (offset = state.rng_offset + offset_add). */
#define ARDAE_STEP_STATE_BYTES 32
int ardae_step_state_advance(void* state, uint64_t rng_inc, double lr, double beta1, double beta2, void* stream);
int ardae_philox_normal_dev(float* out, int64_t n, uint64_t seed, const void* state, uint64_t offset_add, void* stream);
/* Elements [first_element, first_element + n) of the draw identified by (seed, offset + state.rng_offset): a rank that holds
 * rows [r0, r1) of a [rows, cols] draw passes first_element = r0 * cols (a multiple of 4) and gets exactly the numbers a single
 * process would have put there, so a data-parallel run does not depend on the number of ranks.  state may be NULL. */
int ardae_philox_normal_at(float* out, int64_t n, uint64_t seed, uint64_t offset, const void* state, uint64_t first_element,
                           void* stream);
int ardae_adam_ref_step_dev(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, int64_t n,
                            double beta1, double beta2, double eps, const void* state, void* stream);
/* torch.optim.RMSprop(lr, momentum) as built at ivae_ardae.py:625-626 (alpha .99, eps 1e-8, not centred) */
int ardae_rmsprop_step(float* p, const float* g, float* square_avg, float* momentum_buffer, int64_t n, double lr,
                       double alpha, double eps, double momentum, void* stream);
/* torch.optim.SGD as ivae_ardae.py:546-547 (model) / :613-614 (cDAE) construct it - lr only: p -= lr g */
int ardae_sgd_step(float* p, const float* g, int64_t n, double lr, void* stream);

/* ---- K4-K6: conditional AR-DAE (models/graddae/mlp.py:341-483, models/resdae/mlp.py:286-413) -----------------
 * Parameters live in ONE flat fp32 buffer in the reference's named_parameters() order
 * (ctx_encode.layers.*, ctx_encode.fc, inp_encode.*, neglogprob.* | dae.*), each tensor [out,in] row-major. */
typedef struct ardae_cdae_desc {
  int kind;        /* 0 = mlp-grad (MLPGradCARDAE: score = input-gradient of an energy MLP), 1 = mlp-res (direct score) */
  int input_dim;   /* z */
  int context_dim; /* c */
  int h_dim;
  int n_layers;    /* --cdae-n-layers */
  int act;         /* any ARDAE_ACT_* but NONE (with a piecewise linear one mlp-grad's second-order terms are zero)  */
} ardae_cdae_desc;
size_t ardae_cdae_param_floats(const ardae_cdae_desc* d);
size_t ardae_cdae_packed_floats(const ardae_cdae_desc* d);
size_t ardae_cdae_workspace_floats(const ardae_cdae_desc* d, int B, int S, int need_grads);
/* refresh the MFMA-packed weight images after every optimiser step */
int ardae_cdae_pack(const ardae_cdae_desc* d, const float* params, float* packed, void* stream);
/* ConditionalARDAE.forward + loss.backward() fused: DAE loss  mean((sigma*score(xbar|ctx,sigma) + eps)^2)  and the
 * gradient of every parameter (graddae/mlp.py:400-444 + ivae_ardae.py:771).  xbar [B*S,z] (already perturbed),
 * sigma [B*S], eps [B*S,z], ctx [B,c]; loss: device scalar; grads: flat buffer (parameter layout), OVERWRITTEN
 * except neglogprob.fc.bias which receives no gradient in the reference and is left untouched; score_out optional. */
int ardae_cdae_loss_grads(const ardae_cdae_desc* d, const float* params, const float* packed, const float* xbar,
                          const float* sigma, const float* eps, const float* ctx, int B, int S, float* workspace,
                          size_t workspace_floats, float* loss, float* grads, float* score_out, void* stream);
/* North star "fused per-sample Gaussian-perturb + sigma-scaling + DAE-forward kernel": ivae_ardae.py:753-776 in one call.  ONE kernel per
 * image draws xi / eps (Philox, keyed as ardae_latent_perturb_draw), computes the latent statistics, sigma and xbar = u + sigma eps
 * (graddae/mlp.py:21-23, ivae_ardae.py:753-767) and - the rows still in LDS - the first layer of the score network's input encoder,
 * a_1 = act(xbar A_1^T + b_1) (graddae/mlp.py:414-434), written into the cDAE workspace; the rest is ardae_cdae_loss_grads from layer 2 on.
 * xbar / sigma / eps_out / std_b are written as by ardae_latent_perturb_draw (the weight gradient of A_1 and the DAE-loss layer read them).
 * ardae_cdae_perturb_fused_ok: 1 if the shape qualifies (nstd == 1, the draw kernel's conditions, z % 8 == 0, z <= 64, nz % 32 == 0,
 * h % 32 == 0, relu / softplus); otherwise call ardae_latent_perturb* and ardae_cdae_loss_grads. */
int ardae_cdae_perturb_fused_ok(const ardae_cdae_desc* d, int nz, int nstd);
int ardae_cdae_perturb_loss_grads(const ardae_cdae_desc* d, const float* params, const float* packed, const float* latent, const float* z0,
                                  const float* ctx, int B, int nz, float std_scale, float delta, uint64_t seed, uint64_t offset_xi,
                                  uint64_t offset_eps, const void* state, uint64_t first_row, float* xbar, float* sigma, float* eps_out,
                                  float* std_b, float* workspace, size_t workspace_floats, float* loss, float* grads, void* stream);
/* ConditionalARDAE.glogprob (graddae/mlp.py:446-483): score at x [B*S,z] for noise level sigma [B*S] */
int ardae_cdae_score(const ardae_cdae_desc* d, const float* params, const float* packed, const float* x,
                     const float* sigma, const float* ctx, int B, int S, float* workspace, size_t workspace_floats,
                     float* score_out, void* stream);


/* ---- K2/K8: implicit-posterior VAE (models/ivae/mnist.py, models/ivae/toy.py enc_type='concat') ---------------
 * Parameters: ONE flat fp32 buffer in the reference's named_parameters() order
 *   kind 0 (MNISTIPVAE): encode.inp_encode.{layers.0..n_layers, fc}, encode.fc.{layers.0, fc}, decode.main.{layers.*, fc},
 *                        decode.reparam.logit_fn;   Bernoulli decoder, x rescaled to 2x-1 inside the encoder
 *   kind 1 (ToyIPVAE)  : encode.inp_encode.{layers.0..n_layers-2, fc}, encode.fc.{layers.0..n_layers-1, fc} (ContextConcatMLP:
 *                        every layer eats [hidden, noise]), decode.main.*, decode.reparam.{mean_fn, logvar_fn}; Gaussian decoder
 *   kind 2 (ConvIPVAE) : models/ivae/conv.py (28 x 28 x 1 only)
 *   kind 3 (MNISTAuxIPVAE, models/ivae/auxmnist.py): encode.aux_encode.{main.*, reparam.{mean_fn, logvar_fn}},
 *                        encode.encode.{fc.*, reparam.{mean_fn, logvar_fn}}, decode.main.*, decode.reparam.logit_fn.  Its sampler takes TWO
 *                        draws per call: every `noise` argument of this kind is ONE [rows, noise_dim + z_dim] tensor, row = [eps0 | eps]
 *                        (z0 = mu0 + exp(lv0/2) eps0,  z = mu + exp(lv/2) eps)
 *   kind 4 (MNISTConvAuxIPVAE, models/ivae/auxconv.py): the same hierarchical sampler with two conv trunks
 *                        (encode.aux_encode.{conv1-3, fc, reparam.*}, encode.encode.{conv1-3, fc, reparam.*}) and ConvIPVAE's decoder;
 *                        noise_dim = z0_dim, h_dim = 800 (the fc width), 28 x 28 x 1 only; same noise layout as kind 3
 *   kind 5 (ResConvIPVAE, models/ivae/resconv.py, `--model resconvct-res`: do_center, enc_type 'res-wn-mlp'): weight-normalised
 *                        residual blocks (models/layers2.py:50-93,237-352; models/layers.py:25-85,559-622), each operator's
 *                        parameters in the order direction, scale, bias: encode.inp_encode.{0,2,4,6,8}.{conv_0h,conv_h1,conv_01},
 *                        encode.inp_encode.11.{dot_0h,dot_h1,dot_01}, encode.fc.layers.0.*, encode.fc.fc.*, decode.dec.{0,2}.dot_*,
 *                        decode.dec.{6,8,12,14,17}.conv_*; c_dim 512, h_dim = the ResMLP width, n_layers 1, act ARDAE_ACT_ELU, 28 x 28 x 1
 *   kind 6 (MNISTResConvAuxIPVAE, models/ivae/auxresconv.py, `--model auxresconvct`): the same trunk as encode.inp_encode.enc.*
 *                        (c_dim = h_dim, 450 in the recipe), encode.aux_encode.reparam.{mean_fn,logvar_fn}, encode.encode.fc.0,
 *                        encode.encode.reparam.{mean_fn,logvar_fn} (log-variances clipped 'spm4'), the same decoder; noise_dim = z0_dim,
 *                        noise layout of kind 3; its hidden1a context is h [B, h_dim]
 *   kind 7 (ToyAuxIPVAE, models/ivae/auxtoy.py; `--model auxmlp`): kind 3's networks WITHOUT the 2x - 1 rescale, the Gaussian decoder of
 *                        kind 1 (decode.main.*, decode.reparam.{mean_fn, logvar_fn}), and a SQUARE sampling scheme: a call with nz rows per
 *                        image (nz = q^2, else refused) draws q z0's per image and q z's per z0 (auxtoy.py:215,230: `_nz = int(sqrt(nz))`);
 *                        noise of such a call = [eps0: B q x noise_dim | eps: B q q x z_dim], two consecutive blocks; hidden1a context
 *                        cat(h0, h) [B, 2 h_dim] */
typedef struct ardae_model_desc {
  int kind;
  int input_dim, noise_dim, h_dim, z_dim;
  int n_layers; /* --model-n-layers */
  int act;      /* any ARDAE_ACT_* but NONE; kinds 5 / 6: ARDAE_ACT_ELU */
  int flags;    /* kinds 5 / 6: ARDAE_MODEL_NO_CENTER = do_center False (--model resconv-res / auxresconv: the trunk sees x, not 2x - 1;
                 * ivae/resconv.py:131-132, vae/auxresconv.py:56-58); 0 elsewhere */
} ardae_model_desc;
enum { ARDAE_MODEL_NO_CENTER = 1,
       /* kind 5 (ResConvIPVAE): the sampler head `encode.fc` (models/ivae/resconv.py:101-116, ivae_ardae.py:323-442), flags bits 1-3:
        * 0 'res-wn-mlp' (--model resconv[ct]-res), 1 'mlp' (resconv[ct]), 2 'res-mlp' (-res2), 3 'res-wn-mlp-lin' (-res3), 4 'res-mlp-lin' (-res4) */
       ARDAE_MODEL_HEAD_SHIFT = 1, ARDAE_MODEL_HEAD_MASK = 7 << 1,
       /* kind 6: MNISTResConvAuxIPVAEClipped (--model auxresconv-clip / auxresconvct-clip, models/ivae/auxresconv2.py:71-72,91): the two
        * log-variance heads WITHOUT the 'spm4' clip, and z0 = mu0 + (std exp(lv0 / 2) + 1) eps0 (`min_std = 1.`).  A sampler call with noise
        * takes it as the unscaled draws (std = 1); a std = 0 pass is a RANDOM draw for this class: ardae_model_encode_hidden_raw */
       ARDAE_MODEL_CLIPPED = 16,
       /* kinds 3 / 7 (MNISTAuxIPVAE / ToyAuxIPVAE): NormalDistribution.clip_logvar of the two Gaussian heads (models/reparam.py:17-41;
        * constructor arguments clip_z0_logvar / clip_z_logvar, models/ivae/auxmnist.py:56-72,144-161) - a code per head:
        * 0 none, 1 'hard' (clamp to [-4, 2]), 2 'softplus', 3 .. 8 'spm10' / 'spm6' / 'spm5' / 'spm4' / 'spm3' / 'spm2'
        * (softplus(x + k) - k), 9 'tanh', 10 '2tanh'.  z0 head: flags bits 8-11, z head: bits 12-15. */
       ARDAE_MODEL_CLIP_Z0_SHIFT = 8, ARDAE_MODEL_CLIP_Z_SHIFT = 12, ARDAE_MODEL_CLIP_MASK = 0xff00 };
size_t ardae_model_param_floats(const ardae_model_desc* d);
size_t ardae_model_packed_floats(const ardae_model_desc* d);
/* mode 0: encode only; mode 1: vae_forward + vae_backward; mode 2: decode only (B = rows, nz = 1); mode 3: encode_pair */
size_t ardae_model_workspace_floats(const ardae_model_desc* d, int B, int nz, int mode);
int ardae_model_pack(const ardae_model_desc* d, const float* params, float* packed, void* stream);
/* Encoder.forward (ivae/mnist.py:102-121): z[B*nz, z] = f(x[B, input_dim], noise[B*nz, noise_dim]); noise NULL = zeros,
 * i.e. encode(x, std=0) */
int ardae_model_encode(const ardae_model_desc* d, const float* params, const float* packed, const float* x,
                       const float* noise, int B, int nz, float* workspace, size_t workspace_floats, float* z_out,
                       void* stream);
/* The two sampler calls that open a cDAE update (ivae_ardae.py:735,749) on the same images in one pass: z0 = encode(x, std=0)
 * [B, z] and z = forward_hidden(x, nz) [B*nz, z] share the per-image trunk (inp_encode and the image half of the first
 * concat layer), which is computed once.  Workspace: ardae_model_workspace_floats(d, B, nz, 3).
 * phase 0: everything; phase 1: trunk + z0 only (no noise needed yet); phase 2: the B*nz-row part only, reading the trunk a
 * phase-1 call left in the same workspace - so a caller can wait for its noise between the two. */
int ardae_model_encode_pair(const ardae_model_desc* d, const float* params, const float* packed, const float* x,
                            const float* noise, int B, int nz, float* workspace, size_t workspace_floats, float* z0_out,
                            float* z_out, int phase, void* stream);
/* ImplicitPosteriorVAE.forward (ivae/mnist.py:267-301): z_out [B*nz, z]; losses[3] = {loss, recon.mean, prior.mean}
 * (device); activations stay in `workspace` for the backward call */
int ardae_model_vae_forward(const ardae_model_desc* d, const float* params, const float* packed, const float* x,
                            const float* noise, int B, int nz, float beta, float* workspace, size_t workspace_floats,
                            float* z_out, float* losses, void* stream);
/* Aux models (kinds 3 and 4): the std = 0 pass of the sampler, returning the latent mean z0 [B, z] (may be NULL) AND the encoder hiddens
 * cat(h0, h) [B, 2 h] that --cdae-ctx-type hidden1a uses as the cDAE context (ivae_ardae.py:737-739, ivae/auxmnist.py:125-132).
 * Workspace: mode 0 with nz = 1. */
int ardae_model_encode_hidden(const ardae_model_desc* d, const float* params, const float* packed, const float* x, int B,
                              float* workspace, size_t workspace_floats, float* z0_out, float* hidden_out, void* stream);
/* The same for the clipped class (kind 6 + ARDAE_MODEL_CLIPPED): raw0 [B, noise_dim] is the unscaled eps0 its z0 keeps at std = 0
 * (z0 = mu0 + raw0; NULL: zeros).  The reference's loop makes TWO such calls per phase with separate draws - one for the context, one for the
 * latent mean (ivae_ardae.py:737-739,748 / 815-817,826) - so z0_out or hidden_out may be NULL here. */
int ardae_model_encode_hidden_raw(const ardae_model_desc* d, const float* params, const float* packed, const float* x, const float* raw0, int B,
                                  float* workspace, size_t workspace_floats, float* z0_out, float* hidden_out, void* stream);
/* Decoder.forward (ivae/mnist.py:188-199, toy.py:725-737) without the sample: head outputs for z [R, z_dim]:
 * out0 = logits (kind 0) / mean (kind 1) [R, input_dim], out1 = logvar (kind 1) or NULL.  Workspace: mode 2.
 * Used by the IWAE evaluator (ivae/mnist.py:420-425). */
int ardae_model_decode(const ardae_model_desc* d, const float* params, const float* packed, const float* z, int R,
                       float* workspace, size_t workspace_floats, float* out0, float* out1, void* stream);
/* per-row -log p(x|z) and prior energy (utils/vae.py:21-30,36-52; utils/energy.py:69-77): rows = B*nz, x [B, input_dim] */
int ardae_model_loss_rows(const ardae_model_desc* d, const float* out0, const float* out1, const float* x, const float* z,
                          int rows, int nz, float* recon_row, float* prior_row, void* stream);
/* grads = grads_beta*grads + d/dparams [ dloss*loss + <dz_extra, z> ]   (model_loss.backward() and
 * latent.backward(seed), ivae_ardae.py:804,834).  Needs the workspace of the matching vae_forward call. */
int ardae_model_vae_backward(const ardae_model_desc* d, const float* params, const float* packed, const float* x,
                             const float* noise, int B, int nz, float beta, float dloss, const float* dz_extra,
                             float* workspace, size_t workspace_floats, float* grads, float grads_beta, void* stream);
/* The same backward in two calls, for callers that compute the entropy seed late (ivae_ardae.py:829-834: it needs the UPDATED
 * cDAE): _decoder = model_loss.backward() through the decoder down to dL/dz (:804; needs only the vae_forward workspace, so it
 * can run beside the cDAE update), _sampler = dL/dz += seed_scale * dz_extra (:834), back-propagation through the sampler and
 * all weight gradients.  MLP models (kind 0 / 1); the conv model keeps the single call. */
int ardae_model_vae_backward_decoder(const ardae_model_desc* d, const float* params, const float* packed, const float* x,
                                     const float* noise, int B, int nz, float beta, float dloss, float* workspace,
                                     size_t workspace_floats, void* stream);
int ardae_model_vae_backward_sampler(const ardae_model_desc* d, const float* params, const float* packed, const float* x,
                                     const float* noise, int B, int nz, const float* dz_extra, float seed_scale,
                                     float* workspace, size_t workspace_floats, float* grads, float grads_beta, void* stream);


/* ---- evaluation / visualisation side of the model surface (csrc/eval_kernels.hip) ----------------------------
 * Decoder samples that ImplicitPosteriorVAE.forward / .generate return next to the losses (ivae/mnist.py:188-199,300,316):
 * relaxed Bernoulli (models/reparam.py:111-158, BernoulliDistribution.sample_logistic_sigmoid):
 *   sample = sigmoid((logit + log(u/(1-u) + 1e-20)) / temperature), mean = sigmoid(logit); u ~ U[0,1) from the caller
 *   (ardae_philox_uniform); sample or mean may be NULL. */
int ardae_relaxed_bernoulli(const float* logit, const float* u, int64_t n, float temperature, float* sample, float* mean,
                            void* stream);
/* models/reparam.py:42-51 (sample_gaussian; Gaussian decoder of ivae/toy.py:725-737): sample = mu + exp(logvar/2) * eps */
int ardae_gaussian_sample(const float* mu, const float* logvar, const float* eps, int64_t n, float* sample, void* stream);
/* Lower Cholesky factors of `batch` symmetric n x n matrices (n <= 64), one launch: the factorisation inside
 * MultivariateNormal(mu, cov) of the IWAE proposal (ivae/mnist.py:397-406).  Not positive definite -> NaNs in that factor. */
int ardae_cholesky_batched(const float* A, int batch, int n, float* L, void* stream);


/* ---- scalar log channel + static-binarised batches ------------------------------------------------
 * The scalars the reference logs per --log-interval (ivae_ardae.py:850-906; five .item() synchronisations per step there,
 * :756-758,774,837-841) leave the step through a device ring buffer instead: one record per step, written by ONE kernel at
 * the end of the step (graph-replayable: the slot comes from the device step state, iter = Adam's t), read in bulk by the host.
 * Record (ARDAE_LOG_RECORD_FLOATS floats): [0] iter (low 31 bits, int bit pattern), [1] model loss, [2] recon, [3] prior,
 * [4] beta, [5] cDAE loss, [6..8] mean / max / min over the B images of std (ivae_ardae.py:756-758), [9] cDAE lr,
 * [10] iter >> 31.  ring: [capacity, ARDAE_LOG_RECORD_FLOATS]. */
#define ARDAE_LOG_RECORD_FLOATS 16
int ardae_log_scalars(const float* cdae_loss, const float* model_losses, const float* std_b, int B, float beta, float d_lr,
                      const void* state, float* ring, int capacity, void* stream);
/* out[b, :] = table[idx[b], :]: a batch of a statically binarised set (fixed pre-drawn rows, datasets/sbmnist.py:34-60) by
 * int64 row indices, all on the device */
int ardae_gather_rows(const float* table, const int64_t* idx, int B, int D, float* out, void* stream);


/* ---- data-parallel gradient exchange: RCCL over xGMI (SURVEY 8(b) `dp_allreduce_flat`, 8(e); kernel inventory K11) -----------
 * The reference has no collectives (single process, single device: SURVEY 2.1).  Under data parallelism each rank holds
 * a shard of the image batch and the reference's batch-mean losses (`cdae_loss.backward()` ivae_ardae.py:771,
 * `model_loss.backward()` :804, the entropy seed :826-834) become a mean over ranks of the two flat gradient buffers.
 * One communicator per rank, created once; one in-place all-reduce per buffer, issued on `stream` like any kernel
 * of this library - it may be CAPTURED into the step's HIP graph.
 * RCCL is bound at run time (dlopen; a copy already resident in the process - PyTorch's - is reused): without a usable
 * librccl.so these entry points fail with a message and everything else works.  Status codes of this group:
 * 0 ok, <0 invalid argument, 1..999 a hipError_t, 1000 + n an ncclResult_t n.                                         */
#define ARDAE_DP_UNIQUE_ID_BYTES 128
/* "RCCL <version> (<path of the bound library>)", "" when none could be bound */
const char* ardae_dp_backend(void);
/* rank 0: a fresh id (ncclGetUniqueId) into HOST memory; the caller ships the 128 bytes to every rank
 * (torch.distributed broadcast, MPI, a file ...) */
int ardae_dp_unique_id(void* host_id);
/* every rank, with its GPU selected (hipSetDevice / torch.cuda.set_device): joins the communicator; blocks until all
 * `nranks` processes have called it.  One rank per device (RCCL refuses two ranks on one GPU). */
int ardae_dp_comm_create(const void* host_id, int nranks, int rank, void** comm_out);
/* what RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank) + the device it was created on */
int ardae_dp_comm_query(void* comm, int* nranks, int* rank, int* device);
int ardae_dp_comm_destroy(void* comm);
/* buf[0..n) <- mean over ranks of buf[0..n), in place, stream-ordered (ncclAllReduce, ncclAvg).  One rank: unchanged. */
int ardae_dp_allreduce_mean(void* comm, float* buf, size_t n, void* stream);


/* ---- live per-kernel timing (bench.py roofline): HIP events around every launch on the launch stream ---------- */
typedef struct ardae_profile_entry {
  char name[96];   /* kernel name as rocprofv3 prints it (template arguments included)                         */
  int calls;
  double total_ms; /* sum of event-to-event durations                                                         */
  double flops;    /* algorithmic FLOPs (2*MAC) of those launches                                             */
  double bytes;    /* algorithmic HBM bytes of those launches (operands read once + results written once)     */
} ardae_profile_entry;
int ardae_profile_enable(int on);
/* Diagnostics: a one-thread kernel that writes the device's constant 100 MHz clock (s_memrealtime) to slots[slot] when the stream
 * reaches it - an unprofiled timeline of a step (also inside captured graphs; tools / scratch use it, the product does not). */
int ardae_debug_stamp(unsigned long long* slots, int slot, void* stream);
/* synchronises, aggregates by kernel, clears the log; returns the number of distinct kernels */
int ardae_profile_report(ardae_profile_entry* entries, int max_entries);

#ifdef __cplusplus
}
#endif
#endif /* ARDAE_HIP_H */
