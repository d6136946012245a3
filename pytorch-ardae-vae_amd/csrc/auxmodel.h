// MNISTAuxIPVAE (`--model auxmnist`, ardae_model_desc.kind == 3) entry points; dispatched from csrc/model.hip.
#pragma once
#include "ardae_hip.h"
#include "common.h"

namespace ardae {
size_t aux_model_param_floats(const ardae_model_desc& d);
size_t aux_model_packed_floats(const ardae_model_desc& d);
size_t aux_model_workspace_floats(const ardae_model_desc& d, int B, int nz, int mode);
int aux_model_pack(const ardae_model_desc& d, const float* params, float* packed, hipStream_t st);
// noise [B*nz, noise_dim + z_dim] (rows [eps0 | eps]) or null = zeros; hidden_out [B, 2 h] (nz == 1) or null
int aux_model_encode(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                     float* workspace, size_t wsf, float* z_out, float* hidden_out, hipStream_t st);
int aux_model_decode(const ardae_model_desc& d, const float* params, const float* packed, const float* z, int R, float* workspace, size_t wsf,
                     float* out0, hipStream_t st, float* out1 = nullptr);
int aux_model_vae_forward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                          float beta, float* workspace, size_t wsf, float* z_out, float* losses, hipStream_t st);
int aux_model_vae_backward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                           float beta, float dloss, const float* dz_extra, float* workspace, size_t wsf, float* grads, float grads_beta,
                           hipStream_t st);
// the two reparameterisation steps shared with the hierarchical conv model (csrc/convmodel.hip)
//   out[r][c] = mu[g][c] + exp(lv[g][c] / 2) * eps[r * ld_eps + c],  g = r / rows_per_group   (mu, lv: [rows / rpg, cols])
// min_std / raw (the clipped aux-resconv class, ivae/auxresconv2.py:29-36,91): out = mu + exp(lv / 2) eps + min_std (raw ? raw : eps) - `eps` carries
// std * draw, the extra term the UNSCALED draw (raw [rows, cols], row stride ld_raw; NULL: eps itself, i.e. std = 1)
int launch_reparam_fwd(const float* mu, const float* lv, const float* eps, int ld_eps, int64_t rows, int cols, int rows_per_group, float* out,
                       hipStream_t st, float min_std = 0.f, const float* raw = nullptr, int ld_raw = 0);
// dlv = dz (z - mu - min_std eps) / 2   (eps: the draw of the forward call, needed only with min_std != 0)
int launch_reparam_bwd(const float* dz, const float* z, const float* mu, int64_t rows, int cols, int rows_per_group, float* dlv, hipStream_t st,
                       float min_std = 0.f, const float* eps = nullptr, int ld_eps = 0);
}  // namespace ardae
