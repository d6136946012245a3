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
                     float* out0, hipStream_t st);
int aux_model_vae_forward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                          float beta, float* workspace, size_t wsf, float* z_out, float* losses, hipStream_t st);
int aux_model_vae_backward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                           float beta, float dloss, const float* dz_extra, float* workspace, size_t wsf, float* grads, float grads_beta,
                           hipStream_t st);
}  // namespace ardae
