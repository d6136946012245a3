// Weight-normalised residual-conv models (ardae_model_desc.kind 5: ResConvIPVAE `--model resconvct-res`, 6: MNISTResConvAuxIPVAE
// `--model auxresconvct`) entry points; dispatched from csrc/model.hip.  Noise: kind 5 [rows, noise_dim]; kind 6 [rows, noise_dim + z_dim]
// (rows [eps0 | eps], noise_dim = z0_dim); the hidden1a context of kind 6 is h [B, c_dim = h_dim] (ivae/auxresconv.py:125-132).
#pragma once
#include "ardae_hip.h"
#include "common.h"

namespace ardae {
size_t res_model_param_floats(const ardae_model_desc& d);
size_t res_model_packed_floats(const ardae_model_desc& d);
size_t res_model_workspace_floats(const ardae_model_desc& d, int B, int nz, int mode);
int res_model_pack(const ardae_model_desc& d, const float* params, float* packed, hipStream_t st);
int res_model_encode(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                     float* workspace, size_t wsf, float* z_out, float* hidden_out, hipStream_t st, const float* raw0 = nullptr);
int res_model_decode(const ardae_model_desc& d, const float* params, const float* packed, const float* z, int R, float* workspace, size_t wsf,
                     float* out0, hipStream_t st);
int res_model_vae_forward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                          float beta, float* workspace, size_t wsf, float* z_out, float* losses, hipStream_t st);
int res_model_vae_backward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                           float beta, float dloss, const float* dz_extra, float* workspace, size_t wsf, float* grads, float grads_beta,
                           hipStream_t st);
}  // namespace ardae
