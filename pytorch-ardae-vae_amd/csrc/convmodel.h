// ConvIPVAE (`--model mnist-conv`, ardae_model_desc.kind == 2) entry points; dispatched from csrc/model.hip.
#pragma once
#include "ardae_hip.h"
#include "common.h"

namespace ardae {
size_t conv_model_param_floats(const ardae_model_desc& d);
size_t conv_model_packed_floats(const ardae_model_desc& d);
size_t conv_model_workspace_floats(const ardae_model_desc& d, int B, int nz, int mode);
int conv_model_pack(const ardae_model_desc& d, const float* params, float* packed, hipStream_t st);
int conv_model_encode(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B,
                      int nz, float* workspace, size_t wsf, float* z_out, hipStream_t st);
int conv_model_decode(const ardae_model_desc& d, const float* params, const float* packed, const float* z, int R, float* workspace,
                      size_t wsf, float* out0, hipStream_t st);
int conv_model_vae_forward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B,
                           int nz, float beta, float* workspace, size_t wsf, float* z_out, float* losses, hipStream_t st);
int conv_model_vae_backward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B,
                            int nz, float beta, float dloss, const float* dz_extra, float* workspace, size_t wsf, float* grads,
                            float grads_beta, hipStream_t st);
// MNISTConvAuxIPVAE (`--model auxconv`, kind == 4): same entry points (noise [R, noise_dim + z_dim], hidden context [B, 1600])
size_t auxconv_model_param_floats(const ardae_model_desc& d);
size_t auxconv_model_packed_floats(const ardae_model_desc& d);
size_t auxconv_model_workspace_floats(const ardae_model_desc& d, int B, int nz, int mode);
int auxconv_model_pack(const ardae_model_desc& d, const float* params, float* packed, hipStream_t st);
int auxconv_model_encode(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                         float* workspace, size_t wsf, float* z_out, float* hidden_out, hipStream_t st);
int auxconv_model_decode(const ardae_model_desc& d, const float* params, const float* packed, const float* z, int R, float* workspace, size_t wsf,
                         float* out0, hipStream_t st);
int auxconv_model_vae_forward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B,
                              int nz, float beta, float* workspace, size_t wsf, float* z_out, float* losses, hipStream_t st);
int auxconv_model_vae_backward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B,
                               int nz, float beta, float dloss, const float* dz_extra, float* workspace, size_t wsf, float* grads,
                               float grads_beta, hipStream_t st);
}  // namespace ardae
