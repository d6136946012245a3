// ConvIPVAE (models/ivae/conv.py:44-245 + models/vae/conv.py:79-136; `--model mnist-conv`, BASELINE config #4) on gfx950.
//
// The reference hard-wires the architecture (28x28x1 input; conv 1->16->32->32, k5 s2 p2: 28->14->7->4; fc4 (512+noise)->800,
// fc5 800->z; decoder MLP z->300->512, ConvTranspose2d 32->32 (4->7, zero-padded to 8), 32->16 (8->15), 16->1 (15->29,
// cropped to 28)), so this file does too.  Every (transposed) convolution is im2col / col2im (two small gather kernels,
// NHWC activations as [rows = b*H*W, C]) around the FP32-MFMA linear and wgrad kernels; the conv trunk runs on B rows only,
// fc4's image half enters as a per-image row bias exactly like the MLP sampler (csrc/model.hip).
#include <vector>

#include "ardae_hip.h"
#include "common.h"
#include "auxmodel.h"
#include "convmodel.h"
#include "elementwise.h"
#include "linear.h"
#include "wgrad.h"

namespace ardae {
namespace {

// collect pack requests; flushed with one launch by PACK_FLUSH
#define PACK_PUSH(W_, ldw_, nout_, k_, tr_, out_) pack_items__.push_back(PackItem{W_, ldw_, nout_, k_, (tr_) ? 1 : 0, out_})
#define PACK_FLUSH(st_) ARDAE_TRY(launch_pack_batch(pack_items__.data(), (int)pack_items__.size(), st_))

// ------------------------------------------------------------------------------------------------ data-movement kernels
// cols[(b*OH+oh)*OW+ow][c*25+kh*5+kw] = x[b][2oh-2+kh][2ow-2+kw][c]  (0 outside H x W); x is NHWC [B,H,W,C]
__global__ void im2col_s2_kernel(const float* __restrict__ x, int H, int W, int C, int OH, int OW, float* __restrict__ cols, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int J = C * 25;
  const int j = (int)(e % J);
  const int64_t row = e / J;
  const int ow = (int)(row % OW), oh = (int)((row / OW) % OH);
  const int64_t b = row / ((int64_t)OW * OH);
  const int c = j / 25, kk = j - c * 25, kh = kk / 5, kw = kk - kh * 5;
  const int h = 2 * oh - 2 + kh, w = 2 * ow - 2 + kw;
  cols[e] = (h >= 0 && h < H && w >= 0 && w < W) ? x[((b * H + h) * W + w) * C + c] : 0.f;
}

// out[b][oh][ow][o] = f( bias[o] + sum_{kh,kw} cols[(b*IH+ih)*IW+iw][o*25+kh*5+kw] ),  ih = (oh+2-kh)/2 exact and in range;
// positions with oh >= VH or ow >= VW are written as 0 (the reference's ZeroPad2d after the activation).  `mulS`: multiply
// by act'(S) instead of applying an activation (backward-data of a convolution followed by the previous layer's act').
__global__ void col2im_s2_kernel(const float* __restrict__ cols, int IH, int IW, int O, int OH, int OW, int VH, int VW,
                                 const float* __restrict__ bias, int act, const float* __restrict__ mulS, float* __restrict__ out,
                                 int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int o = (int)(e % O);
  const int64_t pix = e / O;
  const int ow = (int)(pix % OW), oh = (int)((pix / OW) % OH);
  const int64_t b = pix / ((int64_t)OW * OH);
  float v = 0.f;
  if (oh < VH && ow < VW) {
    v = bias ? bias[o] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
      const int th = oh + 2 - kh;
      if (th < 0 || (th & 1)) continue;
      const int ih = th >> 1;
      if (ih >= IH) continue;
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const int tw = ow + 2 - kw;
        if (tw < 0 || (tw & 1)) continue;
        const int iw = tw >> 1;
        if (iw >= IW) continue;
        v += cols[((b * IH + ih) * IW + iw) * (int64_t)(O * 25) + o * 25 + kh * 5 + kw];
      }
    }
    if (mulS) v *= act_d1_rt(act, mulS[e]);
    else v = act_fwd_rt(act, v);
  }
  out[e] = v;
}

// [B, HW, C] (NHWC rows) <-> [B, C*HW] (PyTorch's .view(B,-1) of NCHW)
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, int HW, int C, float* __restrict__ out, int64_t total, int reverse) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const int hw = (int)((e / C) % HW);
  const int64_t b = e / ((int64_t)C * HW);
  const int64_t nchw = (b * C + c) * HW + hw;
  if (reverse) out[e] = in[nchw]; else out[nchw] = in[e];
}

__global__ void mul_dact_kernel(const float* __restrict__ x, const float* __restrict__ S, int act, float* __restrict__ y, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) y[e] = x[e] * act_d1_rt(act, S[e]);
}

__global__ void fill_kernel(float* __restrict__ p, float v, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) p[e] = v;
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }

// ------------------------------------------------------------------------------------------------ layout
struct Lin { size_t w, b; int out, in; };   // weight viewed as [out, in] (convs: [O, C*25]; deconvs: [in_ch, out_ch*25])

struct ConvLayout {
  int nd, zd, act;
  Lin conv[3], fc4, fc5, dfc[2], dcv[3];
  size_t total;
  ConvLayout() : nd(0), zd(0), act(0), total(0) {}     // decoder-only view filled by AuxConvLayout
  explicit ConvLayout(const ardae_model_desc& d) : nd(d.noise_dim), zd(d.z_dim), act(d.act) {
    size_t off = 0;
    auto add = [&](Lin& l, int out, int in, int nbias) { l.out = out; l.in = in; l.w = off; off += (size_t)out * in; l.b = off; off += nbias; };
    add(conv[0], 16, 1 * 25, 16); add(conv[1], 32, 16 * 25, 32); add(conv[2], 32, 32 * 25, 32);
    add(fc4, 800, 512 + nd, 800); add(fc5, zd, 800, zd);
    add(dfc[0], 300, zd, 300); add(dfc[1], 512, 300, 512);
    add(dcv[0], 32, 32 * 25, 32); add(dcv[1], 32, 16 * 25, 16); add(dcv[2], 16, 1 * 25, 1);
    total = off;
  }
};

size_t al64(size_t n) { return (n + 63) & ~size_t(63); }

struct ConvPacked {
  size_t conv_f[3], conv_b[3], fc4i_f, fc4i_b, fc4n_f, fc5_f, fc5_b, dfc_f[2], dfc_b[2], dcv_f[3], dcv_b[3], total;
  ConvPacked() : total(0) {}
  explicit ConvPacked(const ConvLayout& P) {
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += al64(n); return o; };
    for (int i = 0; i < 3; ++i) { conv_f[i] = take(packed_floats(P.conv[i].out, P.conv[i].in)); conv_b[i] = take(packed_floats(P.conv[i].in, P.conv[i].out)); }
    fc4i_f = take(packed_floats(800, 512)); fc4i_b = take(packed_floats(512, 800)); fc4n_f = take(packed_floats(800, P.nd));
    fc5_f = take(packed_floats(P.zd, 800)); fc5_b = take(packed_floats(800, P.zd));
    for (int i = 0; i < 2; ++i) { dfc_f[i] = take(packed_floats(P.dfc[i].out, P.dfc[i].in)); dfc_b[i] = take(packed_floats(P.dfc[i].in, P.dfc[i].out)); }
    for (int i = 0; i < 3; ++i) { dcv_f[i] = take(packed_floats(P.dcv[i].in, P.dcv[i].out)); dcv_b[i] = take(packed_floats(P.dcv[i].out, P.dcv[i].in)); }
    total = off;
  }
};

struct Bump {
  float* base; size_t cap; size_t off = 0; bool ok = true;
  Bump(float* b, size_t c) : base(b), cap(c) {}
  float* take(size_t n) { size_t o = off; off += al64(n); if (off > cap) { ok = false; return base; } return base + o; }
};

// spatial sizes: encoder 28 -> 14 -> 7 -> 4; decoder grids 4 -> 8 (7 valid) -> 15 -> 28 (of 29)
struct ConvWs {
  // encoder (B rows)
  float *x2, *cols[3], *hcv[3], *inp, *rb, *t1, *z;
  // decoder (R rows)
  float *d1, *d2, *g0, *c1, *u1, *c2, *u2, *c3, *logit, *rec_row, *pri_row;
  // backward
  float *dlogit, *dc3, *dp2, *dc2, *dp1, *dc1, *dg0, *dd2, *dd1, *dzq, *dz, *dt1, *drb, *dinp, *dinp_t, *dh3, *dcols3, *dh2, *dcols2, *dh1, *ones;
};

constexpr int EH[4] = {28, 14, 7, 4};      // encoder spatial sizes
constexpr int ECH[4] = {1, 16, 32, 32};    // encoder channels

void carve(const ConvLayout& P, Bump& ws, int B, int nz, int mode, ConvWs& W) {
  const size_t R = (size_t)B * nz;
  W.x2 = ws.take((size_t)B * 784);
  for (int i = 0; i < 3; ++i) {
    W.cols[i] = ws.take((size_t)B * EH[i + 1] * EH[i + 1] * ECH[i] * 25);
    W.hcv[i] = ws.take((size_t)B * EH[i + 1] * EH[i + 1] * ECH[i + 1]);
  }
  W.inp = ws.take((size_t)B * 512); W.rb = ws.take((size_t)B * 800);
  W.t1 = ws.take(R * 800); W.z = ws.take(R * P.zd);
  if (mode == 0) return;
  W.d1 = ws.take(R * 300); W.d2 = ws.take(R * 512); W.g0 = ws.take(R * 512);
  W.c1 = ws.take(R * 16 * 800); W.u1 = ws.take(R * 64 * 32);
  W.c2 = ws.take(R * 64 * 400); W.u2 = ws.take(R * 225 * 16);
  W.c3 = ws.take(R * 225 * 25); W.logit = ws.take(R * 784);
  W.rec_row = ws.take(R); W.pri_row = ws.take(R);
  if (mode == 2) return;
  W.dlogit = ws.take(R * 784); W.dc3 = ws.take(R * 225 * 25); W.dp2 = ws.take(R * 225 * 16);
  W.dc2 = ws.take(R * 64 * 400); W.dp1 = ws.take(R * 64 * 32); W.dc1 = ws.take(R * 16 * 800);
  W.dg0 = ws.take(R * 512); W.dd2 = ws.take(R * 512); W.dd1 = ws.take(R * 300);
  W.dzq = ws.take(R * P.zd); W.dz = ws.take(R * P.zd); W.dt1 = ws.take(R * 800);
  W.drb = ws.take((size_t)B * 800); W.dinp = ws.take((size_t)B * 512); W.dinp_t = ws.take((size_t)B * 512);
  W.dh3 = ws.take((size_t)B * 512); W.dcols3 = ws.take((size_t)B * 16 * 800); W.dh2 = ws.take((size_t)B * 49 * 32);
  W.dcols2 = ws.take((size_t)B * 49 * 400); W.dh1 = ws.take((size_t)B * 196 * 16);
  W.ones = ws.take(R * 784);
}

constexpr int N_WGRAD = 18;   // problems of one backward (see conv_vae_backward)

size_t wgrad_scratch(const ConvLayout& P, int B, int R, std::vector<int>* out) {
  std::vector<int> sp; size_t tot = 0;
  auto one = [&](int M, int O, int I) { const int s = wgrad_splits(M, O, I, N_WGRAD); sp.push_back(s); tot += al64((size_t)s * O * I) + al64((size_t)s * 2 * O); };
  one(R * 225, 16, 25); one(R * 784, 1, 1);          // deconv3 weight, bias
  one(R * 64, 32, 400); one(R * 225, 16, 1);         // deconv2
  one(R * 16, 32, 800); one(R * 64, 32, 1);          // deconv1
  one(R, 512, 300); one(R, 300, P.zd);               // decoder fc
  one(R, P.zd, 800); one(R, 800, P.nd); one(B, 800, 512);   // fc5, fc4 noise half (+bias), fc4 image half
  one(B * 16, 32, 800); one(B * 49, 32, 400); one(B * 196, 16, 25);   // conv3, conv2, conv1
  if (out) *out = sp;
  return tot;
}

size_t conv_workspace(const ConvLayout& P, int B, int nz, int mode) {
  // run the carve on a null arena to count
  Bump b(nullptr, ~size_t(0));
  ConvWs W;
  carve(P, b, B, nz, mode, W);
  size_t t = b.off + al64((size_t)B * nz * P.nd);    // + zero-noise buffer for encode(std=0)
  if (mode == 1) t += wgrad_scratch(P, B, B * nz, nullptr);
  return t;
}

int lin1(int epi, int act, int M, int Nout, const float* x, int ldx, int K, const float* wp, LinArgs a, hipStream_t st) {
  a.M = M; a.Nout = Nout; a.nsrc = 1; a.act = act;
  a.src[0].x = x; a.src[0].ld = ldx; a.src[0].K = K; a.src[0].wp = wp;
  return launch_linear(a, epi, st);
}

int im2col(const float* x, int Bn, int H, int Wd, int C, int OH, int OW, float* cols, hipStream_t st) {
  const int64_t total = (int64_t)Bn * OH * OW * C * 25;
  hipLaunchKernelGGL(im2col_s2_kernel, dim3(nblk(total)), dim3(256), 0, st, x, H, Wd, C, OH, OW, cols, total);
  ARDAE_LAUNCH_CHECK();
  return 0;
}
int col2im(const float* cols, int Bn, int IH, int IW, int O, int OH, int OW, int VH, int VW, const float* bias, int act, const float* mulS,
           float* out, hipStream_t st) {
  const int64_t total = (int64_t)Bn * OH * OW * O;
  hipLaunchKernelGGL(col2im_s2_kernel, dim3(nblk(total)), dim3(256), 0, st, cols, IH, IW, O, OH, OW, VH, VW, bias, act, mulS, out, total);
  ARDAE_LAUNCH_CHECK();
  return 0;
}
int transpose_hw_c(const float* in, int Bn, int HW, int C, float* out, bool to_nchw, hipStream_t st) {
  const int64_t total = (int64_t)Bn * HW * C;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(nblk(total)), dim3(256), 0, st, in, HW, C, out, total, to_nchw ? 0 : 1);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

// conv trunk 1 -> 16 -> 32 -> 32 (k5 s2 p2, 28 -> 14 -> 7 -> 4) on the rescaled images x2 [B, 784]: fills cols / hcv and the
// NCHW-flattened output inp [B, 512] (shared by ConvIPVAE and the two trunks of the hierarchical conv model)
int trunk_fwd(const Lin* conv, const size_t* conv_f, const float* params, const float* packed, const float* x2, float* const* cols,
              float* const* hcv, float* inp, int B, int act, hipStream_t st) {
  const float* cur = x2;
  for (int i = 0; i < 3; ++i) {                                                 // conv_i = im2col + Linear([O, C*25]) + act
    const int OH = EH[i + 1], Kc = ECH[i] * 25;
    ARDAE_TRY(im2col(cur, B, EH[i], EH[i], ECH[i], OH, OH, cols[i], st));
    LinArgs A{}; A.bias = params + conv[i].b; A.Y = hcv[i]; A.ldY = ECH[i + 1];
    ARDAE_TRY(lin1(EPI_ACT, act, B * OH * OH, ECH[i + 1], cols[i], Kc, Kc, packed + conv_f[i], A, st));
    cur = hcv[i];
  }
  return transpose_hw_c(hcv[2], B, 16, 32, inp, true, st);                      // h3.view(B,-1) of NCHW
}

// backward of the trunk from d(inp) [B, 512] (NCHW-flat): dh3 / dh2 / dh1 = d(pre-activation) of conv3 / conv2 / conv1 (NHWC rows)
int trunk_bwd(const size_t* conv_b, const float* packed, const float* dinp, float* dinp_t, float* const* hcv, float* dh3, float* dcols3,
              float* dh2, float* dcols2, float* dh1, int B, int act, hipStream_t st) {
  ARDAE_TRY(transpose_hw_c(dinp, B, 16, 32, dinp_t, false, st));               // NCHW-flat -> NHWC rows
  {
    const int64_t n = (int64_t)B * 512;
    hipLaunchKernelGGL(mul_dact_kernel, dim3(nblk(n)), dim3(256), 0, st, dinp_t, hcv[2], act, dh3, n);
    ARDAE_LAUNCH_CHECK();
  }
  { LinArgs A{}; A.Y = dcols3; A.ldY = 800;                                     // conv3 backward-data: dcols = dpre . W3
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B * 16, 800, dh3, 32, 32, packed + conv_b[2], A, st)); }
  ARDAE_TRY(col2im(dcols3, B, 4, 4, 32, 7, 7, 7, 7, nullptr, act, hcv[1], dh2, st));
  { LinArgs A{}; A.Y = dcols2; A.ldY = 400;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B * 49, 400, dh2, 32, 32, packed + conv_b[1], A, st)); }
  return col2im(dcols2, B, 7, 7, 16, 14, 14, 14, 14, nullptr, act, hcv[0], dh1, st);
}

int conv_encode_fwd(const ConvLayout& P, const ConvPacked& K, const float* params, const float* packed, const float* x, const float* noise,
                    int B, int nz, ConvWs& W, float* z_out, hipStream_t st) {
  const int R = B * nz, act = P.act;
  ARDAE_TRY(launch_affine(x, (int64_t)B * 784, 2.f, -1.f, W.x2, st));           // ivae/conv.py:81
  ARDAE_TRY(trunk_fwd(P.conv, K.conv_f, params, packed, W.x2, W.cols, W.hcv, W.inp, B, act, st));
  {
    LinArgs A{}; A.bias = params + P.fc4.b; A.Y = W.rb; A.ldY = 800;            // image half of fc4, once per image
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, 800, W.inp, 512, 512, packed + K.fc4i_f, A, st));
  }
  {
    LinArgs A{}; A.rowbias = W.rb; A.rowbias_ld = 800; A.rows_per_group = nz; A.Y = W.t1; A.ldY = 800;
    ARDAE_TRY(lin1(EPI_ACT, act, R, 800, noise, P.nd, P.nd, packed + K.fc4n_f, A, st));
  }
  {
    LinArgs A{}; A.bias = params + P.fc5.b; A.Y = W.z; A.ldY = P.zd;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.zd, W.t1, 800, 800, packed + K.fc5_f, A, st));
  }
  if (z_out) ARDAE_TRY(launch_copy(W.z, (size_t)R * P.zd, z_out, st));
  return 0;
}

int conv_decode_fwd(const ConvLayout& P, const ConvPacked& K, const float* params, const float* packed, const float* z, int R, ConvWs& W,
                    hipStream_t st) {
  const int act = P.act;
  { LinArgs A{}; A.bias = params + P.dfc[0].b; A.Y = W.d1; A.ldY = 300; ARDAE_TRY(lin1(EPI_ACT, act, R, 300, z, P.zd, P.zd, packed + K.dfc_f[0], A, st)); }
  { LinArgs A{}; A.bias = params + P.dfc[1].b; A.Y = W.d2; A.ldY = 512; ARDAE_TRY(lin1(EPI_ACT, act, R, 512, W.d1, 300, 300, packed + K.dfc_f[1], A, st)); }
  ARDAE_TRY(transpose_hw_c(W.d2, R, 16, 32, W.g0, false, st));                  // h1.view(R,32,4,4) -> NHWC rows
  // deconv1 32->32: 4x4 -> 7x7, activation, zero-pad to 8x8
  { LinArgs A{}; A.Y = W.c1; A.ldY = 800; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R * 16, 800, W.g0, 32, 32, packed + K.dcv_f[0], A, st)); }
  ARDAE_TRY(col2im(W.c1, R, 4, 4, 32, 8, 8, 7, 7, params + P.dcv[0].b, act, nullptr, W.u1, st));
  // deconv2 32->16: 8x8 -> 15x15, activation
  { LinArgs A{}; A.Y = W.c2; A.ldY = 400; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R * 64, 400, W.u1, 32, 32, packed + K.dcv_f[1], A, st)); }
  ARDAE_TRY(col2im(W.c2, R, 8, 8, 16, 15, 15, 15, 15, params + P.dcv[1].b, act, nullptr, W.u2, st));
  // logit deconv 16->1: 15x15 -> 29x29, cropped to 28x28
  { LinArgs A{}; A.Y = W.c3; A.ldY = 25; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R * 225, 25, W.u2, 16, 16, packed + K.dcv_f[2], A, st)); }
  ARDAE_TRY(col2im(W.c3, R, 15, 15, 1, 28, 28, 28, 28, params + P.dcv[2].b, ACT_NONE, nullptr, W.logit, st));
  return 0;
}

// decoder backward from W.dlogit (and W.dzq = prior + seed part of dL/dz): fills dc3/dp2/dc2/dp1/dc1/dg0/dd2/dd1 and W.dz
int conv_decoder_bwd(const ConvLayout& P, const ConvPacked& K, const float* packed, ConvWs& W, int R, hipStream_t st) {
  const int act = P.act;
  {
    const int64_t n = (int64_t)R * 784;
    hipLaunchKernelGGL(fill_kernel, dim3(nblk(n)), dim3(256), 0, st, W.ones, 1.0f, n);
    ARDAE_LAUNCH_CHECK();
  }
  // ---- decoder backward.  d(cols) of a transposed conv = im2col of the output gradient over the deconv's INPUT grid.
  ARDAE_TRY(im2col(W.dlogit, R, 28, 28, 1, 15, 15, W.dc3, st));                 // cropped row/col 28 has no gradient
  { LinArgs A{}; A.S = W.u2; A.ldS = 16; A.Y = W.dp2; A.ldY = 16;               // dpre2 = (dc3 . W3^T) (.) act'(u2)
    ARDAE_TRY(lin1(EPI_DACT, act, R * 225, 16, W.dc3, 25, 25, packed + K.dcv_b[2], A, st)); }
  ARDAE_TRY(im2col(W.dp2, R, 15, 15, 16, 8, 8, W.dc2, st));
  { LinArgs A{}; A.S = W.u1; A.ldS = 32; A.Y = W.dp1; A.ldY = 32;               // zero at the padded positions: act'(0) = 0
    ARDAE_TRY(lin1(EPI_DACT, act, R * 64, 32, W.dc2, 400, 400, packed + K.dcv_b[1], A, st)); }
  ARDAE_TRY(im2col(W.dp1, R, 8, 8, 32, 4, 4, W.dc1, st));
  { LinArgs A{}; A.S = W.g0; A.ldS = 32; A.Y = W.dg0; A.ldY = 32;               // g0 is the (permuted) activated output of decode.fc
    ARDAE_TRY(lin1(EPI_DACT, act, R * 16, 32, W.dc1, 800, 800, packed + K.dcv_b[0], A, st)); }
  ARDAE_TRY(transpose_hw_c(W.dg0, R, 16, 32, W.dd2, true, st));                 // -> d(pre) of decode.fc.fc  [R,512]
  { LinArgs A{}; A.S = W.d1; A.ldS = 300; A.Y = W.dd1; A.ldY = 300;
    ARDAE_TRY(lin1(EPI_DACT, act, R, 300, W.dd2, 512, 512, packed + K.dfc_b[1], A, st)); }
  { LinArgs A{}; A.S = W.dzq; A.ldS = P.zd; A.Q = W.dzq; A.ldQ = P.zd; A.Y = W.dz; A.ldY = P.zd;   // + prior + injected seed
    ARDAE_TRY(lin1(EPI_DACT, ACT_NONE, R, P.zd, W.dd1, 300, 300, packed + K.dfc_b[0], A, st)); }
  return 0;
}

// the decoder's eight weight-gradient problems (order == wgrad_scratch)
template <class PUSH>
void conv_decoder_wgrads(const ConvLayout& P, ConvWs& W, int R, float* grads, PUSH&& push) {
  // ConvTranspose2d: dW[in][out*25] = sum_rows input[row][in] * dcols[row][out*25]; its bias = sum of the output gradient
  push(R * 225, 16, 25, W.u2, W.dc3, grads + P.dcv[2].w, 25, nullptr);
  push(R * 784, 1, 1, W.dlogit, W.ones, grads + P.dcv[2].b, 1, nullptr);
  push(R * 64, 32, 400, W.u1, W.dc2, grads + P.dcv[1].w, 400, nullptr);
  push(R * 225, 16, 1, W.dp2, W.ones, grads + P.dcv[1].b, 1, nullptr);
  push(R * 16, 32, 800, W.g0, W.dc1, grads + P.dcv[0].w, 800, nullptr);
  push(R * 64, 32, 1, W.dp1, W.ones, grads + P.dcv[0].b, 1, nullptr);
  push(R, 512, 300, W.dd2, W.d1, grads + P.dfc[1].w, 300, grads + P.dfc[1].b);
  push(R, 300, P.zd, W.dd1, W.z, grads + P.dfc[0].w, P.zd, grads + P.dfc[0].b);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ entry points (kind == 2)
size_t conv_model_param_floats(const ardae_model_desc& d) { return ConvLayout(d).total; }
size_t conv_model_packed_floats(const ardae_model_desc& d) { return ConvPacked(ConvLayout(d)).total; }
size_t conv_model_workspace_floats(const ardae_model_desc& d, int B, int nz, int mode) { return conv_workspace(ConvLayout(d), B, nz, mode); }

int conv_model_pack(const ardae_model_desc& d, const float* params, float* packed, hipStream_t st) {
  std::vector<PackItem> pack_items__;
  const ConvLayout P(d);
  const ConvPacked K(P);
  for (int i = 0; i < 3; ++i) {
    PACK_PUSH(params + P.conv[i].w, P.conv[i].in, P.conv[i].out, P.conv[i].in, false, packed + K.conv_f[i]);
    PACK_PUSH(params + P.conv[i].w, P.conv[i].in, P.conv[i].in, P.conv[i].out, true, packed + K.conv_b[i]);
  }
  const int ld4 = 512 + P.nd;
  PACK_PUSH(params + P.fc4.w, ld4, 800, 512, false, packed + K.fc4i_f);
  PACK_PUSH(params + P.fc4.w, ld4, 512, 800, true, packed + K.fc4i_b);
  PACK_PUSH(params + P.fc4.w + 512, ld4, 800, P.nd, false, packed + K.fc4n_f);
  PACK_PUSH(params + P.fc5.w, 800, P.zd, 800, false, packed + K.fc5_f);
  PACK_PUSH(params + P.fc5.w, 800, 800, P.zd, true, packed + K.fc5_b);
  for (int i = 0; i < 2; ++i) {
    PACK_PUSH(params + P.dfc[i].w, P.dfc[i].in, P.dfc[i].out, P.dfc[i].in, false, packed + K.dfc_f[i]);
    PACK_PUSH(params + P.dfc[i].w, P.dfc[i].in, P.dfc[i].in, P.dfc[i].out, true, packed + K.dfc_b[i]);
  }
  for (int i = 0; i < 3; ++i) {   // ConvTranspose2d weight [in, out*25]: forward = X . W (transposed pack), backward-data = dC . W^T (natural)
    PACK_PUSH(params + P.dcv[i].w, P.dcv[i].in, P.dcv[i].in, P.dcv[i].out, true, packed + K.dcv_f[i]);
    PACK_PUSH(params + P.dcv[i].w, P.dcv[i].in, P.dcv[i].out, P.dcv[i].in, false, packed + K.dcv_b[i]);
  }
  PACK_FLUSH(st);
  return 0;
}

static const float* zero_noise(Bump& ws, const ConvLayout& P, int B, int nz, hipStream_t st) {
  float* zero = ws.take((size_t)B * nz * P.nd);
  if (launch_fill(zero, (size_t)B * nz * P.nd, 0.f, st) != 0) return nullptr;
  return zero;
}

int conv_model_encode(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B,
                      int nz, float* workspace, size_t wsf, float* z_out, hipStream_t st) {
  const ConvLayout P(d);
  const ConvPacked K(P);
  Bump ws(workspace, wsf);
  ConvWs W;
  carve(P, ws, B, nz, 0, W);
  const float* nptr = noise ? noise : zero_noise(ws, P, B, nz, st);
  ARDAE_CHECK_ARG(ws.ok, "conv_model_encode: workspace too small");
  return conv_encode_fwd(P, K, params, packed, x, nptr, B, nz, W, z_out, st);
}

int conv_model_decode(const ardae_model_desc& d, const float* params, const float* packed, const float* z, int R, float* workspace,
                      size_t wsf, float* out0, hipStream_t st) {
  const ConvLayout P(d);
  const ConvPacked K(P);
  Bump ws(workspace, wsf);
  ConvWs W;
  carve(P, ws, R, 1, 2, W);
  ARDAE_CHECK_ARG(ws.ok, "conv_model_decode: workspace too small");
  ARDAE_TRY(conv_decode_fwd(P, K, params, packed, z, R, W, st));
  ARDAE_TRY(launch_copy(W.logit, (size_t)R * 784, out0, st));
  return 0;
}

int conv_model_vae_forward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B,
                           int nz, float beta, float* workspace, size_t wsf, float* z_out, float* losses, hipStream_t st) {
  const ConvLayout P(d);
  const ConvPacked K(P);
  Bump ws(workspace, wsf);
  ConvWs W;
  carve(P, ws, B, nz, 1, W);
  ARDAE_CHECK_ARG(ws.ok, "conv_model_vae_forward: workspace too small");
  const int R = B * nz;
  ARDAE_TRY(conv_encode_fwd(P, K, params, packed, x, noise, B, nz, W, z_out, st));
  ARDAE_TRY(conv_decode_fwd(P, K, params, packed, W.z, R, W, st));
  ARDAE_TRY(launch_vae_loss(0, W.logit, nullptr, x, W.z, R, nz, 784, P.zd, beta, 0, 0.f, nullptr, W.rec_row, W.pri_row, nullptr, nullptr, nullptr, st));
  return launch_vae_loss_finalize(W.rec_row, W.pri_row, R, beta, losses, st);
}

int conv_model_vae_backward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B,
                            int nz, float beta, float dloss, const float* dz_extra, float* workspace, size_t wsf, float* grads,
                            float grads_beta, hipStream_t st) {
  const ConvLayout P(d);
  const ConvPacked K(P);
  Bump ws(workspace, wsf);
  ConvWs W;
  carve(P, ws, B, nz, 1, W);
  const int R = B * nz, act = P.act;
  const float gscale = dloss / (float)R;
  ARDAE_TRY(launch_vae_loss(0, W.logit, nullptr, x, W.z, R, nz, 784, P.zd, beta, 1, gscale, dz_extra, W.rec_row, W.pri_row, W.dlogit, nullptr,
                            W.dzq, st));
  ARDAE_TRY(conv_decoder_bwd(P, K, packed, W, R, st));
  // ---- sampler backward
  { LinArgs A{}; A.S = W.t1; A.ldS = 800; A.Y = W.dt1; A.ldY = 800;
    ARDAE_TRY(lin1(EPI_DACT, act, R, 800, W.dz, P.zd, P.zd, packed + K.fc5_b, A, st)); }
  ARDAE_TRY(launch_segment_sum(W.dt1, 800, B, nz, 800, 1.0f, W.drb, 800, st));
  { LinArgs A{}; A.Y = W.dinp; A.ldY = 512;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, 512, W.drb, 800, 800, packed + K.fc4i_b, A, st)); }
  ARDAE_TRY(trunk_bwd(K.conv_b, packed, W.dinp, W.dinp_t, W.hcv, W.dh3, W.dcols3, W.dh2, W.dcols2, W.dh1, B, act, st));
  // ---- weight gradients (order must match wgrad_scratch)
  std::vector<int> splits;
  wgrad_scratch(P, B, R, &splits);
  std::vector<WgradProblem> probs;
  auto push = [&](int M, int O, int I, const float* G, const float* X, float* out, int ldout, float* out_bias) {
    WgradProblem p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.O = O; p.I = I; p.npairs = 1;
    p.G[0] = G; p.ldG[0] = O; p.X[0] = X; p.ldX[0] = I;
    p.bias_pair = out_bias ? 0 : -1;
    p.splits = splits[probs.size()];
    p.partial = ws.take((size_t)p.splits * O * I);
    p.partial_vec = ws.take((size_t)p.splits * 2 * O);
    p.out = out; p.ldout = ldout; p.out_bias = out_bias; p.beta = grads_beta;
    probs.push_back(p);
  };
  conv_decoder_wgrads(P, W, R, grads, push);
  push(R, P.zd, 800, W.dz, W.t1, grads + P.fc5.w, 800, grads + P.fc5.b);
  push(R, 800, P.nd, W.dt1, noise, grads + P.fc4.w + 512, 512 + P.nd, grads + P.fc4.b);
  push(B, 800, 512, W.drb, W.inp, grads + P.fc4.w, 512 + P.nd, nullptr);
  push(B * 16, 32, 800, W.dh3, W.cols[2], grads + P.conv[2].w, 800, grads + P.conv[2].b);
  push(B * 49, 32, 400, W.dh2, W.cols[1], grads + P.conv[1].w, 400, grads + P.conv[1].b);
  push(B * 196, 16, 25, W.dh1, W.cols[0], grads + P.conv[0].w, 25, grads + P.conv[0].b);
  ARDAE_CHECK_ARG(ws.ok, "conv_model_vae_backward: workspace too small");
  return launch_wgrad_batch(probs.data(), (int)probs.size(), st);
}


// =====================================================================================================================
// MNISTConvAuxIPVAE (`--model auxconv`, kind == 4): models/ivae/auxconv.py:48-126 - the hierarchical sampler of csrc/auxmodel.hip with
// conv trunks in place of the MLPs (models/vae/auxconv.py:32-140) and ConvIPVAE's decoder (models/vae/conv.py:79-136):
//   per image:   h3a = trunk_a(2x-1); h4a = act(Fa h3a + fa); mu0 = M0 h4a + m0; lv0 = L0 h4a + l0;  h3 = trunk_e(2x-1); rb = Fi h3 + f
//   per sample:  z0 = mu0[b] + exp(lv0[b]/2) eps0;  h4 = act(Fn z0 + rb[b]);  mu = M h4 + m; lv = L h4 + l;  z = mu + exp(lv/2) eps
// Noise layout as for kind 3: one [R, noise_dim + z_dim] tensor per sampler call.  hidden1a context = cat(h4a, h4) [B, 1600].
// =====================================================================================================================
namespace {

struct AuxConvLayout {
  int nd, zd, act;
  Lin aconv[3], afc, mean0, logvar0, econv[3], efc, mean, logvar;
  ConvLayout dec;      // dfc / dcv only
  size_t total;
  explicit AuxConvLayout(const ardae_model_desc& d) : nd(d.noise_dim), zd(d.z_dim), act(d.act) {
    size_t off = 0;
    auto add = [&](Lin& l, int out, int in, int nbias) { l.out = out; l.in = in; l.w = off; off += (size_t)out * in; l.b = off; off += nbias; };
    add(aconv[0], 16, 25, 16); add(aconv[1], 32, 400, 32); add(aconv[2], 32, 800, 32);
    add(afc, 800, 512, 800); add(mean0, nd, 800, nd); add(logvar0, nd, 800, nd);
    add(econv[0], 16, 25, 16); add(econv[1], 32, 400, 32); add(econv[2], 32, 800, 32);
    add(efc, 800, 512 + nd, 800); add(mean, zd, 800, zd); add(logvar, zd, 800, zd);
    dec.nd = nd; dec.zd = zd; dec.act = act;
    add(dec.dfc[0], 300, zd, 300); add(dec.dfc[1], 512, 300, 512);
    add(dec.dcv[0], 32, 32 * 25, 32); add(dec.dcv[1], 32, 16 * 25, 16); add(dec.dcv[2], 16, 25, 1);
    total = off;
  }
};

struct AuxConvPacked {
  size_t aconv_f[3], aconv_b[3], afc_f, afc_b, mean0_f, mean0_b, logvar0_f, logvar0_b, econv_f[3], econv_b[3], efci_f, efci_b, efcn_f, efcn_b,
      mean_f, mean_b, logvar_f, logvar_b, total;
  ConvPacked dec;
  explicit AuxConvPacked(const AuxConvLayout& P) {
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += al64(n); return o; };
    for (int i = 0; i < 3; ++i) { aconv_f[i] = take(packed_floats(P.aconv[i].out, P.aconv[i].in)); aconv_b[i] = take(packed_floats(P.aconv[i].in, P.aconv[i].out)); }
    afc_f = take(packed_floats(800, 512)); afc_b = take(packed_floats(512, 800));
    mean0_f = take(packed_floats(P.nd, 800)); mean0_b = take(packed_floats(800, P.nd));
    logvar0_f = take(packed_floats(P.nd, 800)); logvar0_b = take(packed_floats(800, P.nd));
    for (int i = 0; i < 3; ++i) { econv_f[i] = take(packed_floats(P.econv[i].out, P.econv[i].in)); econv_b[i] = take(packed_floats(P.econv[i].in, P.econv[i].out)); }
    efci_f = take(packed_floats(800, 512)); efci_b = take(packed_floats(512, 800));
    efcn_f = take(packed_floats(800, P.nd)); efcn_b = take(packed_floats(P.nd, 800));
    mean_f = take(packed_floats(P.zd, 800)); mean_b = take(packed_floats(800, P.zd));
    logvar_f = take(packed_floats(P.zd, 800)); logvar_b = take(packed_floats(800, P.zd));
    for (int i = 0; i < 2; ++i) { dec.dfc_f[i] = take(packed_floats(P.dec.dfc[i].out, P.dec.dfc[i].in)); dec.dfc_b[i] = take(packed_floats(P.dec.dfc[i].in, P.dec.dfc[i].out)); }
    for (int i = 0; i < 3; ++i) { dec.dcv_f[i] = take(packed_floats(P.dec.dcv[i].in, P.dec.dcv[i].out)); dec.dcv_b[i] = take(packed_floats(P.dec.dcv[i].out, P.dec.dcv[i].in)); }
    total = off;
  }
};

struct AuxConvWs {
  ConvWs D;    // x2, decoder buffers, z, t1 (= h4), dzq / dz, ones, rec_row / pri_row and the decoder's backward buffers
  float *acols[3], *ahcv[3], *ainp, *h4a, *mu0, *lv0, *z0, *ecols[3], *ehcv[3], *einp, *rb, *mu, *lv, *zero;
  // backward
  float *dlv, *dt1, *dz0, *dlv0r, *drb, *dmu0, *dlv0, *dh4a, *dinp_a, *dinp_e, *dinp_t;
  float *dh3[2], *dcols3[2], *dh2[2], *dcols2[2], *dh1[2];    // [0] = aux trunk, [1] = encoder trunk
};

void aux_carve(const AuxConvLayout& P, Bump& ws, int B, int nz, int mode, AuxConvWs& W) {
  const size_t R = (size_t)B * nz;
  ConvWs& D = W.D;
  D.x2 = ws.take((size_t)B * 784);
  for (int i = 0; i < 3; ++i) {
    W.acols[i] = ws.take((size_t)B * EH[i + 1] * EH[i + 1] * ECH[i] * 25); W.ahcv[i] = ws.take((size_t)B * EH[i + 1] * EH[i + 1] * ECH[i + 1]);
    W.ecols[i] = ws.take((size_t)B * EH[i + 1] * EH[i + 1] * ECH[i] * 25); W.ehcv[i] = ws.take((size_t)B * EH[i + 1] * EH[i + 1] * ECH[i + 1]);
  }
  W.ainp = ws.take((size_t)B * 512); W.h4a = ws.take((size_t)B * 800); W.mu0 = ws.take((size_t)B * P.nd); W.lv0 = ws.take((size_t)B * P.nd);
  W.einp = ws.take((size_t)B * 512); W.rb = ws.take((size_t)B * 800);
  W.z0 = ws.take(R * P.nd); D.t1 = ws.take(R * 800); W.mu = ws.take(R * P.zd); W.lv = ws.take(R * P.zd); D.z = ws.take(R * P.zd);
  W.zero = ws.take(R * (P.nd + P.zd));
  if (mode == 0) return;
  D.d1 = ws.take(R * 300); D.d2 = ws.take(R * 512); D.g0 = ws.take(R * 512);
  D.c1 = ws.take(R * 16 * 800); D.u1 = ws.take(R * 64 * 32);
  D.c2 = ws.take(R * 64 * 400); D.u2 = ws.take(R * 225 * 16);
  D.c3 = ws.take(R * 225 * 25); D.logit = ws.take(R * 784);
  D.rec_row = ws.take(R); D.pri_row = ws.take(R);
  if (mode == 2) return;
  D.dlogit = ws.take(R * 784); D.dc3 = ws.take(R * 225 * 25); D.dp2 = ws.take(R * 225 * 16);
  D.dc2 = ws.take(R * 64 * 400); D.dp1 = ws.take(R * 64 * 32); D.dc1 = ws.take(R * 16 * 800);
  D.dg0 = ws.take(R * 512); D.dd2 = ws.take(R * 512); D.dd1 = ws.take(R * 300);
  D.dzq = ws.take(R * P.zd); D.dz = ws.take(R * P.zd); D.ones = ws.take(R * 784);
  W.dlv = ws.take(R * P.zd); W.dt1 = ws.take(R * 800); W.dz0 = ws.take(R * P.nd); W.dlv0r = ws.take(R * P.nd);
  W.drb = ws.take((size_t)B * 800); W.dmu0 = ws.take((size_t)B * P.nd); W.dlv0 = ws.take((size_t)B * P.nd); W.dh4a = ws.take((size_t)B * 800);
  W.dinp_a = ws.take((size_t)B * 512); W.dinp_e = ws.take((size_t)B * 512); W.dinp_t = ws.take((size_t)B * 512);
  for (int k = 0; k < 2; ++k) {
    W.dh3[k] = ws.take((size_t)B * 512); W.dcols3[k] = ws.take((size_t)B * 16 * 800); W.dh2[k] = ws.take((size_t)B * 49 * 32);
    W.dcols2[k] = ws.take((size_t)B * 49 * 400); W.dh1[k] = ws.take((size_t)B * 196 * 16);
  }
}

constexpr int AUX_N_WGRAD = 21;   // 8 decoder + 13 sampler problems, launched as two batches (20 per batch at most)

size_t aux_wgrad_scratch(const AuxConvLayout& P, int B, int R, std::vector<int>* out) {
  std::vector<int> sp; size_t tot = 0;
  auto one = [&](int M, int O, int I) { const int s = wgrad_splits(M, O, I, AUX_N_WGRAD / 2); sp.push_back(s); tot += al64((size_t)s * O * I) + al64((size_t)s * 2 * O); };
  one(R * 225, 16, 25); one(R * 784, 1, 1); one(R * 64, 32, 400); one(R * 225, 16, 1); one(R * 16, 32, 800); one(R * 64, 32, 1);
  one(R, 512, 300); one(R, 300, P.zd);                                              // decoder (as ConvIPVAE)
  one(R, P.zd, 800); one(R, P.zd, 800); one(R, 800, P.nd); one(B, 800, 512);       // mean, logvar, fc z0 half (+bias), fc image half
  one(B * 16, 32, 800); one(B * 49, 32, 400); one(B * 196, 16, 25);                // encoder trunk
  one(B, P.nd, 800); one(B, P.nd, 800); one(B, 800, 512);                          // mean0, logvar0, aux fc
  one(B * 16, 32, 800); one(B * 49, 32, 400); one(B * 196, 16, 25);                // aux trunk
  if (out) *out = sp;
  return tot;
}

size_t aux_workspace(const AuxConvLayout& P, int B, int nz, int mode) {
  Bump b(nullptr, ~size_t(0));
  AuxConvWs W;
  aux_carve(P, b, B, nz, mode, W);
  size_t t = b.off;
  if (mode == 1) t += aux_wgrad_scratch(P, B, B * nz, nullptr);
  return t;
}

// keep_hidden = false (forward-only encodes of the cDAE phase): the [R, 800] hidden rows need not exist
int aux_sampler_fwd(const AuxConvLayout& P, const AuxConvPacked& K, const float* params, const float* packed, const float* x, const float* noise,
                    int B, int nz, AuxConvWs& W, bool keep_hidden, hipStream_t st) {
  const int R = B * nz, act = P.act, ldn = P.nd + P.zd;
  ARDAE_TRY(launch_affine(x, (int64_t)B * 784, 2.f, -1.f, W.D.x2, st));
  ARDAE_TRY(trunk_fwd(P.aconv, K.aconv_f, params, packed, W.D.x2, W.acols, W.ahcv, W.ainp, B, act, st));
  { LinArgs A{}; A.bias = params + P.afc.b; A.Y = W.h4a; A.ldY = 800;
    ARDAE_TRY(lin1(EPI_ACT, act, B, 800, W.ainp, 512, 512, packed + K.afc_f, A, st)); }
  { LinArgs A{}; A.bias = params + P.mean0.b; A.Y = W.mu0; A.ldY = P.nd;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, P.nd, W.h4a, 800, 800, packed + K.mean0_f, A, st)); }
  { LinArgs A{}; A.bias = params + P.logvar0.b; A.Y = W.lv0; A.ldY = P.nd;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, P.nd, W.h4a, 800, 800, packed + K.logvar0_f, A, st)); }
  ARDAE_TRY(launch_reparam_fwd(W.mu0, W.lv0, noise, ldn, R, P.nd, nz, W.z0, st));
  ARDAE_TRY(trunk_fwd(P.econv, K.econv_f, params, packed, W.D.x2, W.ecols, W.ehcv, W.einp, B, act, st));
  { LinArgs A{}; A.bias = params + P.efc.b; A.Y = W.rb; A.ldY = 800;               // image half of the encoder's fc, once per image
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, 800, W.einp, 512, 512, packed + K.efci_f, A, st)); }
  if (!keep_hidden) {    // z0 -> 800 -> (mean | logvar) in one launch, hidden rows on chip (linear_shortk.hip::sampler_tail_kernel)
    LinArgs A{}; A.M = R; A.Nout = 800; A.act = act; A.nsrc = 1; A.rowbias = W.rb; A.rowbias_ld = 800; A.rows_per_group = nz;
    A.src[0].x = W.z0; A.src[0].ld = P.nd; A.src[0].K = P.nd; A.src[0].wp = packed + K.efcn_f;
    if (sampler_tail_eligible(A, P.zd)) {
      ARDAE_TRY(launch_sampler_tail(A, packed + K.mean_f, params + P.mean.b, W.mu, P.zd, P.zd, st, packed + K.logvar_f, params + P.logvar.b, W.lv, P.zd));
      return launch_reparam_fwd(W.mu, W.lv, noise + P.nd, ldn, R, P.zd, 1, W.D.z, st);
    }
  }
  { LinArgs A{}; A.rowbias = W.rb; A.rowbias_ld = 800; A.rows_per_group = nz; A.Y = W.D.t1; A.ldY = 800;
    ARDAE_TRY(lin1(EPI_ACT, act, R, 800, W.z0, P.nd, P.nd, packed + K.efcn_f, A, st)); }
  { LinArgs A{}; A.bias = params + P.mean.b; A.Y = W.mu; A.ldY = P.zd;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.zd, W.D.t1, 800, 800, packed + K.mean_f, A, st)); }
  { LinArgs A{}; A.bias = params + P.logvar.b; A.Y = W.lv; A.ldY = P.zd;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.zd, W.D.t1, 800, 800, packed + K.logvar_f, A, st)); }
  return launch_reparam_fwd(W.mu, W.lv, noise + P.nd, ldn, R, P.zd, 1, W.D.z, st);
}

}  // namespace

size_t auxconv_model_param_floats(const ardae_model_desc& d) { return AuxConvLayout(d).total; }
size_t auxconv_model_packed_floats(const ardae_model_desc& d) { return AuxConvPacked(AuxConvLayout(d)).total; }
size_t auxconv_model_workspace_floats(const ardae_model_desc& d, int B, int nz, int mode) { return aux_workspace(AuxConvLayout(d), B, nz, mode == 3 ? 0 : mode); }

int auxconv_model_pack(const ardae_model_desc& d, const float* params, float* packed, hipStream_t st) {
  std::vector<PackItem> pack_items__;
  const AuxConvLayout P(d);
  const AuxConvPacked K(P);
  auto both = [&](const Lin& l, size_t f, size_t b) {
    PACK_PUSH(params + l.w, l.in, l.out, l.in, false, packed + f);
    PACK_PUSH(params + l.w, l.in, l.in, l.out, true, packed + b);
  };
  for (int i = 0; i < 3; ++i) { both(P.aconv[i], K.aconv_f[i], K.aconv_b[i]); both(P.econv[i], K.econv_f[i], K.econv_b[i]); }
  both(P.afc, K.afc_f, K.afc_b); both(P.mean0, K.mean0_f, K.mean0_b); both(P.logvar0, K.logvar0_f, K.logvar0_b);
  const int lde = 512 + P.nd;
  PACK_PUSH(params + P.efc.w, lde, 800, 512, false, packed + K.efci_f);
  PACK_PUSH(params + P.efc.w, lde, 512, 800, true, packed + K.efci_b);
  PACK_PUSH(params + P.efc.w + 512, lde, 800, P.nd, false, packed + K.efcn_f);
  PACK_PUSH(params + P.efc.w + 512, lde, P.nd, 800, true, packed + K.efcn_b);
  both(P.mean, K.mean_f, K.mean_b); both(P.logvar, K.logvar_f, K.logvar_b);
  for (int i = 0; i < 2; ++i) both(P.dec.dfc[i], K.dec.dfc_f[i], K.dec.dfc_b[i]);
  for (int i = 0; i < 3; ++i) {   // ConvTranspose2d weight [in, out*25] (see conv_model_pack)
    PACK_PUSH(params + P.dec.dcv[i].w, P.dec.dcv[i].in, P.dec.dcv[i].in, P.dec.dcv[i].out, true, packed + K.dec.dcv_f[i]);
    PACK_PUSH(params + P.dec.dcv[i].w, P.dec.dcv[i].in, P.dec.dcv[i].out, P.dec.dcv[i].in, false, packed + K.dec.dcv_b[i]);
  }
  PACK_FLUSH(st);
  return 0;
}

int auxconv_model_encode(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                         float* workspace, size_t wsf, float* z_out, float* hidden_out, hipStream_t st) {
  const AuxConvLayout P(d);
  const AuxConvPacked K(P);
  Bump ws(workspace, wsf);
  AuxConvWs W;
  aux_carve(P, ws, B, nz, 0, W);
  ARDAE_CHECK_ARG(ws.ok, "auxconv_model_encode: workspace too small");
  const float* nptr = noise;
  if (!noise) {
    ARDAE_TRY(launch_fill(W.zero, (size_t)B * nz * (P.nd + P.zd), 0.f, st));
    nptr = W.zero;
  }
  ARDAE_TRY(aux_sampler_fwd(P, K, params, packed, x, nptr, B, nz, W, hidden_out != nullptr, st));
  if (z_out) ARDAE_TRY(launch_copy(W.D.z, (size_t)B * nz * P.zd, z_out, st));
  if (hidden_out) {
    ARDAE_CHECK_ARG(nz == 1, "auxconv_model_encode: the hidden context is defined for nz == 1");
    ARDAE_TRY(launch_copy2d(W.h4a, 800, hidden_out, 1600, B, 800, st));
    ARDAE_TRY(launch_copy2d(W.D.t1, 800, hidden_out + 800, 1600, B, 800, st));
  }
  return 0;
}

int auxconv_model_decode(const ardae_model_desc& d, const float* params, const float* packed, const float* z, int R, float* workspace, size_t wsf,
                         float* out0, hipStream_t st) {
  const AuxConvLayout P(d);
  const AuxConvPacked K(P);
  Bump ws(workspace, wsf);
  AuxConvWs W;
  aux_carve(P, ws, R, 1, 2, W);
  ARDAE_CHECK_ARG(ws.ok, "auxconv_model_decode: workspace too small");
  ARDAE_TRY(conv_decode_fwd(P.dec, K.dec, params, packed, z, R, W.D, st));
  ARDAE_TRY(launch_copy(W.D.logit, (size_t)R * 784, out0, st));
  return 0;
}

int auxconv_model_vae_forward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B,
                              int nz, float beta, float* workspace, size_t wsf, float* z_out, float* losses, hipStream_t st) {
  const AuxConvLayout P(d);
  const AuxConvPacked K(P);
  Bump ws(workspace, wsf);
  AuxConvWs W;
  aux_carve(P, ws, B, nz, 1, W);
  ARDAE_CHECK_ARG(ws.ok, "auxconv_model_vae_forward: workspace too small");
  const int R = B * nz;
  ARDAE_TRY(aux_sampler_fwd(P, K, params, packed, x, noise, B, nz, W, true, st));
  ARDAE_TRY(launch_copy(W.D.z, (size_t)R * P.zd, z_out, st));
  ARDAE_TRY(conv_decode_fwd(P.dec, K.dec, params, packed, W.D.z, R, W.D, st));
  ARDAE_TRY(launch_vae_loss(0, W.D.logit, nullptr, x, W.D.z, R, nz, 784, P.zd, beta, 0, 0.f, nullptr, W.D.rec_row, W.D.pri_row, nullptr, nullptr, nullptr, st));
  return launch_vae_loss_finalize(W.D.rec_row, W.D.pri_row, R, beta, losses, st);
}

int auxconv_model_vae_backward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B,
                               int nz, float beta, float dloss, const float* dz_extra, float* workspace, size_t wsf, float* grads,
                               float grads_beta, hipStream_t st) {
  (void)noise;
  const AuxConvLayout P(d);
  const AuxConvPacked K(P);
  Bump ws(workspace, wsf);
  AuxConvWs W;
  aux_carve(P, ws, B, nz, 1, W);
  ConvWs& D = W.D;
  const int R = B * nz, act = P.act;
  const float gscale = dloss / (float)R;
  ARDAE_TRY(launch_vae_loss(0, D.logit, nullptr, x, D.z, R, nz, 784, P.zd, beta, 1, gscale, dz_extra, D.rec_row, D.pri_row, D.dlogit, nullptr, D.dzq, st));
  ARDAE_TRY(conv_decoder_bwd(P.dec, K.dec, packed, D, R, st));
  // second reparameterisation and the encoder's fc
  ARDAE_TRY(launch_reparam_bwd(D.dz, D.z, W.mu, R, P.zd, 1, W.dlv, st));
  {
    LinArgs A{}; A.M = R; A.Nout = 800; A.nsrc = 2; A.act = act; A.S = D.t1; A.ldS = 800; A.Y = W.dt1; A.ldY = 800;
    A.src[0].x = D.dz; A.src[0].ld = P.zd; A.src[0].K = P.zd; A.src[0].wp = packed + K.mean_b;
    A.src[1].x = W.dlv; A.src[1].ld = P.zd; A.src[1].K = P.zd; A.src[1].wp = packed + K.logvar_b;
    ARDAE_TRY(launch_linear(A, EPI_DACT, st));
  }
  { LinArgs A{}; A.Y = W.dz0; A.ldY = P.nd;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.nd, W.dt1, 800, 800, packed + K.efcn_b, A, st)); }
  ARDAE_TRY(launch_segment_sum(W.dt1, 800, B, nz, 800, 1.0f, W.drb, 800, st));
  { LinArgs A{}; A.Y = W.dinp_e; A.ldY = 512;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, 512, W.drb, 800, 800, packed + K.efci_b, A, st)); }
  ARDAE_TRY(trunk_bwd(K.econv_b, packed, W.dinp_e, W.dinp_t, W.ehcv, W.dh3[1], W.dcols3[1], W.dh2[1], W.dcols2[1], W.dh1[1], B, act, st));
  // first reparameterisation, reduced over the nz samples of each image, and the aux encoder
  ARDAE_TRY(launch_reparam_bwd(W.dz0, W.z0, W.mu0, R, P.nd, nz, W.dlv0r, st));
  ARDAE_TRY(launch_segment_sum(W.dz0, P.nd, B, nz, P.nd, 1.0f, W.dmu0, P.nd, st));
  ARDAE_TRY(launch_segment_sum(W.dlv0r, P.nd, B, nz, P.nd, 1.0f, W.dlv0, P.nd, st));
  {
    LinArgs A{}; A.M = B; A.Nout = 800; A.nsrc = 2; A.act = act; A.S = W.h4a; A.ldS = 800; A.Y = W.dh4a; A.ldY = 800;
    A.src[0].x = W.dmu0; A.src[0].ld = P.nd; A.src[0].K = P.nd; A.src[0].wp = packed + K.mean0_b;
    A.src[1].x = W.dlv0; A.src[1].ld = P.nd; A.src[1].K = P.nd; A.src[1].wp = packed + K.logvar0_b;
    ARDAE_TRY(launch_linear(A, EPI_DACT, st));
  }
  { LinArgs A{}; A.Y = W.dinp_a; A.ldY = 512;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, 512, W.dh4a, 800, 800, packed + K.afc_b, A, st)); }
  ARDAE_TRY(trunk_bwd(K.aconv_b, packed, W.dinp_a, W.dinp_t, W.ahcv, W.dh3[0], W.dcols3[0], W.dh2[0], W.dcols2[0], W.dh1[0], B, act, st));
  // ---- weight gradients: 8 decoder + 14 sampler problems in two batches (order == aux_wgrad_scratch)
  std::vector<int> splits;
  aux_wgrad_scratch(P, B, R, &splits);
  std::vector<WgradProblem> probs;
  auto push = [&](int M, int O, int I, const float* G, const float* X, float* out, int ldout, float* out_bias) {
    WgradProblem p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.O = O; p.I = I; p.npairs = 1;
    p.G[0] = G; p.ldG[0] = O; p.X[0] = X; p.ldX[0] = I;
    p.bias_pair = out_bias ? 0 : -1;
    p.splits = splits[probs.size()];
    p.partial = ws.take((size_t)p.splits * O * I);
    p.partial_vec = ws.take((size_t)p.splits * 2 * O);
    p.out = out; p.ldout = ldout; p.out_bias = out_bias; p.beta = grads_beta;
    probs.push_back(p);
  };
  conv_decoder_wgrads(P.dec, D, R, grads, push);
  push(R, P.zd, 800, D.dz, D.t1, grads + P.mean.w, 800, grads + P.mean.b);
  push(R, P.zd, 800, W.dlv, D.t1, grads + P.logvar.w, 800, grads + P.logvar.b);
  push(R, 800, P.nd, W.dt1, W.z0, grads + P.efc.w + 512, 512 + P.nd, grads + P.efc.b);
  push(B, 800, 512, W.drb, W.einp, grads + P.efc.w, 512 + P.nd, nullptr);
  push(B * 16, 32, 800, W.dh3[1], W.ecols[2], grads + P.econv[2].w, 800, grads + P.econv[2].b);
  push(B * 49, 32, 400, W.dh2[1], W.ecols[1], grads + P.econv[1].w, 400, grads + P.econv[1].b);
  push(B * 196, 16, 25, W.dh1[1], W.ecols[0], grads + P.econv[0].w, 25, grads + P.econv[0].b);
  push(B, P.nd, 800, W.dmu0, W.h4a, grads + P.mean0.w, 800, grads + P.mean0.b);
  push(B, P.nd, 800, W.dlv0, W.h4a, grads + P.logvar0.w, 800, grads + P.logvar0.b);
  push(B, 800, 512, W.dh4a, W.ainp, grads + P.afc.w, 512, grads + P.afc.b);
  push(B * 16, 32, 800, W.dh3[0], W.acols[2], grads + P.aconv[2].w, 800, grads + P.aconv[2].b);
  push(B * 49, 32, 400, W.dh2[0], W.acols[1], grads + P.aconv[1].w, 400, grads + P.aconv[1].b);
  push(B * 196, 16, 25, W.dh1[0], W.acols[0], grads + P.aconv[0].w, 25, grads + P.aconv[0].b);
  ARDAE_CHECK_ARG(ws.ok, "auxconv_model_vae_backward: workspace too small (%zu floats needed so far, %zu given, %zu problems)", ws.off, wsf, probs.size());
  ARDAE_CHECK_ARG((int)probs.size() == AUX_N_WGRAD, "auxconv_model_vae_backward: internal problem count");
  ARDAE_TRY(launch_wgrad_batch(probs.data(), AUX_N_WGRAD / 2, st));
  return launch_wgrad_batch(probs.data() + AUX_N_WGRAD / 2, AUX_N_WGRAD - AUX_N_WGRAD / 2, st);
}

}  // namespace ardae
