// Weight-normalised residual-conv implicit-posterior VAEs on gfx950 (ardae_model_desc.kind 5 and 6):
//   kind 5  ResConvIPVAE           models/ivae/resconv.py:53-245 (`--model resconvct-res`: do_center, enc_type 'res-wn-mlp')
//   kind 6  MNISTResConvAuxIPVAE   models/ivae/auxresconv.py:48-255 + models/vae/auxresconv.py:26-185 (`--model auxresconvct`)
// both with the decoder of models/vae/resconv.py:79-136.  The shipped "implicit resconv" / "hierarchical resconv" recipes
// (run_vae_dbmnist.sh) train them with ELU, an mlp-res cDAE, --std-scale 100.
//
// Every residual block (models/layers2.py:305-352, models/layers.py:66-85) is
//     out = W_h1 f(W_0h x + b_0h) + W_01 x + (b_h1 + b_01)           f = ELU inside ResConv2d, ReLU inside ResLinear
// with weight-normalised operators W = scale * direction / ||direction|| (layers2.py:73-83,255-265; ResMLP's operators skip the
// normalisation, layers.py:47-53 with norm=False).  Here a block is TWO launches of the FP32-MFMA linear kernel: the inner
// operator, then the second operator and the skip as ONE two-source product accumulating into the same output (the concat-input
// form of ardae_linear), with 3x3 convolutions as im2col / col2im gathers around them (NHWC activations as [rows = b*H*W, C]).
// The effective weights are composed once per optimiser step (ardae_model_pack) into the packed buffer; the backward turns
// dL/dW into dL/d(direction), dL/d(scale) in one launch over all operators.  Everything per-image runs on B rows; only the
// sampler's tail runs on the B*nz Monte-Carlo rows, with the per-image half of its concat input hoisted into a row bias.
#include <vector>

#include "ardae_hip.h"
#include "auxmodel.h"
#include "common.h"
#include "elementwise.h"
#include "linear.h"
#include "resmodel.h"
#include "wgrad.h"

namespace ardae {
namespace {

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }
size_t al64(size_t n) { return (n + 63) & ~size_t(63); }

// ------------------------------------------------------------------------------------------------ kernels
// weff[o][:] = scale[o] * dir[o][:] * inv,  inv = norm ? 1 / ||dir[o][:]|| : 1   (one workgroup per output row)
// all operators of a model in ONE launch (a re-pack follows every optimiser step: ~50 launches of a few us each otherwise):
// grid (max O, items); item = one weight-normalised operator, or (dir == nullptr) a bias sum  out[i] = a[i] + b[i]
constexpr int WNC_MAX = 64;
struct WnComposeItem {
  const float* dir; const float* scale; float* weff; float* inv; int O, I, norm;   // operator: weff [O, I], inv [O]
  const float* add_a; const float* add_b;                                         // bias sum (dir == nullptr): weff[i] = add_a[i] + add_b[i], O entries
};
struct WnComposeBatch { int n; WnComposeItem it[WNC_MAX]; };
__global__ __launch_bounds__(256) void wn_compose_batch_kernel(const WnComposeBatch b) {
  __shared__ float red[256];
  const WnComposeItem& w = b.it[blockIdx.y];
  const int o = blockIdx.x, t = threadIdx.x;
  if (w.dir == nullptr) {       // bias sum
    const int i = o * 256 + t;
    if (i < w.O) w.weff[i] = w.add_a[i] + w.add_b[i];
    return;
  }
  if (o >= w.O) return;
  const float* d = w.dir + (size_t)o * w.I;
  float s = 0.f;
  for (int i = t; i < w.I; i += 256) s += d[i] * d[i];
  red[t] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (t < k) red[t] += red[t + k];
    __syncthreads();
  }
  const float inv = w.norm ? 1.f / sqrtf(red[0]) : 1.f;
  const float f = w.scale[o] * inv;
  for (int i = t; i < w.I; i += 256) w.weff[(size_t)o * w.I + i] = f * d[i];
  if (t == 0) w.inv[o] = inv;
}

// dL/dW [O, I] -> dL/ddir, dL/dscale (and the bias gradients, plain copies) for up to WNB_MAX operators in one launch
constexpr int WNB_MAX = 48;
struct WnBwdItem {
  const float* dW; const float* dir; const float* scale; const float* inv; const float* db;   // db: [O] bias gradient (scratch)
  float* gdir; float* gscale; float* gbias;
  int O, I, norm;
};
struct WnBwdBatch { int n; float beta; WnBwdItem it[WNB_MAX]; };
__global__ __launch_bounds__(256) void wn_backward_kernel(const WnBwdBatch b) {
  __shared__ float red[256];
  const WnBwdItem& w = b.it[blockIdx.y];
  const int t = threadIdx.x;
  for (int o = blockIdx.x; o < w.O; o += gridDim.x) {
    const float* dW = w.dW + (size_t)o * w.I;
    const float* d = w.dir + (size_t)o * w.I;
    float s = 0.f;
    for (int i = t; i < w.I; i += 256) s += dW[i] * d[i];
    __syncthreads();
    red[t] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
      if (t < k) red[t] += red[t + k];
      __syncthreads();
    }
    const float s1 = red[0], inv = w.inv[o], sc = w.scale[o];
    const float f = sc * inv, c = w.norm ? inv * inv * s1 : 0.f;
    float* gd = w.gdir + (size_t)o * w.I;
    for (int i = t; i < w.I; i += 256) {
      const float v = f * (dW[i] - d[i] * c);
      gd[i] = b.beta != 0.f ? b.beta * gd[i] + v : v;
    }
    if (t == 0) {
      w.gscale[o] = (b.beta != 0.f ? b.beta * w.gscale[o] : 0.f) + s1 * inv;
      w.gbias[o] = (b.beta != 0.f ? b.beta * w.gbias[o] : 0.f) + w.db[o];
    }
  }
}

// cols[(b*OH+oh)*OW+ow][c*9+kh*3+kw] = x[b][s*oh-1+kh][s*ow-1+kw][c]  (0 outside); x NHWC [B,H,W,C]: 3x3, padding 1, stride s
__global__ void im2col3_kernel(const float* __restrict__ x, int H, int W, int C, int OH, int OW, int s, float* __restrict__ cols, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int J = C * 9;
  const int j = (int)(e % J);
  const int64_t row = e / J;
  const int ow = (int)(row % OW), oh = (int)((row / OW) % OH);
  const int64_t b = row / ((int64_t)OW * OH);
  const int c = j / 9, kk = j - c * 9, kh = kk / 3, kw = kk - kh * 3;
  const int h = s * oh - 1 + kh, w = s * ow - 1 + kw;
  cols[e] = (h >= 0 && h < H && w >= 0 && w < W) ? x[((b * H + h) * W + w) * C + c] : 0.f;
}

// transpose of im2col3 as a gather: dx[b][h][w][c] = sum_{kh,kw: (h+1-kh) % s == 0, oh = (h+1-kh)/s in range} dcols[(b,oh,ow)][c*9+kh*3+kw]
// mulS: multiply by act'(S[e]) on the way out (the activation in front of the convolution: one launch less per block)
__global__ void col2im3_kernel(const float* __restrict__ dcols, int H, int W, int C, int OH, int OW, int s, float* __restrict__ dx, int64_t total,
                               const float* __restrict__ mulS, int act) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const int64_t pix = e / C;
  const int w = (int)(pix % W), h = (int)((pix / W) % H);
  const int64_t b = pix / ((int64_t)W * H);
  float v = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int th = h + 1 - kh;
    if (th < 0 || th % s) continue;
    const int oh = th / s;
    if (oh >= OH) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int tw = w + 1 - kw;
      if (tw < 0 || tw % s) continue;
      const int ow = tw / s;
      if (ow >= OW) continue;
      v += dcols[((b * OH + oh) * OW + ow) * (int64_t)(C * 9) + c * 9 + kh * 3 + kw];
    }
  }
  dx[e] = mulS ? v * act_d1_rt(act, mulS[e]) : v;
}

__global__ void act_inplace_kernel(float* __restrict__ x, int act, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) x[e] = act_fwd_rt(act, x[e]);
}
// y = x * act'(S)  (S = saved post-activation)
__global__ void mul_dact_kernel(const float* __restrict__ x, const float* __restrict__ S, int act, float* __restrict__ y, int64_t n) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) y[e] = x[e] * act_d1_rt(act, S[e]);
}

// bilinear x2 upsampling, align_corners=True (nn.Upsample in models/vae/resconv.py:96-106), NHWC; src = o (IH-1)/(OH-1)
__device__ __forceinline__ void up_src(int o, int IH, int OH, int& i0, int& i1, float& f) {
  const float s = OH > 1 ? (float)o * (float)(IH - 1) / (float)(OH - 1) : 0.f;
  i0 = (int)floorf(s);
  if (i0 > IH - 1) i0 = IH - 1;
  i1 = i0 + 1 < IH ? i0 + 1 : IH - 1;
  f = s - (float)i0;
}
__global__ void upsample2_fwd_kernel(const float* __restrict__ x, int IH, int C, float* __restrict__ y, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int OH = 2 * IH;
  const int c = (int)(e % C);
  const int64_t pix = e / C;
  const int ow = (int)(pix % OH), oh = (int)((pix / OH) % OH);
  const int64_t b = pix / ((int64_t)OH * OH);
  int h0, h1, w0, w1; float fh, fw;
  up_src(oh, IH, OH, h0, h1, fh); up_src(ow, IH, OH, w0, w1, fw);
  const float* xb = x + b * (int64_t)IH * IH * C + c;
  const float v00 = xb[(h0 * IH + w0) * C], v01 = xb[(h0 * IH + w1) * C], v10 = xb[(h1 * IH + w0) * C], v11 = xb[(h1 * IH + w1) * C];
  y[e] = (1.f - fh) * ((1.f - fw) * v00 + fw * v01) + fh * ((1.f - fw) * v10 + fw * v11);
}
// transpose as a gather over the (few) output pixels that read input pixel (ih, iw)
__global__ void upsample2_bwd_kernel(const float* __restrict__ dy, int IH, int C, float* __restrict__ dx, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int OH = 2 * IH;
  const int c = (int)(e % C);
  const int64_t pix = e / C;
  const int iw = (int)(pix % IH), ih = (int)((pix / IH) % IH);
  const int64_t b = pix / ((int64_t)IH * IH);
  // outputs o with floor(src(o)) in {i-1, i}: src(o) in (i-1, i+1)  ->  o in ((i-1) r, (i+1) r), r = (OH-1)/(IH-1)
  const float r = IH > 1 ? (float)(OH - 1) / (float)(IH - 1) : 1.f;
  int lo_h = (int)floorf((float)(ih - 1) * r) - 1, hi_h = (int)ceilf((float)(ih + 1) * r) + 1;
  int lo_w = (int)floorf((float)(iw - 1) * r) - 1, hi_w = (int)ceilf((float)(iw + 1) * r) + 1;
  if (lo_h < 0) lo_h = 0; if (lo_w < 0) lo_w = 0;
  if (hi_h > OH - 1) hi_h = OH - 1; if (hi_w > OH - 1) hi_w = OH - 1;
  const float* dyb = dy + b * (int64_t)OH * OH * C + c;
  float v = 0.f;
  for (int oh = lo_h; oh <= hi_h; ++oh) {
    int h0, h1; float fh;
    up_src(oh, IH, OH, h0, h1, fh);
    const float wh = (h0 == ih ? 1.f - fh : 0.f) + (h1 == ih ? fh : 0.f);
    if (wh == 0.f) continue;
    for (int ow = lo_w; ow <= hi_w; ++ow) {
      int w0, w1; float fw;
      up_src(ow, IH, OH, w0, w1, fw);
      const float ww = (w0 == iw ? 1.f - fw : 0.f) + (w1 == iw ? fw : 0.f);
      if (ww != 0.f) v += wh * ww * dyb[(oh * OH + ow) * C];
    }
  }
  dx[e] = v;
}
// [B, HS, HS, C] -> [B, HD, HD, C]: crop (HD < HS: slicer[:, :, :-1, :-1]) or zero-pad (HD > HS: its transpose)
__global__ void crop_pad_kernel(const float* __restrict__ x, int HS, int HD, int C, float* __restrict__ y, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const int64_t pix = e / C;
  const int w = (int)(pix % HD), h = (int)((pix / HD) % HD);
  const int64_t b = pix / ((int64_t)HD * HD);
  y[e] = (h < HS && w < HS) ? x[((b * HS + h) * HS + w) * C + c] : 0.f;
}
// [B, HW, C] (NHWC rows) <-> [B, C*HW] (PyTorch's .view(B, -1) of NCHW)
__global__ void nhwc_nchw_kernel(const float* __restrict__ in, int HW, int C, float* __restrict__ out, int64_t total, int to_nhwc) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const int hw = (int)((e / C) % HW);
  const int64_t b = e / ((int64_t)C * HW);
  const int64_t nchw = (b * C + c) * HW + hw;
  if (to_nhwc) out[e] = in[nchw]; else out[nchw] = in[e];
}
// NormalDistribution.clip_logvar 'spm4' (models/reparam.py:30-31): y = softplus(x + 4) - 4; backward: dx = dy * sigmoid(x + 4)
// identity != 0 (the clipped class of ivae/auxresconv2.py:71-72 builds its heads WITHOUT the clip): y = x, dx = dy
__global__ void spm4_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ y, int64_t n, int identity) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  if (identity) { y[e] = dy ? dy[e] : x[e]; return; }
  const float u = x[e] + 4.f;
  if (dy) y[e] = dy[e] / (1.f + expf(-u));
  else y[e] = (u > 20.f ? u : log1pf(expf(u))) - 4.f;
}

#define RES_LAUNCH(kernel, n, ...)                                                            \
  do {                                                                                        \
    hipLaunchKernelGGL(kernel, dim3(nblk(n)), dim3(256), 0, st, __VA_ARGS__);                 \
    ARDAE_LAUNCH_CHECK();                                                                     \
  } while (0)

// ------------------------------------------------------------------------------------------------ layout
// plain: the operator is an nn.Linear (weight at `dir`, no scale, effective weight = weight): ResMLP(layer='linear')
struct WN { size_t dir, scale, bias; int O, I; bool norm; bool plain; };          // flat-parameter offsets; I = fan-in (C*9 for a conv)
struct Lin { size_t w, b; int out, in; };
struct Blk { WN a, h, s; bool conv; int Cin, Cout, stride, Hin, Hout; bool same; };   // a = *_0h, h = *_h1, s = *_01 (skip; absent when same: identity)
// One operator of ResConvIPVAE's sampler head `encode.fc` (models/ivae/resconv.py:101-116) on the B nz rows: a ResLinear (inner ReLU) or a
// plain Linear, reading [trunk output | noise] (concat: the first one) or its predecessor's output, with ELU behind it or not
struct HeadOp { bool res; Blk b; Lin l; int in, out; bool concat; int act; };
// desc.flags bits 1-3 of kind 5: which head (ivae_ardae.py:323-442)
enum { HEAD_RES_WN_MLP = 0, HEAD_MLP = 1, HEAD_RES_MLP = 2, HEAD_RES_WN_MLP_LIN = 3, HEAD_RES_MLP_LIN = 4 };

struct ResLayout {
  int kind, nd, zd, cdim, hdim;
  bool center;                   // do_center: the trunk sees 2x - 1 (desc.flags without ARDAE_MODEL_NO_CENTER)
  bool clipped;                  // kind 6 + ARDAE_MODEL_CLIPPED: MNISTResConvAuxIPVAEClipped (no 'spm4' clip, z0 keeps an unscaled eps0)
  std::vector<Blk> trunk, dec;   // trunk: 5 conv + ResLinear(512 -> cdim); dec: 2 ResLinear + 5 conv
  std::vector<HeadOp> head;      // kind 5: encode.fc (the shipped recipe: ResLinear(cdim + nd -> hdim), ELU, ResLinear(hdim -> zd), un-normalised WN operators)
  Lin mu0, lv0, efc, mu, lv;     // kind 6: aux_encode.reparam.{mean,logvar}_fn, encode.fc.0, encode.reparam.{mean,logvar}_fn
  size_t total = 0;

  explicit ResLayout(const ardae_model_desc& d)
      : kind(d.kind), nd(d.noise_dim), zd(d.z_dim), cdim(d.kind == 5 ? 512 : d.h_dim), hdim(d.h_dim), center(!(d.flags & ARDAE_MODEL_NO_CENTER)),
        clipped(d.kind == 6 && (d.flags & ARDAE_MODEL_CLIPPED)) {
    size_t off = 0;
    auto wn = [&](int O, int I, bool norm, bool plain = false) {
      WN w; w.O = O; w.I = I; w.norm = norm; w.plain = plain;
      w.dir = off; off += (size_t)O * I; w.scale = off; if (!plain) off += O; w.bias = off; off += O;
      return w;
    };
    auto block = [&](int Cout, int Cin, bool conv, int stride, int Hin, bool norm, bool plain = false, bool same = false) {
      Blk b; b.conv = conv; b.Cin = Cin; b.Cout = Cout; b.stride = stride; b.Hin = Hin; b.same = same;
      b.Hout = conv ? (Hin + 2 - 3) / stride + 1 : 1;
      const int I = conv ? Cin * 9 : Cin, Ih = conv ? Cout * 9 : Cout;
      b.a = wn(Cout, I, norm, plain); b.h = wn(Cout, Ih, norm, plain);
      if (same) { b.s = b.h; b.s.O = 0; } else b.s = wn(Cout, I, norm, plain);
      return b;
    };
    auto lin = [&](int out, int in) { Lin l; l.out = out; l.in = in; l.w = off; off += (size_t)out * in; l.b = off; off += out; return l; };
    trunk.push_back(block(16, 1, true, 2, 28, true));    // 28 -> 14
    trunk.push_back(block(16, 16, true, 1, 14, true));
    trunk.push_back(block(32, 16, true, 2, 14, true));   // 14 -> 7
    trunk.push_back(block(32, 32, true, 1, 7, true));
    trunk.push_back(block(32, 32, true, 2, 7, true));    // 7 -> 4
    trunk.push_back(block(cdim, 512, false, 1, 1, true));
    if (kind == 5) {
      // encode.fc in the reference's parameter order (models/ivae/resconv.py:101-116, models/layers.py:477-515,559-622)
      const int ht = (d.flags >> 1) & 7, nl = d.n_layers, cin = cdim + nd;
      auto res_op = [&](int out, int in, bool plain, int act, bool concat) {
        HeadOp o; o.res = true; o.b = block(out, in, false, 1, 1, false, plain, in == out); o.in = in; o.out = out; o.concat = concat; o.act = act;
        o.l = Lin{0, 0, 0, 0};
        head.push_back(o);
      };
      auto lin_op = [&](int out, int in, int act, bool concat) {
        HeadOp o; o.res = false; o.l = lin(out, in); o.in = in; o.out = out; o.concat = concat; o.act = act; o.b = Blk{};
        head.push_back(o);
      };
      if (ht == HEAD_MLP) {                       // MLP(cin -> hdim x nl -> zd), ELU after the hidden layers
        for (int i = 0; i < nl; ++i) lin_op(hdim, i == 0 ? cin : hdim, ACT_ELU, i == 0);
        lin_op(zd, hdim, ACT_NONE, false);
      } else if (ht == HEAD_RES_WN_MLP || ht == HEAD_RES_MLP) {   // ResMLP(cin -> hdim x nl -> zd): ResLinear blocks, ELU between them
        const bool plain = ht == HEAD_RES_MLP;
        for (int i = 0; i < nl; ++i) res_op(hdim, i == 0 ? cin : hdim, plain, ACT_ELU, i == 0);
        res_op(zd, hdim, plain, ACT_NONE, false);
      } else {                                    // Sequential(ResMLP(cin -> hdim x (nl - 1) -> hdim, ELU on its output), Linear(hdim -> zd))
        const bool plain = ht == HEAD_RES_MLP_LIN;
        for (int i = 0; i < nl - 1; ++i) res_op(hdim, i == 0 ? cin : hdim, plain, ACT_ELU, i == 0);
        res_op(hdim, nl - 1 == 0 ? cin : hdim, plain, ACT_ELU, nl - 1 == 0);
        lin_op(zd, hdim, ACT_NONE, false);
      }
    } else {
      mu0 = lin(nd, cdim); lv0 = lin(nd, cdim); efc = lin(cdim, cdim + nd); mu = lin(zd, cdim); lv = lin(zd, cdim);
    }
    dec.push_back(block(cdim, zd, false, 1, 1, true));
    dec.push_back(block(512, cdim, false, 1, 1, true));
    dec.push_back(block(32, 32, true, 1, 8, true));
    dec.push_back(block(32, 32, true, 1, 8, true));
    dec.push_back(block(16, 32, true, 1, 14, true));
    dec.push_back(block(16, 16, true, 1, 14, true));
    dec.push_back(block(1, 16, true, 1, 28, true));
    total = off;
  }
};

// offsets into the packed buffer
struct WNPk { size_t weff, inv, f, b; };
struct BlkPk { WNPk a, h, s; size_t bsum; size_t a_fi, a_fn, a_bi, s_fi, s_fn, s_bi; };   // *_fi / *_fn / *_bi: image half / noise half of a concat-input operator
struct LinPk { size_t f, b, fi, fn, bi, bn; };   // fi / fn: image / noise columns forward; bi / bn: their transposes
struct ResPacked {
  std::vector<BlkPk> trunk, dec;
  struct HeadPk { BlkPk k; LinPk lk; };
  std::vector<HeadPk> head;
  LinPk mu0, lv0, efc, mu, lv;
  size_t total = 0;
  explicit ResPacked(const ResLayout& P) {
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += al64(n); return o; };
    auto wn = [&](const WN& w) { WNPk k; k.weff = take((size_t)w.O * w.I); k.inv = take(w.O); k.f = take(packed_floats(w.O, w.I)); k.b = take(packed_floats(w.I, w.O)); return k; };
    auto blk = [&](const Blk& b, int split) {
      BlkPk k; k.a = wn(b.a); k.h = wn(b.h); k.s = b.same ? k.h : wn(b.s); k.bsum = take(b.Cout);
      k.a_fi = k.a_fn = k.a_bi = k.s_fi = k.s_fn = k.s_bi = 0;
      if (split) {   // concat input [image part (split columns) | noise part]  (a concat block always has its projected skip: cdim + nd != Cout)
        const int nn = b.a.I - split;
        k.a_fi = take(packed_floats(b.a.O, split)); k.a_fn = take(packed_floats(b.a.O, nn)); k.a_bi = take(packed_floats(split, b.a.O));
        k.s_fi = take(packed_floats(b.s.O, split)); k.s_fn = take(packed_floats(b.s.O, nn)); k.s_bi = take(packed_floats(split, b.s.O));
      }
      return k;
    };
    auto lin = [&](const Lin& l, int split) {
      LinPk k; k.f = take(packed_floats(l.out, l.in)); k.b = take(packed_floats(l.in, l.out)); k.fi = k.fn = k.bi = k.bn = 0;
      if (split) {
        k.fi = take(packed_floats(l.out, split)); k.fn = take(packed_floats(l.out, l.in - split));
        k.bi = take(packed_floats(split, l.out)); k.bn = take(packed_floats(l.in - split, l.out));
      }
      return k;
    };
    for (auto& b : P.trunk) trunk.push_back(blk(b, 0));
    if (P.kind == 5) {
      for (auto& o : P.head) {
        HeadPk hp{};
        if (o.res) hp.k = blk(o.b, o.concat ? P.cdim : 0); else hp.lk = lin(o.l, o.concat ? P.cdim : 0);
        head.push_back(hp);
      }
    } else { mu0 = lin(P.mu0, 0); lv0 = lin(P.lv0, 0); efc = lin(P.efc, P.cdim); mu = lin(P.mu, 0); lv = lin(P.lv, 0); }
    for (auto& b : P.dec) dec.push_back(blk(b, 0));
    total = off;
  }
};

struct Bump {
  float* base; size_t cap; size_t off = 0; bool ok = true;
  Bump(float* b, size_t c) : base(b), cap(c) {}
  float* take(size_t n) {
    size_t o = off; off += al64(n);
    if (off > cap) { ok = false; return base; }
    return base ? base + o : nullptr;
  }
};

int res_desc_ok(const ardae_model_desc& d) {
  ARDAE_CHECK_ARG(d.input_dim == 784 && d.noise_dim >= 1 && d.z_dim >= 1 && d.h_dim >= 1, "model: the residual-conv models are hard-wired to 28x28x1 inputs");
  ARDAE_CHECK_ARG(d.act == ACT_ELU, "model: the residual-conv models use ELU (--model-nonlin elu; models/ivae/auxresconv.py:69 asserts it)");
  ARDAE_CHECK_ARG(d.kind != 5 || (d.n_layers >= 1 && d.n_layers <= 4), "model: ResConvIPVAE takes --model-n-layers 1 .. 4");
  ARDAE_CHECK_ARG(d.kind != 5 || ((d.flags >> 1) & 7) <= HEAD_RES_MLP_LIN, "model: unknown sampler head %d (desc.flags bits 1-3)", (d.flags >> 1) & 7);
  ARDAE_CHECK_ARG(d.kind == 5 || ((d.flags >> 1) & 7) == 0, "model: the sampler-head bits of desc.flags belong to kind 5");
  ARDAE_CHECK_ARG(d.kind == 6 || !(d.flags & ARDAE_MODEL_CLIPPED), "model: ARDAE_MODEL_CLIPPED belongs to kind 6");
  if (d.kind == 5) {   // a concat block needs its projected skip (same_dim would make the skip the 612-wide concat itself)
    const int ht = (d.flags >> 1) & 7;
    ARDAE_CHECK_ARG(ht == HEAD_MLP || 512 + d.noise_dim != d.h_dim, "model: ResConvIPVAE with c_dim + noise_dim == h_dim (identity skip over the concat input) is not built");
  }
  return 0;
}

// one linear launch, one or two sources
int lin1(int epi, int act, int M, int Nout, const float* x, int ldx, int K, const float* wp, LinArgs a, hipStream_t st) {
  a.M = M; a.Nout = Nout; a.nsrc = 1; a.act = act;
  a.src[0].x = x; a.src[0].ld = ldx; a.src[0].K = K; a.src[0].wp = wp;
  return launch_linear(a, epi, st);
}
int lin2(int epi, int act, int M, int Nout, const float* x0, int ld0, int K0, const float* wp0, const float* x1, int ld1, int K1, const float* wp1,
         LinArgs a, hipStream_t st) {
  a.M = M; a.Nout = Nout; a.nsrc = 2; a.act = act;
  a.src[0].x = x0; a.src[0].ld = ld0; a.src[0].K = K0; a.src[0].wp = wp0;
  a.src[1].x = x1; a.src[1].ld = ld1; a.src[1].K = K1; a.src[1].wp = wp1;
  return launch_linear(a, epi, st);
}

// saved activations of one block
struct BlkBuf { float *colsx, *hmid, *colsh, *out; };
size_t blk_rows(const Blk& b, int images) { return (size_t)images * b.Hout * b.Hout; }
void blk_carve(const Blk& b, int images, Bump& ws, BlkBuf& u, bool keep_cols) {
  const size_t R = blk_rows(b, images);
  u.colsx = b.conv ? ws.take(R * b.a.I) : nullptr;
  u.hmid = ws.take(R * b.Cout);
  u.colsh = b.conv ? ws.take(R * b.h.I) : nullptr;
  u.out = ws.take(R * b.Cout);
  (void)keep_cols;
}

// forward of a residual block on `images` images (conv) / rows (linear); x: [images*Hin*Hin, Cin]; act_out: ELU after the block or none
int blk_fwd(const Blk& b, const BlkPk& k, const float* params, const float* packed, const float* x, int images, int act_out, BlkBuf& u, hipStream_t st) {
  const int R = (int)blk_rows(b, images);
  const float* cx = x;
  if (b.conv) {
    RES_LAUNCH(im2col3_kernel, (int64_t)R * b.a.I, x, b.Hin, b.Hin, b.Cin, b.Hout, b.Hout, b.stride, u.colsx, (int64_t)R * b.a.I);
    cx = u.colsx;
  }
  {
    LinArgs A{}; A.bias = params + b.a.bias; A.Y = u.hmid; A.ldY = b.Cout;
    // conv blocks: ELU as the epilogue activation (their Cout <= 32 columns run on the generic kernel either way)
    ARDAE_TRY(lin1(EPI_ACT, b.conv ? ACT_ELU : ACT_RELU, R, b.Cout, cx, b.a.I, b.a.I, packed + k.a.f, A, st));
  }
  const float* ch = u.hmid;
  if (b.conv) {
    RES_LAUNCH(im2col3_kernel, (int64_t)R * b.h.I, u.hmid, b.Hout, b.Hout, b.Cout, b.Hout, b.Hout, 1, u.colsh, (int64_t)R * b.h.I);
    ch = u.colsh;
  }
  LinArgs A{}; A.bias = packed + k.bsum; A.Y = u.out; A.ldY = b.Cout;
  // the activation behind a conv block rides in the epilogue too; linear blocks keep it apart (their h-wide N-row products stay
  // eligible for the software-pipelined kernel, which is not instantiated for ELU)
  const int act_epi = b.conv ? act_out : ACT_NONE;
  ARDAE_TRY(lin2(EPI_ACT, act_epi, R, b.Cout, ch, b.h.I, b.h.I, packed + k.h.f, cx, b.a.I, b.s.I, packed + k.s.f, A, st));
  if (act_out != act_epi) RES_LAUNCH(act_inplace_kernel, (int64_t)R * b.Cout, u.out, act_out, (int64_t)R * b.Cout);
  return 0;
}

// scratch of one block's backward (reused from block to block) and the gradient destinations of its three operators
struct BwdScratch { float *g, *dh, *dcols; };
struct WnGrad { float* dW; float* db; float beta; };   // beta: accumulate into the destination (plain operators write the gradient buffer itself)
struct BlkGrad { WnGrad a, h, s; };

void push_wgrad(std::vector<WgradProblem>& probs, std::vector<std::pair<size_t, size_t>>& need, int M, int O, int I, const float* G, const float* X, int ldX,
                float* out, int ldout, float* out_bias) {
  WgradProblem p;
  memset(&p, 0, sizeof(p));
  p.M = M; p.O = O; p.I = I; p.npairs = 1;
  p.G[0] = G; p.ldG[0] = O; p.X[0] = X; p.ldX[0] = ldX;
  p.bias_pair = out_bias ? 0 : -1;
  p.out = out; p.ldout = ldout; p.out_bias = out_bias; p.beta = 0.f;
  probs.push_back(p);
  (void)need;
}
// assign splits + scratch and launch (at most ARDAE_WGRAD_MAX_PROBLEMS per batch)
int flush_wgrad(std::vector<WgradProblem>& probs, float* scratch, size_t scratch_floats, hipStream_t st) {
  size_t i0 = 0;
  while (i0 < probs.size()) {
    const size_t n = std::min<size_t>(probs.size() - i0, ARDAE_WGRAD_MAX_PROBLEMS);
    Bump ws(scratch, scratch_floats);
    for (size_t i = i0; i < i0 + n; ++i) {
      WgradProblem& p = probs[i];
      p.splits = wgrad_splits(p.M, p.O, p.I, (int)n);
      p.partial = ws.take((size_t)p.splits * p.O * p.I);
      p.partial_vec = ws.take((size_t)p.splits * 2 * p.O);
    }
    ARDAE_CHECK_ARG(ws.ok, "res model: weight-gradient scratch too small");
    ARDAE_TRY(launch_wgrad_batch(probs.data() + i0, (int)n, st));
    i0 += n;
  }
  probs.clear();
  return 0;
}
size_t wgrad_scratch_floats(int M, int O, int I) {
  const int s = wgrad_splits(M, O, I, 3);
  const int s1 = wgrad_splits(M, O, I, 1), s20 = wgrad_splits(M, O, I, ARDAE_WGRAD_MAX_PROBLEMS);
  const int sm = std::max(s, std::max(s1, s20));
  return al64((size_t)sm * O * I) + al64((size_t)sm * 2 * O);
}

// backward of a block: d_out [R, Cout] (gradient w.r.t. the block's output AFTER act_out) -> d_x [rows_in, Cin] (may be null)
int blk_bwd(const Blk& b, const BlkPk& k, const float* packed, const float* x, int images, int act_out, const BlkBuf& u, const float* d_out, float* d_x,
            const BwdScratch& sc, const BlkGrad& gr, float* wscratch, size_t wscratch_floats, hipStream_t st) {
  const int R = (int)blk_rows(b, images);
  const int64_t nout = (int64_t)R * b.Cout;
  const float* g = d_out;
  if (act_out != ACT_NONE) {
    RES_LAUNCH(mul_dact_kernel, nout, d_out, u.out, act_out, sc.g, nout);
    g = sc.g;
  }
  const float* cx = b.conv ? u.colsx : x;
  const float* ch = b.conv ? u.colsh : u.hmid;
  // d hmid
  if (b.conv) {
    LinArgs A{}; A.Y = sc.dcols; A.ldY = b.h.I;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, b.h.I, g, b.Cout, b.Cout, packed + k.h.b, A, st));
    RES_LAUNCH(col2im3_kernel, nout, sc.dcols, b.Hout, b.Hout, b.Cout, b.Hout, b.Hout, 1, sc.dh, nout, (const float*)u.hmid, (int)ACT_ELU);
  } else {
    LinArgs A{}; A.S = u.hmid; A.ldS = b.Cout; A.Y = sc.dh; A.ldY = b.Cout;
    ARDAE_TRY(lin1(EPI_DACT, ACT_RELU, R, b.Cout, g, b.Cout, b.Cout, packed + k.h.b, A, st));
  }
  // weight gradients of the three operators (their bias gradients: column sums of d hmid / g; b_h1 and b_01 share g's)
  std::vector<WgradProblem> probs;
  std::vector<std::pair<size_t, size_t>> dummy;
  push_wgrad(probs, dummy, R, b.Cout, b.h.I, g, ch, b.h.I, gr.h.dW, b.h.I, gr.h.db);
  push_wgrad(probs, dummy, R, b.Cout, b.s.I, g, cx, b.s.I, gr.s.dW, b.s.I, gr.s.db);
  push_wgrad(probs, dummy, R, b.Cout, b.a.I, sc.dh, cx, b.a.I, gr.a.dW, b.a.I, gr.a.db);
  ARDAE_TRY(flush_wgrad(probs, wscratch, wscratch_floats, st));
  if (!d_x) return 0;
  // d x = W_01^T g + W_0h^T d hmid
  if (b.conv) {
    LinArgs A{}; A.Y = sc.dcols; A.ldY = b.a.I;
    ARDAE_TRY(lin2(EPI_ACT, ACT_NONE, R, b.a.I, g, b.Cout, b.Cout, packed + k.s.b, sc.dh, b.Cout, b.Cout, packed + k.a.b, A, st));
    const int64_t nin = (int64_t)images * b.Hin * b.Hin * b.Cin;
    RES_LAUNCH(col2im3_kernel, nin, sc.dcols, b.Hin, b.Hin, b.Cin, b.Hout, b.Hout, b.stride, d_x, nin, (const float*)nullptr, 0);
  } else {
    LinArgs A{}; A.Y = d_x; A.ldY = b.a.I;
    ARDAE_TRY(lin2(EPI_ACT, ACT_NONE, R, b.a.I, g, b.Cout, b.Cout, packed + k.s.b, sc.dh, b.Cout, b.Cout, packed + k.a.b, A, st));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ workspace carving
struct ResWs {
  // trunk (B images)
  float* x2; std::vector<BlkBuf> tb; float *flat, *inp;             // 2x-1, block buffers, NCHW-flattened conv output [B,512], trunk output [B,cdim]
  // sampler (R = B*nz rows)
  float *zero;                                                      // zero noise for encode(std=0)
  struct HeadBuf { float *rba, *rbs, *hmid, *out; };                // kind 5, per head operator: per-image row biases [B,out] (concat), ResLinear hidden [R,out], output [R,out]
  std::vector<HeadBuf> hb; float* z;                                // z [R,zd]
  float *mu0, *lv0r, *lv0, *z0, *rbh, *hh, *mu, *lvr, *lv;          // kind 6
  // decoder (R images)
  std::vector<BlkBuf> db; float *d4, *u8, *c7, *u14, *u28;          // block buffers; NHWC 4x4x32, upsampled / cropped maps
  float *rec_row, *pri_row, *dlogit, *dzq;
  // backward scratch
  BwdScratch sc; float *da, *dbuf, *dflat, *dinp, *dR0, *dR1, *dB0, *dB1, *dB2;
  float* dweff; float* dbias; float* wscratch; size_t wscratch_floats;
};

size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

// mode 0: encode (trunk + sampler forward), 1: vae forward + backward, 2: decode only (B = rows, nz = 1)
void res_carve(const ResLayout& P, Bump& ws, int B, int nz, int mode, ResWs& W) {
  const size_t R = (size_t)B * nz;
  const bool enc = mode != 2, dec = mode != 0, bwd = mode == 1;
  if (enc) {
    W.x2 = ws.take((size_t)B * 784);
    W.tb.resize(P.trunk.size());
    for (size_t i = 0; i < P.trunk.size(); ++i) blk_carve(P.trunk[i], B, ws, W.tb[i], bwd);
    W.flat = ws.take((size_t)B * 512);
    W.inp = W.tb.back().out;
    W.zero = ws.take(R * (P.nd + P.zd));
    if (P.kind == 5) {
      W.hb.resize(P.head.size());
      for (size_t i = 0; i < P.head.size(); ++i) {
        const HeadOp& o = P.head[i];
        W.hb[i].rba = o.concat ? ws.take((size_t)B * o.out) : nullptr;
        W.hb[i].rbs = (o.concat && o.res) ? ws.take((size_t)B * o.out) : nullptr;
        W.hb[i].hmid = o.res ? ws.take(R * o.out) : nullptr;
        W.hb[i].out = ws.take(R * o.out);
      }
      W.z = ws.take(R * P.zd);
    } else {
      W.mu0 = ws.take((size_t)B * P.nd); W.lv0r = ws.take((size_t)B * P.nd); W.lv0 = ws.take((size_t)B * P.nd); W.z0 = ws.take(R * P.nd);
      W.rbh = ws.take((size_t)B * P.cdim); W.hh = ws.take(R * P.cdim); W.mu = ws.take(R * P.zd); W.lvr = ws.take(R * P.zd); W.lv = ws.take(R * P.zd);
      W.z = ws.take(R * P.zd);
    }
  }
  if (dec) {
    W.db.resize(P.dec.size());
    for (size_t i = 0; i < P.dec.size(); ++i) blk_carve(P.dec[i], (int)R, ws, W.db[i], bwd);
    W.d4 = ws.take(R * 512); W.u8 = ws.take(R * 64 * 32); W.c7 = ws.take(R * 49 * 32); W.u14 = ws.take(R * 196 * 32); W.u28 = ws.take(R * 784 * 16);
  }
  if (mode == 1) { W.rec_row = ws.take(R); W.pri_row = ws.take(R); W.dlogit = ws.take(R * 784); W.dzq = ws.take(R * P.zd); }
  if (bwd) {
    // the largest per-block scratch: decoder convs at 28 x 28 (R images) and the trunk's first convs (B images)
    size_t mg = 0, mc = 0;
    auto upd = [&](const Blk& b, size_t images) {
      const size_t rows = images * b.Hout * b.Hout;
      mg = max_sz(mg, rows * b.Cout);
      mc = max_sz(mc, rows * max_sz(b.h.I, b.a.I));
    };
    for (auto& b : P.trunk) upd(b, B);
    for (auto& b : P.dec) upd(b, R);
    W.sc.g = ws.take(mg); W.sc.dh = ws.take(mg); W.sc.dcols = ws.take(mc);
    const size_t act_max = max_sz(R * 784 * 16, (size_t)B * 196 * 16);     // largest activation map (decoder 28 x 28 x 16)
    W.da = ws.take(act_max); W.dbuf = ws.take(act_max);
    W.dflat = ws.take((size_t)B * 512); W.dinp = ws.take((size_t)B * P.cdim);
    W.dR0 = ws.take(R * max_sz(P.hdim, P.cdim)); W.dR1 = ws.take(R * max_sz(P.hdim, P.cdim));
    W.dB0 = ws.take((size_t)B * max_sz(P.hdim, P.cdim)); W.dB1 = ws.take((size_t)B * max_sz(P.hdim, P.cdim)); W.dB2 = ws.take((size_t)B * max_sz(P.hdim, P.cdim));
    W.dweff = ws.take(P.total); W.dbias = ws.take(P.total / 8 + 4096);
    size_t wsf = 0;
    auto wupd = [&](const Blk& b, size_t images) {
      const int rows = (int)(images * b.Hout * b.Hout);
      wsf = max_sz(wsf, wgrad_scratch_floats(rows, b.Cout, b.h.I) + wgrad_scratch_floats(rows, b.Cout, b.s.I) + wgrad_scratch_floats(rows, b.Cout, b.a.I));
    };
    for (auto& b : P.trunk) wupd(b, B);
    for (auto& b : P.dec) wupd(b, R);
    // sampler tail: all problems of a head operator in one batch
    const int Ri = (int)R;
    for (auto& o : P.head) {
      const int cin = o.concat ? P.nd : o.in;
      size_t need = 3 * wgrad_scratch_floats(Ri, o.out, std::max(cin, o.out));
      if (o.concat) need += 2 * wgrad_scratch_floats(B, o.out, P.cdim);
      wsf = max_sz(wsf, need);
    }
    wsf = max_sz(wsf, 2 * (wgrad_scratch_floats(Ri, P.hdim, P.hdim) + wgrad_scratch_floats(Ri, P.hdim, P.cdim + P.nd) + wgrad_scratch_floats(Ri, P.zd, P.hdim)) +
                          3 * wgrad_scratch_floats(B, P.nd, P.cdim) + 2 * wgrad_scratch_floats(B, P.hdim, P.cdim));
    W.wscratch_floats = wsf + 1024;
    W.wscratch = ws.take(W.wscratch_floats);
  }
}

size_t res_workspace(const ResLayout& P, int B, int nz, int mode) {
  Bump ws(nullptr, ~size_t(0) >> 2);
  ResWs W;
  res_carve(P, ws, B, nz, mode, W);
  return ws.off;
}

// ------------------------------------------------------------------------------------------------ forward pieces
int trunk_fwd(const ResLayout& P, const ResPacked& K, const float* params, const float* packed, const float* x, int B, ResWs& W, hipStream_t st) {
  if (P.center) ARDAE_TRY(launch_affine(x, (int64_t)B * 784, 2.f, -1.f, W.x2, st));       // do_center (ivae/resconv.py:131-132)
  else ARDAE_TRY(launch_affine(x, (int64_t)B * 784, 1.f, 0.f, W.x2, st));
  const float* cur = W.x2;
  for (size_t i = 0; i < 5; ++i) {
    ARDAE_TRY(blk_fwd(P.trunk[i], K.trunk[i], params, packed, cur, B, ACT_ELU, W.tb[i], st));
    cur = W.tb[i].out;
  }
  RES_LAUNCH(nhwc_nchw_kernel, (int64_t)B * 512, cur, 16, 32, W.flat, (int64_t)B * 512, 0);
  return blk_fwd(P.trunk[5], K.trunk[5], params, packed, W.flat, B, ACT_ELU, W.tb[5], st);
}

// sampler tail on R = B*nz rows; noise [R, nd] (kind 5) or [R, nd + zd] rows [eps0 | eps] (kind 6); z -> z_out
// raw0 (clipped class, std = 0 pass only - `noise` is then the zero block): the unscaled eps0 [R, nd] z0 keeps, or NULL (zeros)
int sampler_fwd(const ResLayout& P, const ResPacked& K, const float* params, const float* packed, const float* noise, int B, int nz, ResWs& W, float* z_out,
                float* hidden_out, hipStream_t st, const float* raw0 = nullptr, bool std0 = false) {
  const int R = B * nz;
  if (P.kind == 5) {
    // encode.fc operator by operator; the per-image half of a concat input enters as a row bias (computed once per image)
    const float* x = nullptr;
    for (size_t i = 0; i < P.head.size(); ++i) {
      const HeadOp& o = P.head[i];
      const ResPacked::HeadPk& hk = K.head[i];
      ResWs::HeadBuf& u = W.hb[i];
      float* out = i + 1 == P.head.size() ? z_out : u.out;
      if (o.res) {
        const Blk& b = o.b; const BlkPk& k = hk.k;
        if (o.concat) {
          // rba = W0h[:, :c] inp + b0h, rbs = W01[:, :c] inp + (bh1 + b01)
          { LinArgs A{}; A.bias = params + b.a.bias; A.Y = u.rba; A.ldY = o.out; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, o.out, W.inp, P.cdim, P.cdim, packed + k.a_fi, A, st)); }
          { LinArgs A{}; A.bias = packed + k.bsum; A.Y = u.rbs; A.ldY = o.out; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, o.out, W.inp, P.cdim, P.cdim, packed + k.s_fi, A, st)); }
          { LinArgs A{}; A.rowbias = u.rba; A.rowbias_ld = o.out; A.rows_per_group = nz; A.Y = u.hmid; A.ldY = o.out;
            ARDAE_TRY(lin1(EPI_ACT, ACT_RELU, R, o.out, noise, P.nd, P.nd, packed + k.a_fn, A, st)); }
          { LinArgs A{}; A.rowbias = u.rbs; A.rowbias_ld = o.out; A.rows_per_group = nz; A.Y = out; A.ldY = o.out;
            ARDAE_TRY(lin2(EPI_ACT, ACT_NONE, R, o.out, u.hmid, o.out, o.out, packed + k.h.f, noise, P.nd, P.nd, packed + k.s_fn, A, st)); }
        } else {
          { LinArgs A{}; A.bias = params + b.a.bias; A.Y = u.hmid; A.ldY = o.out; ARDAE_TRY(lin1(EPI_ACT, ACT_RELU, R, o.out, x, o.in, o.in, packed + k.a.f, A, st)); }
          if (!b.same) {
            LinArgs A{}; A.bias = packed + k.bsum; A.Y = out; A.ldY = o.out;
            ARDAE_TRY(lin2(EPI_ACT, ACT_NONE, R, o.out, u.hmid, o.out, o.out, packed + k.h.f, x, o.in, o.in, packed + k.s.f, A, st));
          } else {   // identity skip (ResLinear(same_dim=True), models/layers.py:82-84)
            LinArgs A{}; A.bias = params + b.h.bias; A.Y = out; A.ldY = o.out;
            ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, o.out, u.hmid, o.out, o.out, packed + k.h.f, A, st));
            ARDAE_TRY(launch_axpy(x, (int64_t)R * o.out, 1.f, out, st));
          }
        }
      } else {
        const Lin& l = o.l; const LinPk& k = hk.lk;
        if (o.concat) {
          { LinArgs A{}; A.bias = params + l.b; A.Y = u.rba; A.ldY = o.out; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, o.out, W.inp, P.cdim, P.cdim, packed + k.fi, A, st)); }
          { LinArgs A{}; A.rowbias = u.rba; A.rowbias_ld = o.out; A.rows_per_group = nz; A.Y = out; A.ldY = o.out;
            ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, o.out, noise, P.nd, P.nd, packed + k.fn, A, st)); }
        } else {
          LinArgs A{}; A.bias = params + l.b; A.Y = out; A.ldY = o.out;
          ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, o.out, x, o.in, o.in, packed + k.f, A, st));
        }
      }
      if (o.act != ACT_NONE) RES_LAUNCH(act_inplace_kernel, (int64_t)R * o.out, out, o.act, (int64_t)R * o.out);
      x = out;
    }
    return 0;
  }
  const int ldn = P.nd + P.zd;
  { LinArgs A{}; A.bias = params + P.mu0.b; A.Y = W.mu0; A.ldY = P.nd; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, P.nd, W.inp, P.cdim, P.cdim, packed + K.mu0.f, A, st)); }
  { LinArgs A{}; A.bias = params + P.lv0.b; A.Y = W.lv0r; A.ldY = P.nd; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, P.nd, W.inp, P.cdim, P.cdim, packed + K.lv0.f, A, st)); }
  RES_LAUNCH(spm4_kernel, (int64_t)B * P.nd, W.lv0r, (const float*)nullptr, W.lv0, (int64_t)B * P.nd, (int)P.clipped);
  // clipped: z0 = mu0 + (std exp(lv0 / 2) + 1) eps0; the std = 0 pass (noise = zeros) takes its unscaled draw from raw0 (none: z0 = mu0)
  if (P.clipped && !(std0 && !raw0)) ARDAE_TRY(launch_reparam_fwd(W.mu0, W.lv0, noise, ldn, R, P.nd, nz, W.z0, st, 1.f, std0 ? raw0 : nullptr, P.nd));
  else ARDAE_TRY(launch_reparam_fwd(W.mu0, W.lv0, noise, ldn, R, P.nd, nz, W.z0, st));
  { LinArgs A{}; A.bias = params + P.efc.b; A.Y = W.rbh; A.ldY = P.cdim; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, P.cdim, W.inp, P.cdim, P.cdim, packed + K.efc.fi, A, st)); }
  { LinArgs A{}; A.rowbias = W.rbh; A.rowbias_ld = P.cdim; A.rows_per_group = nz; A.Y = W.hh; A.ldY = P.cdim;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.cdim, W.z0, P.nd, P.nd, packed + K.efc.fn, A, st)); }
  RES_LAUNCH(act_inplace_kernel, (int64_t)R * P.cdim, W.hh, (int)ACT_ELU, (int64_t)R * P.cdim);
  if (hidden_out) ARDAE_TRY(launch_copy(W.hh, (size_t)R * P.cdim, hidden_out, st));
  { LinArgs A{}; A.bias = params + P.mu.b; A.Y = W.mu; A.ldY = P.zd; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.zd, W.hh, P.cdim, P.cdim, packed + K.mu.f, A, st)); }
  { LinArgs A{}; A.bias = params + P.lv.b; A.Y = W.lvr; A.ldY = P.zd; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.zd, W.hh, P.cdim, P.cdim, packed + K.lv.f, A, st)); }
  RES_LAUNCH(spm4_kernel, (int64_t)R * P.zd, W.lvr, (const float*)nullptr, W.lv, (int64_t)R * P.zd, (int)P.clipped);
  return launch_reparam_fwd(W.mu, W.lv, noise + P.nd, ldn, R, P.zd, 1, z_out, st);
}

// decoder on R rows: z [R, zd] -> logits [R, 784]
int decoder_fwd(const ResLayout& P, const ResPacked& K, const float* params, const float* packed, const float* z, int R, ResWs& W, float* logits, hipStream_t st) {
  ARDAE_TRY(blk_fwd(P.dec[0], K.dec[0], params, packed, z, R, ACT_ELU, W.db[0], st));
  ARDAE_TRY(blk_fwd(P.dec[1], K.dec[1], params, packed, W.db[0].out, R, ACT_ELU, W.db[1], st));
  RES_LAUNCH(nhwc_nchw_kernel, (int64_t)R * 512, W.db[1].out, 16, 32, W.d4, (int64_t)R * 512, 1);          // view(-1, 32, 4, 4) -> NHWC
  RES_LAUNCH(upsample2_fwd_kernel, (int64_t)R * 64 * 32, W.d4, 4, 32, W.u8, (int64_t)R * 64 * 32);
  ARDAE_TRY(blk_fwd(P.dec[2], K.dec[2], params, packed, W.u8, R, ACT_ELU, W.db[2], st));
  ARDAE_TRY(blk_fwd(P.dec[3], K.dec[3], params, packed, W.db[2].out, R, ACT_ELU, W.db[3], st));
  RES_LAUNCH(crop_pad_kernel, (int64_t)R * 49 * 32, W.db[3].out, 8, 7, 32, W.c7, (int64_t)R * 49 * 32);    // slicer[:, :, :-1, :-1]
  RES_LAUNCH(upsample2_fwd_kernel, (int64_t)R * 196 * 32, W.c7, 7, 32, W.u14, (int64_t)R * 196 * 32);
  ARDAE_TRY(blk_fwd(P.dec[4], K.dec[4], params, packed, W.u14, R, ACT_ELU, W.db[4], st));
  ARDAE_TRY(blk_fwd(P.dec[5], K.dec[5], params, packed, W.db[4].out, R, ACT_ELU, W.db[5], st));
  RES_LAUNCH(upsample2_fwd_kernel, (int64_t)R * 784 * 16, W.db[5].out, 14, 16, W.u28, (int64_t)R * 784 * 16);
  ARDAE_TRY(blk_fwd(P.dec[6], K.dec[6], params, packed, W.u28, R, ACT_NONE, W.db[6], st));
  if (logits && logits != W.db[6].out) ARDAE_TRY(launch_copy(W.db[6].out, (size_t)R * 784, logits, st));
  return 0;
}

const float* noise_or_zero(const ResLayout& P, const float* noise, int R, ResWs& W, hipStream_t st, int& rc) {
  rc = 0;
  if (noise) return noise;
  if (rc == 0) rc = launch_fill(W.zero, (size_t)R * (P.nd + P.zd), 0.f, st);
  return W.zero;
}

// gradient destinations of a WN operator inside the dweff / dbias scratch
struct GradMap {
  const ResLayout& P; float* dweff; float* dbias; size_t boff = 0;
  std::vector<WnBwdItem> items;
  const float* params; const float* packed; float* grads;
  float grads_beta = 0.f;
  WnGrad wn(const WN& w, const WNPk& k) {
    if (w.plain) return WnGrad{grads + w.dir, grads + w.bias, grads_beta};     // nn.Linear operator: the weight gradient IS the parameter gradient
    WnGrad g; g.dW = dweff + w.dir; g.db = dbias + boff; g.beta = 0.f; boff += al64(w.O);
    WnBwdItem it; it.dW = g.dW; it.dir = params + w.dir; it.scale = params + w.scale; it.inv = packed + k.inv; it.db = g.db;
    it.gdir = grads + w.dir; it.gscale = grads + w.scale; it.gbias = grads + w.bias; it.O = w.O; it.I = w.I; it.norm = w.norm ? 1 : 0;
    items.push_back(it);
    return g;
  }
  BlkGrad blk(const Blk& b, const BlkPk& k) { BlkGrad g; g.a = wn(b.a, k.a); g.h = wn(b.h, k.h); g.s = b.same ? g.h : wn(b.s, k.s); return g; }
};

}  // namespace

// ================================================================================================ entry points
size_t res_model_param_floats(const ardae_model_desc& d) { return res_desc_ok(d) ? 0 : ResLayout(d).total; }
size_t res_model_packed_floats(const ardae_model_desc& d) { return res_desc_ok(d) ? 0 : ResPacked(ResLayout(d)).total; }
size_t res_model_workspace_floats(const ardae_model_desc& d, int B, int nz, int mode) {
  if (res_desc_ok(d)) return 0;
  return res_workspace(ResLayout(d), B, nz, mode == 3 ? 0 : mode);
}

int res_model_pack(const ardae_model_desc& d, const float* params, float* packed, hipStream_t st) {
  ARDAE_TRY(res_desc_ok(d));
  const ResLayout P(d);
  const ResPacked K(P);
  std::vector<PackItem> items;
  std::vector<WnComposeItem> comp;
  // effective weight of an operator: composed into the packed buffer (weight norm), or the parameter itself (plain nn.Linear operator)
  auto weff = [&](const WN& w, const WNPk& k) -> const float* { return w.plain ? params + w.dir : packed + k.weff; };
  auto wn = [&](const WN& w, const WNPk& k) -> int {
    if (!w.plain) comp.push_back(WnComposeItem{params + w.dir, params + w.scale, packed + k.weff, packed + k.inv, w.O, w.I, w.norm ? 1 : 0, nullptr, nullptr});
    items.push_back(PackItem{weff(w, k), w.I, w.O, w.I, 0, packed + k.f});
    items.push_back(PackItem{weff(w, k), w.I, w.I, w.O, 1, packed + k.b});
    return 0;
  };
  auto blk = [&](const Blk& b, const BlkPk& k, int split) -> int {
    ARDAE_TRY(wn(b.a, k.a)); ARDAE_TRY(wn(b.h, k.h));
    if (b.same) return 0;          // identity skip: no third operator, the block's bias is b_h1
    ARDAE_TRY(wn(b.s, k.s));
    // bsum = b_h + b_s rides in the same launch
    comp.push_back(WnComposeItem{nullptr, nullptr, packed + k.bsum, nullptr, b.Cout, 0, 0, params + b.h.bias, params + b.s.bias});
    if (split) {
      const int nn = b.a.I - split;
      items.push_back(PackItem{weff(b.a, k.a), b.a.I, b.a.O, split, 0, packed + k.a_fi});
      items.push_back(PackItem{weff(b.a, k.a) + split, b.a.I, b.a.O, nn, 0, packed + k.a_fn});
      items.push_back(PackItem{weff(b.a, k.a), b.a.I, split, b.a.O, 1, packed + k.a_bi});
      items.push_back(PackItem{weff(b.s, k.s), b.s.I, b.s.O, split, 0, packed + k.s_fi});
      items.push_back(PackItem{weff(b.s, k.s) + split, b.s.I, b.s.O, nn, 0, packed + k.s_fn});
      items.push_back(PackItem{weff(b.s, k.s), b.s.I, split, b.s.O, 1, packed + k.s_bi});
    }
    return 0;
  };
  auto lin = [&](const Lin& l, const LinPk& k, int split) {
    items.push_back(PackItem{params + l.w, l.in, l.out, l.in, 0, packed + k.f});
    items.push_back(PackItem{params + l.w, l.in, l.in, l.out, 1, packed + k.b});
    if (split) {
      items.push_back(PackItem{params + l.w, l.in, l.out, split, 0, packed + k.fi});
      items.push_back(PackItem{params + l.w + split, l.in, l.out, l.in - split, 0, packed + k.fn});
      items.push_back(PackItem{params + l.w, l.in, split, l.out, 1, packed + k.bi});
      items.push_back(PackItem{params + l.w + split, l.in, l.in - split, l.out, 1, packed + k.bn});
    }
  };
  for (size_t i = 0; i < P.trunk.size(); ++i) ARDAE_TRY(blk(P.trunk[i], K.trunk[i], 0));
  if (P.kind == 5) {
    for (size_t i = 0; i < P.head.size(); ++i) {
      const HeadOp& o = P.head[i];
      if (o.res) ARDAE_TRY(blk(o.b, K.head[i].k, o.concat ? P.cdim : 0)); else lin(o.l, K.head[i].lk, o.concat ? P.cdim : 0);
    }
  } else { lin(P.mu0, K.mu0, 0); lin(P.lv0, K.lv0, 0); lin(P.efc, K.efc, P.cdim); lin(P.mu, K.mu, 0); lin(P.lv, K.lv, 0); }
  for (size_t i = 0; i < P.dec.size(); ++i) ARDAE_TRY(blk(P.dec[i], K.dec[i], 0));
  for (size_t i0 = 0; i0 < comp.size(); i0 += WNC_MAX) {
    WnComposeBatch cb;
    cb.n = (int)std::min<size_t>(WNC_MAX, comp.size() - i0);
    int maxo = 1;
    for (int i = 0; i < cb.n; ++i) {
      cb.it[i] = comp[i0 + i];
      maxo = std::max(maxo, cb.it[i].dir ? cb.it[i].O : ceil_div(cb.it[i].O, 256));
    }
    hipLaunchKernelGGL(wn_compose_batch_kernel, dim3(maxo, cb.n), dim3(256), 0, st, cb);
    ARDAE_LAUNCH_CHECK();
  }
  return launch_pack_batch(items.data(), (int)items.size(), st);
}

int res_model_encode(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                     float* workspace, size_t wsf, float* z_out, float* hidden_out, hipStream_t st, const float* raw0) {
  const ResLayout P(d);
  const ResPacked K(P);
  Bump ws(workspace, wsf);
  ResWs W;
  res_carve(P, ws, B, nz, 0, W);
  ARDAE_CHECK_ARG(ws.ok, "res model encode: workspace too small");
  ARDAE_CHECK_ARG(!hidden_out || (P.kind == 6 && nz == 1), "res model: the hidden1a context is the aux model's h at nz = 1");
  int rc;
  const float* nzp = noise_or_zero(P, noise, B * nz, W, st, rc);
  ARDAE_TRY(rc);
  ARDAE_TRY(trunk_fwd(P, K, params, packed, x, B, W, st));
  ARDAE_CHECK_ARG(!raw0 || (P.clipped && !noise), "res model: raw0 is the clipped class's unscaled eps0 of a std = 0 pass (noise NULL)");
  return sampler_fwd(P, K, params, packed, nzp, B, nz, W, z_out ? z_out : W.z, hidden_out, st, raw0, noise == nullptr);
}

int res_model_decode(const ardae_model_desc& d, const float* params, const float* packed, const float* z, int R, float* workspace, size_t wsf,
                     float* out0, hipStream_t st) {
  const ResLayout P(d);
  const ResPacked K(P);
  Bump ws(workspace, wsf);
  ResWs W;
  res_carve(P, ws, R, 1, 2, W);
  ARDAE_CHECK_ARG(ws.ok, "res model decode: workspace too small");
  return decoder_fwd(P, K, params, packed, z, R, W, out0, st);
}

int res_model_vae_forward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                          float beta, float* workspace, size_t wsf, float* z_out, float* losses, hipStream_t st) {
  const ResLayout P(d);
  const ResPacked K(P);
  Bump ws(workspace, wsf);
  ResWs W;
  res_carve(P, ws, B, nz, 1, W);
  ARDAE_CHECK_ARG(ws.ok, "res model vae_forward: workspace too small");
  const int R = B * nz;
  ARDAE_TRY(trunk_fwd(P, K, params, packed, x, B, W, st));
  ARDAE_TRY(sampler_fwd(P, K, params, packed, noise, B, nz, W, W.z, nullptr, st));
  ARDAE_TRY(launch_copy(W.z, (size_t)R * P.zd, z_out, st));
  ARDAE_TRY(decoder_fwd(P, K, params, packed, W.z, R, W, nullptr, st));
  ARDAE_TRY(launch_vae_loss(0, W.db[6].out, nullptr, x, W.z, R, nz, 784, P.zd, beta, 0, 0.f, nullptr, W.rec_row, W.pri_row, nullptr, nullptr, nullptr, st));
  return launch_vae_loss_finalize(W.rec_row, W.pri_row, R, beta, losses, st);
}

int res_model_vae_backward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                           float beta, float dloss, const float* dz_extra, float* workspace, size_t wsf, float* grads, float grads_beta, hipStream_t st) {
  const ResLayout P(d);
  const ResPacked K(P);
  Bump ws(workspace, wsf);
  ResWs W;
  res_carve(P, ws, B, nz, 1, W);
  ARDAE_CHECK_ARG(ws.ok, "res model vae_backward: workspace too small");
  const int R = B * nz;
  const float gscale = dloss / (float)R;
  GradMap gm{P, W.dweff, W.dbias, 0, {}, params, packed, grads};
  gm.grads_beta = grads_beta;
  // loss gradients: d logits, dz = gscale beta z + dz_extra
  ARDAE_TRY(launch_vae_loss(0, W.db[6].out, nullptr, x, W.z, R, nz, 784, P.zd, beta, 1, gscale, dz_extra, W.rec_row, W.pri_row, W.dlogit, nullptr, W.dzq, st));
  // ---- decoder backward (d_out ping-pongs between W.da and W.dbuf)
  std::vector<BlkGrad> gd(P.dec.size());
  for (size_t i = 0; i < P.dec.size(); ++i) gd[i] = gm.blk(P.dec[i], K.dec[i]);
  ARDAE_TRY(blk_bwd(P.dec[6], K.dec[6], packed, W.u28, R, ACT_NONE, W.db[6], W.dlogit, W.da, W.sc, gd[6], W.wscratch, W.wscratch_floats, st));      // -> d u28 [R,28,28,16]
  RES_LAUNCH(upsample2_bwd_kernel, (int64_t)R * 196 * 16, W.da, 14, 16, W.dbuf, (int64_t)R * 196 * 16);
  ARDAE_TRY(blk_bwd(P.dec[5], K.dec[5], packed, W.db[4].out, R, ACT_ELU, W.db[5], W.dbuf, W.da, W.sc, gd[5], W.wscratch, W.wscratch_floats, st));
  ARDAE_TRY(blk_bwd(P.dec[4], K.dec[4], packed, W.u14, R, ACT_ELU, W.db[4], W.da, W.dbuf, W.sc, gd[4], W.wscratch, W.wscratch_floats, st));          // -> d u14 [R,14,14,32]
  RES_LAUNCH(upsample2_bwd_kernel, (int64_t)R * 49 * 32, W.dbuf, 7, 32, W.da, (int64_t)R * 49 * 32);                                                 // -> d c7
  RES_LAUNCH(crop_pad_kernel, (int64_t)R * 64 * 32, W.da, 7, 8, 32, W.dbuf, (int64_t)R * 64 * 32);                                                    // -> d (block 3 out) [R,8,8,32]
  ARDAE_TRY(blk_bwd(P.dec[3], K.dec[3], packed, W.db[2].out, R, ACT_ELU, W.db[3], W.dbuf, W.da, W.sc, gd[3], W.wscratch, W.wscratch_floats, st));
  ARDAE_TRY(blk_bwd(P.dec[2], K.dec[2], packed, W.u8, R, ACT_ELU, W.db[2], W.da, W.dbuf, W.sc, gd[2], W.wscratch, W.wscratch_floats, st));            // -> d u8
  RES_LAUNCH(upsample2_bwd_kernel, (int64_t)R * 16 * 32, W.dbuf, 4, 32, W.da, (int64_t)R * 16 * 32);                                                 // -> d d4 (NHWC)
  RES_LAUNCH(nhwc_nchw_kernel, (int64_t)R * 512, W.da, 16, 32, W.dbuf, (int64_t)R * 512, 0);                                                          // -> NCHW-flat [R,512]
  ARDAE_TRY(blk_bwd(P.dec[1], K.dec[1], packed, W.db[0].out, R, ACT_ELU, W.db[1], W.dbuf, W.da, W.sc, gd[1], W.wscratch, W.wscratch_floats, st));
  ARDAE_TRY(blk_bwd(P.dec[0], K.dec[0], packed, W.z, R, ACT_ELU, W.db[0], W.da, W.dbuf, W.sc, gd[0], W.wscratch, W.wscratch_floats, st));             // -> dz (decoder part) [R,zd]
  ARDAE_TRY(launch_axpy(W.dbuf, (int64_t)R * P.zd, 1.f, W.dzq, st));                                                                                 // dz total in W.dzq
  // ---- sampler backward -> W.dinp [B, cdim]
  std::vector<WgradProblem> probs;
  std::vector<std::pair<size_t, size_t>> dummy;
  std::vector<WnBwdItem> plain;     // plain nn.Linear gradients are written in place by the wgrad kernel (no weight norm)
  if (P.kind == 5) {
    // encode.fc backwards, operator by operator: g = dL/d(output of the operator) [R, out]
    auto pw = [&](int M, int O, int I, const float* G, const float* X, int ldX, const WnGrad& g, int col0, int ldout, bool bias) {
      push_wgrad(probs, dummy, M, O, I, G, X, ldX, g.dW + col0, ldout, bias ? g.db : nullptr);
      probs.back().beta = g.beta;
    };
    float* g = W.dzq;
    float* pp[2] = {W.dR0, W.dR1};
    for (int i = (int)P.head.size() - 1; i >= 0; --i) {
      const HeadOp& o = P.head[i];
      const ResPacked::HeadPk& hk = K.head[i];
      const ResWs::HeadBuf& u = W.hb[i];
      const float* out = i + 1 == (int)P.head.size() ? W.z : u.out;
      const float* x = i == 0 ? nullptr : W.hb[i - 1].out;        // the operator's input (post-activation output of its predecessor)
      float* dx = pp[i & 1];
      if (o.act != ACT_NONE) RES_LAUNCH(mul_dact_kernel, (int64_t)R * o.out, g, out, o.act, g, (int64_t)R * o.out);
      if (o.res) {
        const Blk& b = o.b; const BlkPk& k = hk.k;
        const BlkGrad gr = gm.blk(b, k);
        float* dh = W.sc.dh;       // [R, out]: d hmid = (g W_h1) relu'(hmid)
        { LinArgs A{}; A.S = u.hmid; A.ldS = o.out; A.Y = dh; A.ldY = o.out; ARDAE_TRY(lin1(EPI_DACT, ACT_RELU, R, o.out, g, o.out, o.out, packed + k.h.b, A, st)); }
        pw(R, o.out, o.out, g, u.hmid, o.out, gr.h, 0, o.out, true);
        if (o.concat) {
          const int I0 = P.cdim + P.nd;
          ARDAE_TRY(launch_segment_sum(g, o.out, B, nz, o.out, 1.f, W.dB0, o.out, st));       // d rbs [B, out]
          ARDAE_TRY(launch_segment_sum(dh, o.out, B, nz, o.out, 1.f, W.dB1, o.out, st));      // d rba [B, out]
          pw(R, o.out, P.nd, g, noise, P.nd, gr.s, P.cdim, I0, true);                           // noise columns of W_01 (+ its bias)
          pw(B, o.out, P.cdim, W.dB0, W.inp, P.cdim, gr.s, 0, I0, false);                       // image columns of W_01
          pw(R, o.out, P.nd, dh, noise, P.nd, gr.a, P.cdim, I0, true);
          pw(B, o.out, P.cdim, W.dB1, W.inp, P.cdim, gr.a, 0, I0, false);
          ARDAE_TRY(flush_wgrad(probs, W.wscratch, W.wscratch_floats, st));
          LinArgs A{}; A.Y = W.dinp; A.ldY = P.cdim;
          ARDAE_TRY(lin2(EPI_ACT, ACT_NONE, B, P.cdim, W.dB0, o.out, o.out, packed + k.s_bi, W.dB1, o.out, o.out, packed + k.a_bi, A, st));
        } else {
          if (!b.same) pw(R, o.out, o.in, g, x, o.in, gr.s, 0, o.in, true);
          pw(R, o.out, o.in, dh, x, o.in, gr.a, 0, o.in, true);
          ARDAE_TRY(flush_wgrad(probs, W.wscratch, W.wscratch_floats, st));
          LinArgs A{}; A.Y = dx; A.ldY = o.in;
          if (!b.same) {
            ARDAE_TRY(lin2(EPI_ACT, ACT_NONE, R, o.in, g, o.out, o.out, packed + k.s.b, dh, o.out, o.out, packed + k.a.b, A, st));
          } else {   // identity skip: dx = dh W_0h + g
            ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, o.in, dh, o.out, o.out, packed + k.a.b, A, st));
            ARDAE_TRY(launch_axpy(g, (int64_t)R * o.in, 1.f, dx, st));
          }
        }
      } else {
        const Lin& l = o.l; const LinPk& k = hk.lk;
        const WnGrad gl{grads + l.w, grads + l.b, grads_beta};
        if (o.concat) {
          ARDAE_TRY(launch_segment_sum(g, o.out, B, nz, o.out, 1.f, W.dB0, o.out, st));       // d rba [B, out]
          pw(R, o.out, P.nd, g, noise, P.nd, gl, P.cdim, l.in, true);
          pw(B, o.out, P.cdim, W.dB0, W.inp, P.cdim, gl, 0, l.in, false);
          ARDAE_TRY(flush_wgrad(probs, W.wscratch, W.wscratch_floats, st));
          LinArgs A{}; A.Y = W.dinp; A.ldY = P.cdim;
          ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, P.cdim, W.dB0, o.out, o.out, packed + k.bi, A, st));
        } else {
          pw(R, o.out, o.in, g, x, o.in, gl, 0, l.in, true);
          ARDAE_TRY(flush_wgrad(probs, W.wscratch, W.wscratch_floats, st));
          LinArgs A{}; A.Y = dx; A.ldY = o.in;
          ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, o.in, g, o.out, o.out, packed + k.b, A, st));
        }
      }
      g = dx;
    }
  } else {
    const int ldn = P.nd + P.zd;
    auto plain_wgrad = [&](int M, const Lin& l, const float* G, const float* X, int ldX, int col0, int I, bool bias) {
      push_wgrad(probs, dummy, M, l.out, I, G, X, ldX, grads + l.w + col0, l.in, bias ? grads + l.b : nullptr);
      probs.back().beta = grads_beta;
    };
    // z = mu + exp(lv/2) eps,  lv = spm4(lvr)
    float* dlv = W.sc.g;           // [R, zd]
    ARDAE_TRY(launch_reparam_bwd(W.dzq, W.z, W.mu, R, P.zd, 1, dlv, st));
    RES_LAUNCH(spm4_kernel, (int64_t)R * P.zd, W.lvr, (const float*)dlv, dlv, (int64_t)R * P.zd, (int)P.clipped);
    plain_wgrad(R, P.mu, W.dzq, W.hh, P.cdim, 0, P.cdim, true);
    plain_wgrad(R, P.lv, dlv, W.hh, P.cdim, 0, P.cdim, true);
    float* dhh = W.dR0;            // [R, cdim]
    { LinArgs A{}; A.Y = dhh; A.ldY = P.cdim; ARDAE_TRY(lin2(EPI_ACT, ACT_NONE, R, P.cdim, W.dzq, P.zd, P.zd, packed + K.mu.b, dlv, P.zd, P.zd, packed + K.lv.b, A, st)); }
    RES_LAUNCH(mul_dact_kernel, (int64_t)R * P.cdim, dhh, W.hh, (int)ACT_ELU, dhh, (int64_t)R * P.cdim);
    ARDAE_TRY(launch_segment_sum(dhh, P.cdim, B, nz, P.cdim, 1.f, W.dB0, P.cdim, st));      // d rbh [B, cdim]
    plain_wgrad(R, P.efc, dhh, W.z0, P.nd, P.cdim, P.nd, true);                              // z0 columns + bias
    plain_wgrad(B, P.efc, W.dB0, W.inp, P.cdim, 0, P.cdim, false);                           // image columns
    float* dz0 = W.dR1;            // [R, nd] = dhh W_fc[:, cdim:]
    { LinArgs A{}; A.Y = dz0; A.ldY = P.nd; ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.nd, dhh, P.cdim, P.cdim, packed + K.efc.bn, A, st)); }
    // z0 = mu0[b] + exp(lv0[b] / 2) eps0,  lv0 = spm4(lv0r): reduce over the nz rows of an image first
    float* dlv0_rows = W.sc.dh;    // [R, nd]
    ARDAE_TRY(launch_reparam_bwd(dz0, W.z0, W.mu0, R, P.nd, nz, dlv0_rows, st, P.clipped ? 1.f : 0.f, noise, ldn));
    ARDAE_TRY(launch_segment_sum(dz0, P.nd, B, nz, P.nd, 1.f, W.dB1, P.nd, st));            // d mu0 [B, nd]
    ARDAE_TRY(launch_segment_sum(dlv0_rows, P.nd, B, nz, P.nd, 1.f, W.dB2, P.nd, st));      // d lv0 [B, nd]
    RES_LAUNCH(spm4_kernel, (int64_t)B * P.nd, W.lv0r, (const float*)W.dB2, W.dB2, (int64_t)B * P.nd, (int)P.clipped);
    plain_wgrad(B, P.mu0, W.dB1, W.inp, P.cdim, 0, P.cdim, true);
    plain_wgrad(B, P.lv0, W.dB2, W.inp, P.cdim, 0, P.cdim, true);
    ARDAE_TRY(flush_wgrad(probs, W.wscratch, W.wscratch_floats, st));
    // d inp = d rbh W_fc[:, :cdim] + d mu0 W_mu0 + d lv0r W_lv0
    float* part = W.dflat;         // [B, cdim] scratch (cdim <= 512)
    { LinArgs A{}; A.Y = part; A.ldY = P.cdim;
      ARDAE_TRY(lin2(EPI_ACT, ACT_NONE, B, P.cdim, W.dB0, P.cdim, P.cdim, packed + K.efc.bi, W.dB1, P.nd, P.nd, packed + K.mu0.b, A, st)); }
    { LinArgs A{}; A.S = part; A.ldS = P.cdim; A.Q = part; A.ldQ = P.cdim; A.Y = W.dinp; A.ldY = P.cdim;       // V * 1 + Q
      ARDAE_TRY(lin1(EPI_DACT, ACT_NONE, B, P.cdim, W.dB2, P.nd, P.nd, packed + K.lv0.b, A, st)); }
    (void)ldn;
  }
  (void)plain;
  // ---- trunk backward
  std::vector<BlkGrad> gt(P.trunk.size());
  for (size_t i = 0; i < P.trunk.size(); ++i) gt[i] = gm.blk(P.trunk[i], K.trunk[i]);
  ARDAE_TRY(blk_bwd(P.trunk[5], K.trunk[5], packed, W.flat, B, ACT_ELU, W.tb[5], W.dinp, W.dflat, W.sc, gt[5], W.wscratch, W.wscratch_floats, st));
  RES_LAUNCH(nhwc_nchw_kernel, (int64_t)B * 512, W.dflat, 16, 32, W.da, (int64_t)B * 512, 1);        // NCHW-flat gradient -> NHWC [B,4,4,32]
  float *cur = W.da, *nxt = W.dbuf;
  for (int i = 4; i >= 0; --i) {
    const float* xin = i == 0 ? W.x2 : W.tb[i - 1].out;
    ARDAE_TRY(blk_bwd(P.trunk[i], K.trunk[i], packed, xin, B, ACT_ELU, W.tb[i], cur, i == 0 ? nullptr : nxt, W.sc, gt[i], W.wscratch, W.wscratch_floats, st));
    std::swap(cur, nxt);
  }
  // ---- dL/dW -> dL/d(direction, scale), bias gradients
  for (size_t i0 = 0; i0 < gm.items.size(); i0 += WNB_MAX) {
    WnBwdBatch bt;
    bt.n = (int)std::min<size_t>(WNB_MAX, gm.items.size() - i0);
    bt.beta = grads_beta;
    for (int i = 0; i < bt.n; ++i) bt.it[i] = gm.items[i0 + i];
    hipLaunchKernelGGL(wn_backward_kernel, dim3(32, bt.n), dim3(256), 0, st, bt);
    ARDAE_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace ardae
