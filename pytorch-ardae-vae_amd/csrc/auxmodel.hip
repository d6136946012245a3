// MNISTAuxIPVAE (`--model auxmnist`, ardae_model_desc.kind == 3): the hierarchical implicit-posterior VAE of
// models/ivae/auxmnist.py:47-132 (Encoder = AuxEncoder + SimpleEncoder of models/vae/auxmnist.py:31-190 with enc_input = enc_noise =
// False and no log-variance clipping, as in the shipped recipe run_vae_dbmnist.sh) + models/vae/mnist.py Decoder, orchestrated from the
// K1 / K6w kernels like csrc/model.hip.  Oracle: oracle/ardae_oracle.py::aux_encode (pinned against the reference).
//
//   per image (B rows):   xs = 2x - 1;  h0 = MLP_aux(xs);  mu0 = M0 h0 + m0;  lv0 = L0 h0 + l0;  rb = Wx xs + b_1
//   per sample (R = B nz rows):  z0 = mu0[b] + exp(lv0[b] / 2) eps0;   t_1 = act(Wz z0 + rb[b]);  t_i = act(W_i t_{i-1} + b_i);
//                                 h = t_n;  mu = M h + m;  lv = L h + l;  z = mu + exp(lv / 2) eps
//   (the first encoder layer eats cat[xs, z0]: its image half is computed once per image and enters as a row bias - the
//    reference expands xs to R rows, vae/auxmnist.py:138-141)
// Noise layout of this kind: ONE [R, noise_dim + z_dim] tensor per sampler call, row = [eps0 | eps] (already scaled by std).
//
// kind 7, ToyAuxIPVAE (`--model auxmlp`, models/ivae/auxtoy.py:44-292 + models/vae/auxtoy.py + the Gaussian Decoder of models/vae/toy.py): the same
// networks WITHOUT the 2x - 1 rescale, a decoder with mean_fn / logvar_fn heads, and a SQUARE sampling scheme: the model-level calls take nz = q^2
// rows per image and run Encoder._forward with q = int(sqrt(nz)) (:215,230) - q z0's per image, the second stage on R = B q rows, q z's per
// z0: z [B q q, z] = mu[R][.] + exp(lv[R][.] / 2) eps.  Noise layout of kind 7: [eps0: R x noise_dim | eps: R q x z_dim], two blocks.
// Backward (closed form of the two reparameterisations):  dmu = dz, dlv = dz (z - mu) / 2;  dz0 = dt_1 Wz;
//   dmu0[b] = sum_nz dz0, dlv0[b] = sum_nz dz0 (z0 - mu0[b]) / 2.
#include <vector>

#include "auxmodel.h"
#include "elementwise.h"
#include "linear.h"
#include "wgrad.h"

namespace ardae {
namespace {

#define PACK_PUSH(W_, ldw_, nout_, k_, tr_, out_) pack_items__.push_back(PackItem{W_, ldw_, nout_, k_, (tr_) ? 1 : 0, out_})

struct Lin {
  size_t w, b;
  int out, in;
};

struct AuxLayout {
  int D, nd, h, zd, nl, act;
  int clip0, clip1;               // NormalDistribution.clip_logvar codes of the z0 / z heads (ardae_hip.h: ARDAE_MODEL_CLIP_*), 0 = none
  bool toy;                       // kind 7
  std::vector<Lin> am, ef, dec;   // aux_encode.main, encode.fc, decode.main: nl Linear each (nl - 1 hidden + fc, all followed by act)
  Lin mean0, logvar0, mean, logvar, logit, logvarx;   // logit: decode.reparam.logit_fn, or (toy) mean_fn followed by logvarx = logvar_fn
  size_t total = 0;
  // stage rows per image of a call with nz rows per image: nz, or (toy) q = sqrt(nz)
  int stage(int nz) const {
    if (!toy) return nz;
    int q = 1;
    while ((q + 1) * (q + 1) <= nz) ++q;
    return q;
  }
  explicit AuxLayout(const ardae_model_desc& d)
      : D(d.input_dim), nd(d.noise_dim), h(d.h_dim), zd(d.z_dim), nl(d.n_layers), act(d.act),
        clip0((d.flags >> ARDAE_MODEL_CLIP_Z0_SHIFT) & 15), clip1((d.flags >> ARDAE_MODEL_CLIP_Z_SHIFT) & 15), toy(d.kind == 7) {
    size_t off = 0;
    auto one = [&](int out, int in) {
      Lin l; l.out = out; l.in = in; l.w = off; off += (size_t)out * in; l.b = off; off += out;
      return l;
    };
    for (int l = 0; l < nl; ++l) am.push_back(one(h, l == 0 ? D : h));
    mean0 = one(nd, h); logvar0 = one(nd, h);
    for (int l = 0; l < nl; ++l) ef.push_back(one(h, l == 0 ? D + nd : h));
    mean = one(zd, h); logvar = one(zd, h);
    for (int l = 0; l < nl; ++l) dec.push_back(one(h, l == 0 ? zd : h));
    logit = one(D, h);
    if (toy) logvarx = one(D, h);
    total = off;
  }
};

struct AuxPacked {
  std::vector<size_t> am_f, am_b, ef_f, ef_b, dec_f, dec_b;   // ef_f[0] / ef_b[0]: the z0 half of the first encoder layer
  size_t efx_f, mean0_f, mean0_b, logvar0_f, logvar0_b, mean_f, mean_b, logvar_f, logvar_b, logit_f, logit_b, logvarx_f = 0, logvarx_b = 0;
  size_t total = 0;
  explicit AuxPacked(const AuxLayout& P) {
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 63) & ~size_t(63); return o; };
    for (auto& l : P.am) { am_f.push_back(take(packed_floats(l.out, l.in))); am_b.push_back(take(packed_floats(l.in, l.out))); }
    mean0_f = take(packed_floats(P.nd, P.h)); mean0_b = take(packed_floats(P.h, P.nd));
    logvar0_f = take(packed_floats(P.nd, P.h)); logvar0_b = take(packed_floats(P.h, P.nd));
    efx_f = take(packed_floats(P.h, P.D));
    for (int l = 0; l < P.nl; ++l) {
      const int in = l == 0 ? P.nd : P.h;
      ef_f.push_back(take(packed_floats(P.h, in))); ef_b.push_back(take(packed_floats(in, P.h)));
    }
    mean_f = take(packed_floats(P.zd, P.h)); mean_b = take(packed_floats(P.h, P.zd));
    logvar_f = take(packed_floats(P.zd, P.h)); logvar_b = take(packed_floats(P.h, P.zd));
    for (auto& l : P.dec) { dec_f.push_back(take(packed_floats(l.out, l.in))); dec_b.push_back(take(packed_floats(l.in, l.out))); }
    logit_f = take(packed_floats(P.D, P.h)); logit_b = take(packed_floats(P.h, P.D));
    if (P.toy) { logvarx_f = take(packed_floats(P.D, P.h)); logvarx_b = take(packed_floats(P.h, P.D)); }
    total = off;
  }
};

struct Bump {
  float* base; size_t cap; size_t off = 0; bool ok = true;
  Bump(float* b, size_t c) : base(b), cap(c) {}
  float* take(size_t n) {
    size_t o = off; off += (n + 63) & ~size_t(63);
    if (off > cap) { ok = false; return base; }
    return base + o;
  }
};
size_t al64(size_t n) { return (n + 63) & ~size_t(63); }

struct AuxWs {
  float *xs, *mu0, *lv0, *rb, *z0, *mu, *lv, *z, *zero;
  float *lv0r, *lvr;                    // the heads' raw outputs when a log-variance clip is on (lv0 / lv then hold the clipped values)
  std::vector<float*> e, t, dcd;        // e[l] [B,h] (l = 1..nl), t[i] [R,h], dcd[l] [R,h]
  float *o, *o2, *rec_row, *pri_row;    // decoder logits (toy: mean, o2 = logvar) and row losses
  // backward
  float *dox, *dox2, *dzq, *dz, *dlv, *dz0, *dlv0r, *drb, *dmu0, *dlv0;
  float *dmu_s, *dlv_s;                 // toy: dz / dlv summed over the q z's of a stage row, [R, zd]
  std::vector<float*> ddec, dt, de;
};

int wgrad_nprob(const AuxLayout& P) { return 1 + (P.toy ? 1 : 0) + P.nl + 2 + P.nl + 1 + 2 + P.nl; }

// R: stage rows (B nz, toy: B q); N: z / decoder rows (= R, toy: R q)
size_t wgrad_scratch(const AuxLayout& P, int B, int R, int N, std::vector<int>* splits_out) {
  // problem order must match aux_model_vae_backward
  std::vector<int> sp;
  size_t tot = 0;
  const int nprob = wgrad_nprob(P);
  auto one = [&](int M, int O, int I) {
    const int s = wgrad_splits(M, O, I, nprob);
    sp.push_back(s);
    tot += al64((size_t)s * O * I) + al64((size_t)s * 2 * O);
  };
  one(N, P.D, P.h);                                                  // logit head (toy: mean head)
  if (P.toy) one(N, P.D, P.h);                                       // toy: logvar head
  for (int l = 0; l < P.nl; ++l) one(N, P.h, l == 0 ? P.zd : P.h);   // decoder
  one(R, P.zd, P.h); one(R, P.zd, P.h);                              // mean, logvar
  for (int l = P.nl - 1; l >= 1; --l) one(R, P.h, P.h);              // encoder layers 2..n
  one(R, P.h, P.nd);                                                 // first encoder layer, z0 half (+ bias)
  one(B, P.h, P.D);                                                  // first encoder layer, image half
  one(B, P.nd, P.h); one(B, P.nd, P.h);                              // mean0, logvar0
  for (int l = 0; l < P.nl; ++l) one(B, P.h, l == 0 ? P.D : P.h);    // aux main
  if (splits_out) *splits_out = sp;
  return tot;
}

// mode 0: sampler only; 1: + decoder, losses, backward, weight gradients
void carve(const AuxLayout& P, Bump& ws, int B, int nz, int mode, AuxWs& W) {
  const size_t R = (size_t)B * P.stage(nz), N = (size_t)B * nz, h = P.h;
  W.xs = ws.take((size_t)B * P.D);
  W.e.assign(P.nl + 1, nullptr);
  for (int l = 1; l <= P.nl; ++l) W.e[l] = ws.take((size_t)B * h);
  W.mu0 = ws.take((size_t)B * P.nd); W.lv0 = ws.take((size_t)B * P.nd); W.rb = ws.take((size_t)B * h);
  W.z0 = ws.take(R * P.nd);
  W.t.assign(P.nl + 1, nullptr);
  for (int l = 1; l <= P.nl; ++l) W.t[l] = ws.take(R * h);
  W.mu = ws.take(R * P.zd); W.lv = ws.take(R * P.zd); W.z = ws.take(N * P.zd);
  W.zero = ws.take(R * P.nd + N * P.zd);
  W.lv0r = P.clip0 ? ws.take((size_t)B * P.nd) : W.lv0;
  W.lvr = P.clip1 ? ws.take(R * P.zd) : W.lv;
  if (mode == 0) return;
  W.dcd.assign(P.nl + 1, nullptr);
  for (int l = 1; l <= P.nl; ++l) W.dcd[l] = ws.take(N * h);
  W.o = ws.take(N * P.D); W.o2 = P.toy ? ws.take(N * P.D) : nullptr; W.rec_row = ws.take(N); W.pri_row = ws.take(N);
  W.dox = ws.take(N * P.D); W.dox2 = P.toy ? ws.take(N * P.D) : nullptr;
  W.dzq = ws.take(N * P.zd); W.dz = ws.take(N * P.zd); W.dlv = ws.take(N * P.zd);
  W.dmu_s = P.toy ? ws.take(R * P.zd) : nullptr; W.dlv_s = P.toy ? ws.take(R * P.zd) : nullptr;
  W.dz0 = ws.take(R * P.nd); W.dlv0r = ws.take(R * P.nd);
  W.drb = ws.take((size_t)B * h); W.dmu0 = ws.take((size_t)B * P.nd); W.dlv0 = ws.take((size_t)B * P.nd);
  W.ddec.assign(P.nl + 1, nullptr); W.dt.assign(P.nl + 1, nullptr); W.de.assign(P.nl + 1, nullptr);
  for (int l = 1; l <= P.nl; ++l) { W.ddec[l] = ws.take(N * h); W.dt[l] = ws.take(R * h); W.de[l] = ws.take((size_t)B * h); }
}

size_t workspace_floats(const AuxLayout& P, int B, int nz, int mode) {
  // dry run of carve() on a null arena
  Bump ws(nullptr, ~size_t(0));
  AuxWs W;
  carve(P, ws, B, nz, mode == 0 ? 0 : 1, W);
  size_t t = ws.off;
  if (mode != 0) t += wgrad_scratch(P, B, B * P.stage(nz), B * nz, nullptr);
  return t;
}

// out[r][c] = mu[g][c] + exp(lv[g][c] / 2) * eps[r][c],  g = r / rows_per_group  (models/ivae/auxmnist.py:33-41)
__global__ void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv, int ld_stat, const float* __restrict__ eps,
                                   int ld_eps, int64_t rows, int cols, int rows_per_group, float* __restrict__ out, float min_std,
                                   const float* __restrict__ raw, int ld_raw) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    const int64_t g = r / rows_per_group;
    const float e = eps[r * ld_eps + c];
    float v = mu[g * ld_stat + c] + __expf(0.5f * lv[g * ld_stat + c]) * e;
    if (min_std != 0.f) v += min_std * (raw ? raw[r * ld_raw + c] : e);
    out[i] = v;
  }
}
// dlv[r][c] = dz[r][c] * (z[r][c] - mu[g][c]) / 2    (z - mu = exp(lv / 2) eps: d z / d lv = (z - mu) / 2; with min_std: z - mu - min_std eps)
__global__ void reparam_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ z, const float* __restrict__ mu, int64_t rows,
                                   int cols, int rows_per_group, float* __restrict__ dlv, float min_std, const float* __restrict__ eps, int ld_eps) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    float d = z[i] - mu[(r / rows_per_group) * cols + c];
    if (min_std != 0.f) d -= min_std * eps[r * ld_eps + c];
    dlv[i] = 0.5f * dz[i] * d;
  }
}
int grid_of(int64_t n) {
  const int64_t g = (n + 255) / 256;
  return (int)(g < 4096 ? g : 4096);
}
}  // namespace
// NormalDistribution.clip_logvar (models/reparam.py:17-41) on a head's raw output x: y = c(x), or (dy given) the backward dx = dy c'(x).
// codes: 1 'hard' clamp to [MIN_LOGVAR, MAX_LOGVAR] = [-4, 2] (:7-8; torch.max / torch.min pass the gradient where x is strictly inside),
// 2 'softplus', 3 .. 8 'spmK' = softplus(x + K) - K for K = 10, 6, 5, 4, 3, 2, 9 'tanh', 10 '2tanh'
__global__ void logvar_clip_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ y, int64_t n, int code) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const float v = x[e];
  float f, d;
  if (code == 1) {
    f = fminf(fmaxf(v, -4.f), 2.f);
    d = (v > -4.f && v < 2.f) ? 1.f : ((v == -4.f || v == 2.f) ? 0.5f : 0.f);      // (ties: torch.max / min split the gradient evenly)
  } else if (code <= 8) {
    const float k = code == 2 ? 0.f : code == 3 ? 10.f : code == 4 ? 6.f : code == 5 ? 5.f : code == 6 ? 4.f : code == 7 ? 3.f : 2.f;
    const float u = v + k;
    f = (u > 20.f ? u : log1pf(expf(u))) - k;
    d = 1.f / (1.f + expf(-u));
  } else {
    const float t = tanhf(v), s = code == 9 ? 1.f : 2.f;
    f = s * t;
    d = s * (1.f - t * t);
  }
  y[e] = dy ? dy[e] * d : f;
}
int launch_logvar_clip(const float* x, const float* dy, float* y, int64_t n, int code, hipStream_t st) {
  if (code == 0 || n == 0) return 0;
  hipLaunchKernelGGL(logvar_clip_kernel, dim3(grid_of(n)), dim3(256), 0, st, x, dy, y, n, code);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

int launch_reparam_fwd(const float* mu, const float* lv, const float* eps, int ld_eps, int64_t rows, int cols, int rpg, float* out, hipStream_t st,
                       float min_std, const float* raw, int ld_raw) {
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(grid_of(rows * cols)), dim3(256), 0, st, mu, lv, cols, eps, ld_eps, rows, cols, rpg, out, min_std, raw, ld_raw);
  ARDAE_LAUNCH_CHECK();
  return 0;
}
int launch_reparam_bwd(const float* dz, const float* z, const float* mu, int64_t rows, int cols, int rpg, float* dlv, hipStream_t st, float min_std,
                       const float* eps, int ld_eps) {
  ARDAE_CHECK_ARG(min_std == 0.f || eps, "reparam_bwd: min_std needs the forward draw");
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid_of(rows * cols)), dim3(256), 0, st, dz, z, mu, rows, cols, rpg, dlv, min_std, eps, ld_eps);
  ARDAE_LAUNCH_CHECK();
  return 0;
}

namespace {
int lin1(int epi, int act, int M, int Nout, const float* x, int ldx, int K, const float* wp, LinArgs a, hipStream_t st) {
  a.M = M; a.Nout = Nout; a.nsrc = 1; a.act = act;
  a.src[0].x = x; a.src[0].ld = ldx; a.src[0].K = K; a.src[0].wp = wp;
  return launch_linear(a, epi, st);
}

// the sampler on R = B nz rows; noise [R, nd + zd] (never null here); fills every forward field of W
// (toy: R = B q stage rows, z on B q q rows; noise = [eps0: R x nd | eps: R q x zd])
int sampler_fwd(const AuxLayout& P, const AuxPacked& K, const float* params, const float* packed, const float* x, const float* noise, int B,
                int nz_rows, AuxWs& W, hipStream_t st) {
  const int nz = P.stage(nz_rows);               // samples per image at the stage level
  const int R = B * nz, h = P.h, act = P.act, nl = P.nl;
  const int ld0 = P.toy ? P.nd : P.nd + P.zd, lde = P.toy ? P.zd : P.nd + P.zd;
  const float* eps = P.toy ? noise + (size_t)R * P.nd : noise + P.nd;
  if (P.toy) ARDAE_TRY(launch_copy(x, (int64_t)B * P.D, W.xs, st));      // no rescale (models/vae/auxtoy.py: the `x = 2*x - 1` lines are gone)
  else ARDAE_TRY(launch_affine(x, (int64_t)B * P.D, 2.f, -1.f, W.xs, st));
  for (int l = 1; l <= nl; ++l) {
    LinArgs A{}; A.bias = params + P.am[l - 1].b; A.Y = W.e[l]; A.ldY = h;
    ARDAE_TRY(lin1(EPI_ACT, act, B, h, l == 1 ? W.xs : W.e[l - 1], l == 1 ? P.D : h, P.am[l - 1].in, packed + K.am_f[l - 1], A, st));
  }
  {
    LinArgs A{}; A.bias = params + P.mean0.b; A.Y = W.mu0; A.ldY = P.nd;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, P.nd, W.e[nl], h, h, packed + K.mean0_f, A, st));
    LinArgs A2{}; A2.bias = params + P.logvar0.b; A2.Y = W.lv0r; A2.ldY = P.nd;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, P.nd, W.e[nl], h, h, packed + K.logvar0_f, A2, st));
    ARDAE_TRY(launch_logvar_clip(W.lv0r, nullptr, W.lv0, (int64_t)B * P.nd, P.clip0, st));
    LinArgs A3{}; A3.bias = params + P.ef[0].b; A3.Y = W.rb; A3.ldY = h;   // image half of the first encoder layer (+ its bias)
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, B, h, W.xs, P.D, P.D, packed + K.efx_f, A3, st));
  }
  ARDAE_TRY(launch_reparam_fwd(W.mu0, W.lv0, noise, ld0, R, P.nd, nz, W.z0, st));
  for (int l = 1; l <= nl; ++l) {
    LinArgs A{}; A.Y = W.t[l]; A.ldY = h;
    if (l == 1) { A.rowbias = W.rb; A.rowbias_ld = h; A.rows_per_group = nz; }
    else A.bias = params + P.ef[l - 1].b;
    ARDAE_TRY(lin1(EPI_ACT, act, R, h, l == 1 ? W.z0 : W.t[l - 1], l == 1 ? P.nd : h, l == 1 ? P.nd : h, packed + K.ef_f[l - 1], A, st));
  }
  {
    LinArgs A{}; A.bias = params + P.mean.b; A.Y = W.mu; A.ldY = P.zd;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.zd, W.t[nl], h, h, packed + K.mean_f, A, st));
    LinArgs A2{}; A2.bias = params + P.logvar.b; A2.Y = W.lvr; A2.ldY = P.zd;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.zd, W.t[nl], h, h, packed + K.logvar_f, A2, st));
    ARDAE_TRY(launch_logvar_clip(W.lvr, nullptr, W.lv, (int64_t)R * P.zd, P.clip1, st));
  }
  if (P.toy) return launch_reparam_fwd(W.mu, W.lv, eps, lde, (int64_t)R * nz, P.zd, nz, W.z, st);      // q z's per stage row
  return launch_reparam_fwd(W.mu, W.lv, eps, lde, R, P.zd, 1, W.z, st);
}

// noise of a call with nz rows per image, or the zero block of a std = 0 pass
const float* noise_or_zero(const AuxLayout& P, const float* noise, int B, int nz, AuxWs& W, hipStream_t st, int& rc) {
  rc = 0;
  if (noise) return noise;
  if (rc == 0) rc = launch_fill(W.zero, (size_t)B * P.stage(nz) * P.nd + (size_t)B * nz * P.zd, 0.f, st);
  return W.zero;
}

}  // namespace

size_t aux_model_param_floats(const ardae_model_desc& d) { return AuxLayout(d).total; }
size_t aux_model_packed_floats(const ardae_model_desc& d) { return AuxPacked(AuxLayout(d)).total; }
size_t aux_model_workspace_floats(const ardae_model_desc& d, int B, int nz, int mode) {
  const AuxLayout P(d);
  if (mode == 2) return (size_t)P.nl * al64((size_t)B * nz * P.h);   // decode only
  return workspace_floats(P, B, nz, mode == 3 ? 0 : mode);
}

int aux_model_pack(const ardae_model_desc& d, const float* params, float* packed, hipStream_t st) {
  const AuxLayout P(d);
  const AuxPacked K(P);
  std::vector<PackItem> pack_items__;
  auto both = [&](const Lin& l, size_t f, size_t b) {
    PACK_PUSH(params + l.w, l.in, l.out, l.in, false, packed + f);
    PACK_PUSH(params + l.w, l.in, l.in, l.out, true, packed + b);
  };
  for (int l = 0; l < P.nl; ++l) both(P.am[l], K.am_f[l], K.am_b[l]);
  both(P.mean0, K.mean0_f, K.mean0_b); both(P.logvar0, K.logvar0_f, K.logvar0_b);
  const Lin& e0 = P.ef[0];                                            // [h, D + nd]: image half | z0 half
  PACK_PUSH(params + e0.w, e0.in, P.h, P.D, false, packed + K.efx_f);
  PACK_PUSH(params + e0.w + P.D, e0.in, P.h, P.nd, false, packed + K.ef_f[0]);
  PACK_PUSH(params + e0.w + P.D, e0.in, P.nd, P.h, true, packed + K.ef_b[0]);
  for (int l = 1; l < P.nl; ++l) both(P.ef[l], K.ef_f[l], K.ef_b[l]);
  both(P.mean, K.mean_f, K.mean_b); both(P.logvar, K.logvar_f, K.logvar_b);
  for (int l = 0; l < P.nl; ++l) both(P.dec[l], K.dec_f[l], K.dec_b[l]);
  both(P.logit, K.logit_f, K.logit_b);
  if (P.toy) both(P.logvarx, K.logvarx_f, K.logvarx_b);
  return launch_pack_batch(pack_items__.data(), (int)pack_items__.size(), st);
}

int aux_model_encode(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                     float* workspace, size_t wsf, float* z_out, float* hidden_out, hipStream_t st) {
  const AuxLayout P(d);
  const AuxPacked K(P);
  Bump ws(workspace, wsf);
  AuxWs W;
  carve(P, ws, B, nz, 0, W);
  ARDAE_CHECK_ARG(ws.ok, "aux_model_encode: workspace too small");
  int rc;
  ARDAE_CHECK_ARG(!P.toy || P.stage(nz) * P.stage(nz) == nz, "aux_model_encode: ToyAuxIPVAE draws q z0's x q z's per image - nz (%d) must be a square", nz);
  const float* nz_ptr = noise_or_zero(P, noise, B, nz, W, st, rc);
  ARDAE_TRY(rc);
  ARDAE_TRY(sampler_fwd(P, K, params, packed, x, nz_ptr, B, nz, W, st));
  if (z_out) ARDAE_TRY(launch_copy(W.z, (size_t)B * nz * P.zd, z_out, st));
  if (hidden_out) {   // forward_hidden of the ENCODER (ivae/auxmnist.py:125-132, nz == 1): cat(h0, h)
    ARDAE_CHECK_ARG(nz == 1, "aux_model_encode: the hidden context is defined for nz == 1");
    ARDAE_TRY(launch_copy2d(W.e[P.nl], (size_t)P.h, hidden_out, 2 * (size_t)P.h, B, (size_t)P.h, st));
    ARDAE_TRY(launch_copy2d(W.t[P.nl], (size_t)P.h, hidden_out + P.h, 2 * (size_t)P.h, B, (size_t)P.h, st));
  }
  return 0;
}

int aux_model_decode(const ardae_model_desc& d, const float* params, const float* packed, const float* z, int R, float* workspace, size_t wsf,
                     float* out0, hipStream_t st, float* out1) {
  const AuxLayout P(d);
  const AuxPacked K(P);
  Bump ws(workspace, wsf);
  const float* cur = z;
  for (int l = 1; l <= P.nl; ++l) {
    float* nxt = ws.take((size_t)R * P.h);
    LinArgs A{}; A.bias = params + P.dec[l - 1].b; A.Y = nxt; A.ldY = P.h;
    ARDAE_TRY(lin1(EPI_ACT, P.act, R, P.h, cur, l == 1 ? P.zd : P.h, P.dec[l - 1].in, packed + K.dec_f[l - 1], A, st));
    cur = nxt;
  }
  ARDAE_CHECK_ARG(ws.ok, "aux_model_decode: workspace too small");
  LinArgs A{}; A.bias = params + P.logit.b; A.Y = out0; A.ldY = P.D;
  ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.D, cur, P.h, P.h, packed + K.logit_f, A, st));
  if (P.toy) {
    ARDAE_CHECK_ARG(out1, "aux_model_decode: the Gaussian decoder returns mean (out0) and logvar (out1)");
    LinArgs A2{}; A2.bias = params + P.logvarx.b; A2.Y = out1; A2.ldY = P.D;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.D, cur, P.h, P.h, packed + K.logvarx_f, A2, st));
  }
  return 0;
}

int aux_model_vae_forward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                          float beta, float* workspace, size_t wsf, float* z_out, float* losses, hipStream_t st) {
  const AuxLayout P(d);
  const AuxPacked K(P);
  Bump ws(workspace, wsf);
  AuxWs W;
  carve(P, ws, B, nz, 1, W);
  ARDAE_CHECK_ARG(ws.ok, "aux_model_vae_forward: workspace too small");
  ARDAE_CHECK_ARG(!P.toy || P.stage(nz) * P.stage(nz) == nz, "aux_model_vae_forward: ToyAuxIPVAE needs a square nz (got %d)", nz);
  const int R = B * nz, h = P.h;      // z / decoder rows
  ARDAE_TRY(sampler_fwd(P, K, params, packed, x, noise, B, nz, W, st));
  ARDAE_TRY(launch_copy(W.z, (size_t)R * P.zd, z_out, st));
  for (int l = 1; l <= P.nl; ++l) {
    LinArgs A{}; A.bias = params + P.dec[l - 1].b; A.Y = W.dcd[l]; A.ldY = h;
    ARDAE_TRY(lin1(EPI_ACT, P.act, R, h, l == 1 ? W.z : W.dcd[l - 1], l == 1 ? P.zd : h, P.dec[l - 1].in, packed + K.dec_f[l - 1], A, st));
  }
  {
    LinArgs A{}; A.bias = params + P.logit.b; A.Y = W.o; A.ldY = P.D;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.D, W.dcd[P.nl], h, h, packed + K.logit_f, A, st));
  }
  if (P.toy) {
    LinArgs A{}; A.bias = params + P.logvarx.b; A.Y = W.o2; A.ldY = P.D;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.D, W.dcd[P.nl], h, h, packed + K.logvarx_f, A, st));
  }
  ARDAE_TRY(launch_vae_loss(P.toy ? 1 : 0, W.o, W.o2, x, W.z, R, nz, P.D, P.zd, beta, 0, 0.f, nullptr, W.rec_row, W.pri_row, nullptr, nullptr, nullptr, st));
  return launch_vae_loss_finalize(W.rec_row, W.pri_row, R, beta, losses, st);
}

int aux_model_vae_backward(const ardae_model_desc& d, const float* params, const float* packed, const float* x, const float* noise, int B, int nz,
                           float beta, float dloss, const float* dz_extra, float* workspace, size_t wsf, float* grads, float grads_beta,
                           hipStream_t st) {
  (void)noise;
  const AuxLayout P(d);
  const AuxPacked K(P);
  Bump ws(workspace, wsf);
  AuxWs W;
  carve(P, ws, B, nz, 1, W);
  const int N = B * nz;                                   // z / decoder rows
  const int nzs = P.stage(nz);                            // samples per image at the stage level (toy: q)
  const int R = B * nzs, h = P.h, act = P.act, nl = P.nl;
  const float gscale = dloss / (float)N;
  ARDAE_TRY(launch_vae_loss(P.toy ? 1 : 0, W.o, W.o2, x, W.z, N, nz, P.D, P.zd, beta, 1, gscale, dz_extra, W.rec_row, W.pri_row, W.dox, W.dox2, W.dzq, st));
  // decoder backward
  if (P.toy) {      // two heads (mean_fn, logvar_fn) back into the last hidden layer
    LinArgs A{}; A.M = N; A.Nout = h; A.nsrc = 2; A.act = act; A.S = W.dcd[nl]; A.ldS = h; A.Y = W.ddec[nl]; A.ldY = h;
    A.src[0].x = W.dox; A.src[0].ld = P.D; A.src[0].K = P.D; A.src[0].wp = packed + K.logit_b;
    A.src[1].x = W.dox2; A.src[1].ld = P.D; A.src[1].K = P.D; A.src[1].wp = packed + K.logvarx_b;
    ARDAE_TRY(launch_linear(A, EPI_DACT, st));
  } else {
    LinArgs A{}; A.S = W.dcd[nl]; A.ldS = h; A.Y = W.ddec[nl]; A.ldY = h;
    ARDAE_TRY(lin1(EPI_DACT, act, N, h, W.dox, P.D, P.D, packed + K.logit_b, A, st));
  }
  for (int l = nl; l >= 2; --l) {
    LinArgs A{}; A.S = W.dcd[l - 1]; A.ldS = h; A.Y = W.ddec[l - 1]; A.ldY = h;
    ARDAE_TRY(lin1(EPI_DACT, act, N, h, W.ddec[l], h, h, packed + K.dec_b[l - 1], A, st));
  }
  {  // dz = ddec_1 . D_1 + (prior + injected seed)
    LinArgs A{}; A.S = W.dzq; A.ldS = P.zd; A.Q = W.dzq; A.ldQ = P.zd; A.Y = W.dz; A.ldY = P.zd;
    ARDAE_TRY(lin1(EPI_DACT, ACT_NONE, N, P.zd, W.ddec[1], h, h, packed + K.dec_b[0], A, st));
  }
  // second reparameterisation: dmu = dz, dlv = dz (z - mu) / 2; both heads back into h = t_n
  // (toy: q z's share a stage row's mu / lv: their dz and dlv are summed over the q rows first)
  ARDAE_TRY(launch_reparam_bwd(W.dz, W.z, W.mu, N, P.zd, P.toy ? nzs : 1, W.dlv, st));
  const float* dmu = W.dz;
  const float* dlv = W.dlv;
  if (P.toy) {
    ARDAE_TRY(launch_segment_sum(W.dz, P.zd, R, nzs, P.zd, 1.0f, W.dmu_s, P.zd, st));
    ARDAE_TRY(launch_segment_sum(W.dlv, P.zd, R, nzs, P.zd, 1.0f, W.dlv_s, P.zd, st));
    dmu = W.dmu_s; dlv = W.dlv_s;
  }
  // through the z head's log-variance clip: d lv_raw = d lv c'(lv_raw) (in place; one value per stage row)
  ARDAE_TRY(launch_logvar_clip(W.lvr, dlv, const_cast<float*>(dlv), (int64_t)R * P.zd, P.clip1, st));
  {
    LinArgs A{}; A.M = R; A.Nout = h; A.nsrc = 2; A.act = act; A.S = W.t[nl]; A.ldS = h; A.Y = W.dt[nl]; A.ldY = h;
    A.src[0].x = dmu; A.src[0].ld = P.zd; A.src[0].K = P.zd; A.src[0].wp = packed + K.mean_b;
    A.src[1].x = dlv; A.src[1].ld = P.zd; A.src[1].K = P.zd; A.src[1].wp = packed + K.logvar_b;
    ARDAE_TRY(launch_linear(A, EPI_DACT, st));
  }
  for (int l = nl; l >= 2; --l) {
    LinArgs A{}; A.S = W.t[l - 1]; A.ldS = h; A.Y = W.dt[l - 1]; A.ldY = h;
    ARDAE_TRY(lin1(EPI_DACT, act, R, h, W.dt[l], h, h, packed + K.ef_b[l - 1], A, st));
  }
  {  // dz0 = dt_1 Wz  (no activation between z0 and the layer)
    LinArgs A{}; A.Y = W.dz0; A.ldY = P.nd;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.nd, W.dt[1], h, h, packed + K.ef_b[0], A, st));
  }
  ARDAE_TRY(launch_segment_sum(W.dt[1], h, B, nzs, h, 1.0f, W.drb, h, st));
  // first reparameterisation, reduced over the nz samples of each image
  ARDAE_TRY(launch_reparam_bwd(W.dz0, W.z0, W.mu0, R, P.nd, nzs, W.dlv0r, st));
  ARDAE_TRY(launch_segment_sum(W.dz0, P.nd, B, nzs, P.nd, 1.0f, W.dmu0, P.nd, st));
  ARDAE_TRY(launch_segment_sum(W.dlv0r, P.nd, B, nzs, P.nd, 1.0f, W.dlv0, P.nd, st));
  ARDAE_TRY(launch_logvar_clip(W.lv0r, W.dlv0, W.dlv0, (int64_t)B * P.nd, P.clip0, st));      // ... and through the z0 head's
  {
    LinArgs A{}; A.M = B; A.Nout = h; A.nsrc = 2; A.act = act; A.S = W.e[nl]; A.ldS = h; A.Y = W.de[nl]; A.ldY = h;
    A.src[0].x = W.dmu0; A.src[0].ld = P.nd; A.src[0].K = P.nd; A.src[0].wp = packed + K.mean0_b;
    A.src[1].x = W.dlv0; A.src[1].ld = P.nd; A.src[1].K = P.nd; A.src[1].wp = packed + K.logvar0_b;
    ARDAE_TRY(launch_linear(A, EPI_DACT, st));
  }
  for (int l = nl; l >= 2; --l) {
    LinArgs A{}; A.S = W.e[l - 1]; A.ldS = h; A.Y = W.de[l - 1]; A.ldY = h;
    ARDAE_TRY(lin1(EPI_DACT, act, B, h, W.de[l], h, h, packed + K.am_b[l - 1], A, st));
  }
  // weight gradients: one batched launch (order == wgrad_scratch)
  std::vector<int> splits;
  wgrad_scratch(P, B, R, N, &splits);
  std::vector<WgradProblem> probs;
  auto push = [&](int M, int O, int I, const float* G, const float* X, int ldX, float* out, int ldout, float* out_bias) {
    WgradProblem p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.O = O; p.I = I; p.npairs = 1;
    p.G[0] = G; p.ldG[0] = O; p.X[0] = X; p.ldX[0] = ldX;
    p.bias_pair = out_bias ? 0 : -1;
    p.splits = splits[probs.size()];
    p.partial = ws.take((size_t)p.splits * O * I);
    p.partial_vec = ws.take((size_t)p.splits * 2 * O);
    p.out = out; p.ldout = ldout; p.out_bias = out_bias; p.beta = grads_beta;
    probs.push_back(p);
  };
  push(N, P.D, h, W.dox, W.dcd[nl], h, grads + P.logit.w, h, grads + P.logit.b);
  if (P.toy) push(N, P.D, h, W.dox2, W.dcd[nl], h, grads + P.logvarx.w, h, grads + P.logvarx.b);
  for (int l = 1; l <= nl; ++l)
    push(N, h, P.dec[l - 1].in, W.ddec[l], l == 1 ? W.z : W.dcd[l - 1], l == 1 ? P.zd : h, grads + P.dec[l - 1].w, P.dec[l - 1].in, grads + P.dec[l - 1].b);
  push(R, P.zd, h, dmu, W.t[nl], h, grads + P.mean.w, h, grads + P.mean.b);
  push(R, P.zd, h, dlv, W.t[nl], h, grads + P.logvar.w, h, grads + P.logvar.b);
  for (int l = nl; l >= 2; --l) push(R, h, h, W.dt[l], W.t[l - 1], h, grads + P.ef[l - 1].w, h, grads + P.ef[l - 1].b);
  push(R, h, P.nd, W.dt[1], W.z0, P.nd, grads + P.ef[0].w + P.D, P.ef[0].in, grads + P.ef[0].b);
  push(B, h, P.D, W.drb, W.xs, P.D, grads + P.ef[0].w, P.ef[0].in, nullptr);
  push(B, P.nd, h, W.dmu0, W.e[nl], h, grads + P.mean0.w, h, grads + P.mean0.b);
  push(B, P.nd, h, W.dlv0, W.e[nl], h, grads + P.logvar0.w, h, grads + P.logvar0.b);
  for (int l = 1; l <= nl; ++l)
    push(B, h, P.am[l - 1].in, W.de[l], l == 1 ? W.xs : W.e[l - 1], l == 1 ? P.D : h, grads + P.am[l - 1].w, P.am[l - 1].in, grads + P.am[l - 1].b);
  ARDAE_CHECK_ARG(ws.ok, "aux_model_vae_backward: workspace too small");
  return launch_wgrad_batch(probs.data(), (int)probs.size(), st);
}

}  // namespace ardae
