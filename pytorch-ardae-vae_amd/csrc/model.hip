// Implicit-posterior VAE (sampler + decoder + ELBO pieces) on gfx950, orchestrated from the K1 / K6w kernels.
//
// Reference: models/ivae/mnist.py (kind 0) and models/ivae/toy.py with enc_type='concat' (kind 1).  Both are
// described by one structure:
//   inp  MLP : n_inp Linear->act on B rows                     (mnist: input is 2x-1, n_inp = n_layers+2; toy: n_layers)
//   stack    : n_stack Linear over [hidden | noise] on R = B*nz rows, act on all but the last; the FIRST one's hidden
//              part is the per-image `inp` (computed once per image and added as a row bias - the reference expands it to
//              R rows, ivae/mnist.py:115-116), the noise part is missing where the reference does not concatenate
//              (mnist: stack = [h <- h|noise, z <- h];  toy: every layer h|noise, models/layers.py:717-722)
//   decoder  : n_dec Linear->act from z, then 1 (logits) or 2 (mean, logvar) linear heads
#include <vector>

#include "ardae_hip.h"
#include "common.h"
#include "auxmodel.h"
#include "convmodel.h"
#include "resmodel.h"
#include "elementwise.h"
#include "linear.h"
#include "wgrad.h"

namespace ardae {
namespace {

// collect pack requests; flushed with one launch by PACK_FLUSH
#define PACK_PUSH(W_, ldw_, nout_, k_, tr_, out_) pack_items__.push_back(PackItem{W_, ldw_, nout_, k_, (tr_) ? 1 : 0, out_})
#define PACK_FLUSH(st_) ARDAE_TRY(launch_pack_batch(pack_items__.data(), (int)pack_items__.size(), st_))

struct Lin {
  size_t w, b;
  int out, in;
};

struct ModelLayout {
  int kind, D, nd, h, zd, nl, act;
  std::vector<Lin> inp, stack, dec, heads;
  std::vector<bool> stack_noise;   // does stack[i] take the noise concat?
  size_t total = 0;
  explicit ModelLayout(const ardae_model_desc& d)
      : kind(d.kind), D(d.input_dim), nd(d.noise_dim), h(d.h_dim), zd(d.z_dim), nl(d.n_layers), act(d.act) {
    size_t off = 0;
    auto add = [&](std::vector<Lin>& v, int out, int in) {
      Lin l; l.out = out; l.in = in; l.w = off; off += (size_t)out * in; l.b = off; off += out;
      v.push_back(l);
    };
    const int n_inp = kind == 0 ? nl + 2 : nl;
    for (int l = 0; l < n_inp; ++l) add(inp, h, l == 0 ? D : h);
    if (kind == 0) {
      add(stack, h, h + nd); stack_noise.push_back(true);
      add(stack, zd, h); stack_noise.push_back(false);
    } else {
      for (int l = 0; l < nl; ++l) { add(stack, h, h + nd); stack_noise.push_back(true); }
      add(stack, zd, h + nd); stack_noise.push_back(true);
    }
    const int n_dec = kind == 0 ? nl + 1 : nl;
    for (int l = 0; l < n_dec; ++l) add(dec, h, l == 0 ? zd : h);
    add(heads, D, h);
    if (kind == 1) add(heads, D, h);
    total = off;
  }
};

struct ModelPacked {
  std::vector<size_t> inp_f, inp_b, sh_f, sh_b, sn_f, dec_f, dec_b, head_f, head_b;
  size_t total = 0;
  explicit ModelPacked(const ModelLayout& P) {
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 63) & ~size_t(63); return o; };
    for (auto& l : P.inp) { inp_f.push_back(take(packed_floats(l.out, l.in))); inp_b.push_back(take(packed_floats(l.in, l.out))); }
    for (size_t i = 0; i < P.stack.size(); ++i) {
      sh_f.push_back(take(packed_floats(P.stack[i].out, P.h)));
      sh_b.push_back(take(packed_floats(P.h, P.stack[i].out)));
      sn_f.push_back(P.stack_noise[i] ? take(packed_floats(P.stack[i].out, P.nd)) : 0);
    }
    for (auto& l : P.dec) { dec_f.push_back(take(packed_floats(l.out, l.in))); dec_b.push_back(take(packed_floats(l.in, l.out))); }
    for (auto& l : P.heads) { head_f.push_back(take(packed_floats(l.out, l.in))); head_b.push_back(take(packed_floats(l.in, l.out))); }
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

int desc_ok(const ardae_model_desc* d) {
  ARDAE_CHECK_ARG(d != nullptr, "model: desc is NULL");
  ARDAE_CHECK_ARG(d->kind >= 0 && d->kind <= 7,
                  "model: kind must be 0 (MNISTIPVAE), 1 (ToyIPVAE concat), 2 (ConvIPVAE), 3 (MNISTAuxIPVAE), 4 (MNISTConvAuxIPVAE), 5 (ResConvIPVAE) or "
                  "6 (MNISTResConvAuxIPVAE) or 7 (ToyAuxIPVAE)");
  ARDAE_CHECK_ARG((d->flags & ~(ARDAE_MODEL_NO_CENTER | ARDAE_MODEL_HEAD_MASK | ARDAE_MODEL_CLIPPED | ARDAE_MODEL_CLIP_MASK)) == 0 &&
                      ((d->kind == 5 || d->kind == 6) ? (d->flags & ARDAE_MODEL_CLIP_MASK) == 0
                                                      : ((d->kind == 3 || d->kind == 7) ? (d->flags & ~ARDAE_MODEL_CLIP_MASK) == 0 : d->flags == 0)),
                  "model: unknown flags %d (ARDAE_MODEL_NO_CENTER / the sampler-head bits / ARDAE_MODEL_CLIPPED: residual-conv kinds 5 / 6 only; "
                  "the log-variance clip codes: kinds 3 / 7 only)", d->flags);
  ARDAE_CHECK_ARG(((d->flags >> ARDAE_MODEL_CLIP_Z0_SHIFT) & 15) <= 10 && ((d->flags >> ARDAE_MODEL_CLIP_Z_SHIFT) & 15) <= 10, "model: unknown log-variance clip code");
  if ((d->kind == 5 || d->kind == 6)) {
    ARDAE_CHECK_ARG(d->input_dim == 784 && d->noise_dim >= 1 && d->z_dim >= 1 && d->h_dim >= 1 && d->act == ACT_ELU && (d->kind == 6 || (d->n_layers >= 1 && d->n_layers <= 4)),
                    "model: the residual-conv models are 28x28x1, ELU, and (kind 5) 1 .. 4 hidden layers in the sampler head");
    return 0;
  }
  if (d->kind == 2 || d->kind == 4) {
    ARDAE_CHECK_ARG(d->input_dim == 784 && d->noise_dim >= 1 && d->z_dim >= 1, "model: ConvIPVAE is hard-wired to 28x28x1 inputs (input_dim 784)");
    ARDAE_CHECK_ARG(d->act > ACT_NONE && d->act <= ACT_LAST, "model: unknown activation %d (relu, softplus, elu, tanh, leaky_relu, swish)", d->act);
    return 0;
  }
  ARDAE_CHECK_ARG(d->input_dim >= 1 && d->noise_dim >= 1 && d->h_dim >= 1 && d->z_dim >= 1 && d->n_layers >= 1 && d->n_layers <= 4,
                  "model: bad dimensions");
  ARDAE_CHECK_ARG(d->act > ACT_NONE && d->act <= ACT_LAST, "model: unknown activation %d (relu, softplus, elu, tanh, leaky_relu, swish)", d->act);
  return 0;
}

// saved activations (in `workspace`, same carving in forward and backward)
struct ModelWs {
  float* x2;
  std::vector<float*> e;    // e[l], l = 1..n_inp        [B,h]
  float* rb;                // per-image part of the first stack layer (+ its bias) [B,h]
  std::vector<float*> t;    // t[i], i = 1..n_stack-1    [R,h]
  float* z;                 // sampler output            [R,zd]
  std::vector<float*> dcd;  // dcd[l], l = 1..n_dec      [R,h]
  std::vector<float*> o;    // heads                     [R,D]
  float *rec_row, *pri_row;
  // backward only
  std::vector<float*> dox, ddec, dt, de;
  float *dzq, *dz, *drb;
};

size_t al64(size_t n) { return (n + 63) & ~size_t(63); }

size_t wgrad_scratch(const ModelLayout& P, int B, int R, std::vector<int>* splits_out) {
  std::vector<int> sp;
  size_t tot = 0;
  const int nprob = (int)(P.heads.size() + P.dec.size() + 2 * P.stack.size() + P.inp.size());
  auto one = [&](int M, int O, int I) {
    const int s = wgrad_splits(M, O, I, nprob);
    sp.push_back(s);
    tot += al64((size_t)s * O * I) + al64((size_t)s * 2 * O);
  };
  for (auto& l : P.heads) one(R, l.out, l.in);
  for (auto& l : P.dec) one(R, l.out, l.in);
  for (size_t i = 0; i < P.stack.size(); ++i) {
    one(i == 0 ? B : R, P.stack[i].out, P.h);              // hidden part
    if (P.stack_noise[i]) one(R, P.stack[i].out, P.nd);    // noise part
  }
  for (auto& l : P.inp) one(B, l.out, l.in);
  if (splits_out) *splits_out = sp;
  return tot;
}

size_t workspace_floats(const ModelLayout& P, int B, int nz, int mode) {
  const size_t R = (size_t)B * nz, h = P.h;
  size_t t = al64((size_t)B * P.D) + P.inp.size() * al64((size_t)B * h) + al64((size_t)B * h);
  t += (P.stack.size() - 1) * al64(R * h) + al64(R * P.zd);
  if (mode == 0) return t;
  t += P.dec.size() * al64(R * h) + P.heads.size() * al64(R * P.D) + 2 * al64(R);
  t += P.heads.size() * al64(R * P.D) + P.dec.size() * al64(R * h) + (P.stack.size() - 1) * al64(R * h) + P.inp.size() * al64((size_t)B * h);
  t += 2 * al64(R * P.zd) + al64((size_t)B * h);
  t += wgrad_scratch(P, B, (int)R, nullptr);
  return t;
}

void carve(const ModelLayout& P, Bump& ws, int B, int nz, int mode, ModelWs& W) {
  const size_t R = (size_t)B * nz, h = P.h;
  W.x2 = ws.take((size_t)B * P.D);
  W.e.assign(P.inp.size() + 1, nullptr);
  for (size_t l = 1; l <= P.inp.size(); ++l) W.e[l] = ws.take((size_t)B * h);
  W.rb = ws.take((size_t)B * h);
  W.t.assign(P.stack.size(), nullptr);
  for (size_t i = 1; i < P.stack.size(); ++i) W.t[i] = ws.take(R * h);
  W.z = ws.take(R * P.zd);
  if (mode == 0) return;
  W.dcd.assign(P.dec.size() + 1, nullptr);
  for (size_t l = 1; l <= P.dec.size(); ++l) W.dcd[l] = ws.take(R * h);
  for (size_t k = 0; k < P.heads.size(); ++k) W.o.push_back(ws.take(R * P.D));
  W.rec_row = ws.take(R); W.pri_row = ws.take(R);
  for (size_t k = 0; k < P.heads.size(); ++k) W.dox.push_back(ws.take(R * P.D));
  W.ddec.assign(P.dec.size() + 1, nullptr);
  for (size_t l = 1; l <= P.dec.size(); ++l) W.ddec[l] = ws.take(R * h);
  W.dt.assign(P.stack.size(), nullptr);
  for (size_t i = 1; i < P.stack.size(); ++i) W.dt[i] = ws.take(R * h);
  W.de.assign(P.inp.size() + 1, nullptr);
  for (size_t l = 1; l <= P.inp.size(); ++l) W.de[l] = ws.take((size_t)B * h);
  W.dzq = ws.take(R * P.zd); W.dz = ws.take(R * P.zd); W.drb = ws.take((size_t)B * h);
}

int lin1(int epi, int act, int M, int Nout, const float* x, int ldx, int K, const float* wp, LinArgs a, hipStream_t st) {
  a.M = M; a.Nout = Nout; a.nsrc = 1; a.act = act;
  a.src[0].x = x; a.src[0].ld = ldx; a.src[0].K = K; a.src[0].wp = wp;
  return launch_linear(a, epi, st);
}

// sampler trunk (once per image): inp_encode on B rows, then rb = inp . S_1[:, :h]^T + b_S1.  Fills W.e, W.rb.
int encode_trunk(const ModelLayout& P, const ModelPacked& K, const float* params, const float* packed, const float* x, int B,
                 ModelWs& W, hipStream_t st) {
  const int h = P.h, act = P.act;
  const float* x_in = x;
  if (P.kind == 0) {
    ARDAE_TRY(launch_affine(x, (int64_t)B * P.D, 2.f, -1.f, W.x2, st));
    x_in = W.x2;
  }
  for (size_t l = 1; l <= P.inp.size(); ++l) {
    LinArgs A{}; A.bias = params + P.inp[l - 1].b; A.Y = W.e[l]; A.ldY = h;
    ARDAE_TRY(lin1(EPI_ACT, act, B, h, l == 1 ? x_in : W.e[l - 1], l == 1 ? P.D : h, P.inp[l - 1].in, packed + K.inp_f[l - 1], A, st));
  }
  LinArgs A{}; A.bias = params + P.stack[0].b; A.Y = W.rb; A.ldY = h;
  return lin1(EPI_ACT, ACT_NONE, B, P.stack[0].out, W.e[P.inp.size()], h, h, packed + K.sh_f[0], A, st);
}

// sampler stack on R = B*nz rows: layer 0 takes the per-image rb plus its noise part, the last layer writes zdst [R, zd]
// keep_hidden = false (encode / forward_hidden calls, which run under no_grad in the reference): the hidden rows t[] need not exist
int encode_stack(const ModelLayout& P, const ModelPacked& K, const float* params, const float* packed, const float* noise,
                 int B, int nz, const float* rb, const std::vector<float*>& t, float* zdst, bool keep_hidden, hipStream_t st) {
  const int R = B * nz, h = P.h, act = P.act;
  const size_t ns = P.stack.size();
  if (!keep_hidden && ns == 2 && P.stack_noise[0] && !P.stack_noise[1] && P.stack[0].out == h) {
    // noise -> h -> z with nothing else reading the hidden rows: one launch (linear_shortk.hip::sampler_tail_kernel)
    LinArgs A{}; A.M = R; A.Nout = h; A.act = act; A.nsrc = 1;
    A.rowbias = rb; A.rowbias_ld = h; A.rows_per_group = nz;
    A.src[0].x = noise; A.src[0].ld = P.nd; A.src[0].K = P.nd; A.src[0].wp = packed + K.sn_f[0];
    if (sampler_tail_eligible(A, P.stack[1].out))
      return launch_sampler_tail(A, packed + K.sh_f[1], params + P.stack[1].b, zdst, P.stack[1].out, P.stack[1].out, st);
  }
  for (size_t i = 0; i < ns; ++i) {
    const bool last = i + 1 == ns;
    const int out = P.stack[i].out;
    LinArgs A{}; A.Y = last ? zdst : t[i + 1]; A.ldY = out; A.M = R; A.Nout = out; A.act = last ? ACT_NONE : act;
    int n = 0;
    if (i == 0) {
      A.rowbias = rb; A.rowbias_ld = h; A.rows_per_group = nz;
    } else {
      A.bias = params + P.stack[i].b;
      A.src[n].x = t[i]; A.src[n].ld = h; A.src[n].K = h; A.src[n].wp = packed + K.sh_f[i]; ++n;
    }
    if (P.stack_noise[i]) {
      A.src[n].x = noise; A.src[n].ld = P.nd; A.src[n].K = P.nd; A.src[n].wp = packed + K.sn_f[i]; ++n;
    }
    A.nsrc = n;
    ARDAE_TRY(launch_linear(A, EPI_ACT, st));
  }
  return 0;
}

// sampler forward: fills W.e, W.rb, W.t, W.z (and copies z to z_out when given)
int encode_fwd(const ModelLayout& P, const ModelPacked& K, const float* params, const float* packed, const float* x,
               const float* noise, int B, int nz, ModelWs& W, float* z_out, bool keep_hidden, hipStream_t st) {
  ARDAE_TRY(encode_trunk(P, K, params, packed, x, B, W, st));
  if (z_out && !keep_hidden)     // forward-only: the last layer writes the caller's buffer itself (one launch less per encode)
    return encode_stack(P, K, params, packed, noise, B, nz, W.rb, W.t, z_out, false, st);
  ARDAE_TRY(encode_stack(P, K, params, packed, noise, B, nz, W.rb, W.t, W.z, keep_hidden, st));
  if (z_out) {
    ARDAE_TRY(launch_copy(W.z, (size_t)B * nz * P.zd, z_out, st));
  }
  return 0;
}

// extra floats of the encode_pair workspace: the zero-noise stack's hidden rows and its zero noise block (B rows each)
size_t encode_pair_extra(const ModelLayout& P, int B) {
  return (P.stack.size() - 1) * al64((size_t)B * P.h) + al64((size_t)B * P.nd);
}

}  // namespace
}  // namespace ardae

using namespace ardae;

extern "C" {

size_t ardae_model_param_floats(const ardae_model_desc* d) {
  if (desc_ok(d)) return 0;
  if ((d->kind == 5 || d->kind == 6)) return res_model_param_floats(*d);
  if (d->kind == 4) return auxconv_model_param_floats(*d);
  if ((d->kind == 3 || d->kind == 7)) return aux_model_param_floats(*d);
  return d->kind == 2 ? conv_model_param_floats(*d) : ModelLayout(*d).total;
}
size_t ardae_model_packed_floats(const ardae_model_desc* d) {
  if (desc_ok(d)) return 0;
  if ((d->kind == 5 || d->kind == 6)) return res_model_packed_floats(*d);
  if (d->kind == 4) return auxconv_model_packed_floats(*d);
  if ((d->kind == 3 || d->kind == 7)) return aux_model_packed_floats(*d);
  return d->kind == 2 ? conv_model_packed_floats(*d) : ModelPacked(ModelLayout(*d)).total;
}
size_t ardae_model_workspace_floats(const ardae_model_desc* d, int B, int nz, int mode) {
  if (desc_ok(d) || B <= 0 || nz <= 0) return 0;
  if ((d->kind == 5 || d->kind == 6)) return res_model_workspace_floats(*d, B, nz, mode);
  if (d->kind == 4) return auxconv_model_workspace_floats(*d, B, nz, mode);
  if ((d->kind == 3 || d->kind == 7)) return aux_model_workspace_floats(*d, B, nz, mode);
  if (d->kind == 2) return conv_model_workspace_floats(*d, B, nz, mode == 3 ? 0 : mode);
  const ModelLayout P(*d);
  if (mode == 2) return P.dec.size() * al64((size_t)B * nz * P.h);
  if (mode == 3) return workspace_floats(P, B, nz, 0) + encode_pair_extra(P, B);   // ardae_model_encode_pair
  return workspace_floats(P, B, nz, mode) + (size_t)al64((size_t)B * nz * P.nd);   // + a zero-noise buffer for encode(std=0)
}

int ardae_model_pack(const ardae_model_desc* d, const float* params, float* packed, void* stream) {
  ARDAE_TRY(desc_ok(d));
  ARDAE_CHECK_ARG(params && packed, "model_pack: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if ((d->kind == 5 || d->kind == 6)) return res_model_pack(*d, params, packed, st);
  if (d->kind == 4) return auxconv_model_pack(*d, params, packed, st);
  if ((d->kind == 3 || d->kind == 7)) return aux_model_pack(*d, params, packed, st);
  if (d->kind == 2) return conv_model_pack(*d, params, packed, st);
  std::vector<PackItem> pack_items__;
  const ModelLayout P(*d);
  const ModelPacked K(P);
  for (size_t l = 0; l < P.inp.size(); ++l) {
    PACK_PUSH(params + P.inp[l].w, P.inp[l].in, P.inp[l].out, P.inp[l].in, false, packed + K.inp_f[l]);
    PACK_PUSH(params + P.inp[l].w, P.inp[l].in, P.inp[l].in, P.inp[l].out, true, packed + K.inp_b[l]);
  }
  for (size_t i = 0; i < P.stack.size(); ++i) {
    const Lin& l = P.stack[i];
    PACK_PUSH(params + l.w, l.in, l.out, P.h, false, packed + K.sh_f[i]);
    PACK_PUSH(params + l.w, l.in, P.h, l.out, true, packed + K.sh_b[i]);
    if (P.stack_noise[i]) PACK_PUSH(params + l.w + P.h, l.in, l.out, P.nd, false, packed + K.sn_f[i]);
  }
  for (size_t l = 0; l < P.dec.size(); ++l) {
    PACK_PUSH(params + P.dec[l].w, P.dec[l].in, P.dec[l].out, P.dec[l].in, false, packed + K.dec_f[l]);
    PACK_PUSH(params + P.dec[l].w, P.dec[l].in, P.dec[l].in, P.dec[l].out, true, packed + K.dec_b[l]);
  }
  for (size_t k = 0; k < P.heads.size(); ++k) {
    PACK_PUSH(params + P.heads[k].w, P.heads[k].in, P.heads[k].out, P.heads[k].in, false, packed + K.head_f[k]);
    PACK_PUSH(params + P.heads[k].w, P.heads[k].in, P.heads[k].in, P.heads[k].out, true, packed + K.head_b[k]);
  }
  PACK_FLUSH(st);
  return 0;
}

static int model_common(const ardae_model_desc* d, const float* params, const float* packed, const float* x, int B, int nz,
                        float* workspace, size_t wsf, int mode) {
  ARDAE_TRY(desc_ok(d));
  ARDAE_CHECK_ARG(params && packed && x && workspace, "model: null pointer argument");
  ARDAE_CHECK_ARG(B > 0 && nz > 0 && (int64_t)B * nz < (int64_t)1 << 30, "model: bad batch (B=%d nz=%d)", B, nz);
  const size_t need = ardae_model_workspace_floats(d, B, nz, mode);
  ARDAE_CHECK_ARG(wsf >= need, "model: workspace too small (%zu < %zu floats)", wsf, need);
  return 0;
}

int ardae_model_encode(const ardae_model_desc* d, const float* params, const float* packed, const float* x, const float* noise,
                       int B, int nz, float* workspace, size_t workspace_floats_, float* z_out, void* stream) {
  ARDAE_TRY(model_common(d, params, packed, x, B, nz, workspace, workspace_floats_, 0));
  ARDAE_CHECK_ARG(z_out, "model_encode: z_out is NULL");
  hipStream_t st = (hipStream_t)stream;
  if ((d->kind == 5 || d->kind == 6)) return res_model_encode(*d, params, packed, x, noise, B, nz, workspace, workspace_floats_, z_out, nullptr, st);
  if ((d->kind == 3 || d->kind == 7)) {
    ARDAE_TRY(aux_model_encode(*d, params, packed, x, noise, B, nz, workspace, workspace_floats_, z_out, nullptr, st));
    return 0;
  }
  if (d->kind == 4) return auxconv_model_encode(*d, params, packed, x, noise, B, nz, workspace, workspace_floats_, z_out, nullptr, st);
  if (d->kind == 2) return conv_model_encode(*d, params, packed, x, noise, B, nz, workspace, workspace_floats_, z_out, st);
  const ModelLayout P(*d);
  const ModelPacked K(P);
  Bump ws(workspace, workspace_floats_);
  ModelWs W;
  carve(P, ws, B, nz, 0, W);
  const float* nz_ptr = noise;
  if (!noise) {   // encode(x, std=0): the reference multiplies its draw by 0
    float* zero = ws.take((size_t)B * nz * P.nd);
    ARDAE_TRY(launch_fill(zero, (size_t)B * nz * P.nd, 0.f, st));
    nz_ptr = zero;
  }
  ARDAE_CHECK_ARG(ws.ok, "model_encode: internal workspace accounting error");
  ARDAE_TRY(encode_fwd(P, K, params, packed, x, nz_ptr, B, nz, W, z_out, false, st));
  return 0;
}

int ardae_model_encode_pair(const ardae_model_desc* d, const float* params, const float* packed, const float* x, const float* noise,
                            int B, int nz, float* workspace, size_t workspace_floats_, float* z0_out, float* z_out, int phase,
                            void* stream) {
  ARDAE_TRY(desc_ok(d));
  ARDAE_CHECK_ARG(phase >= 0 && phase <= 2, "model_encode_pair: phase must be 0 (all), 1 (trunk + z0) or 2 (N-row stack)");
  ARDAE_CHECK_ARG(params && packed && x && (noise || phase == 1) && workspace && z0_out && z_out, "model_encode_pair: null pointer argument");
  ARDAE_CHECK_ARG(B > 0 && nz > 0 && (int64_t)B * nz < (int64_t)1 << 30, "model_encode_pair: bad batch (B=%d nz=%d)", B, nz);
  ARDAE_CHECK_ARG(workspace_floats_ >= ardae_model_workspace_floats(d, B, nz, 3), "model_encode_pair: workspace too small");
  if (d->kind >= 2) {   // conv / aux samplers: two passes over the same workspace
    if (phase != 2) ARDAE_TRY(ardae_model_encode(d, params, packed, x, nullptr, B, 1, workspace, workspace_floats_, z0_out, stream));
    if (phase == 1) return 0;
    return ardae_model_encode(d, params, packed, x, noise, B, nz, workspace, workspace_floats_, z_out, stream);
  }
  hipStream_t st = (hipStream_t)stream;
  const ModelLayout P(*d);
  const ModelPacked K(P);
  Bump ws(workspace, workspace_floats_);
  ModelWs W;
  carve(P, ws, B, nz, 0, W);
  std::vector<float*> t0(P.stack.size(), nullptr);
  for (size_t i = 1; i < P.stack.size(); ++i) t0[i] = ws.take((size_t)B * P.h);
  float* zero = ws.take((size_t)B * P.nd);
  ARDAE_CHECK_ARG(ws.ok, "model_encode_pair: internal workspace accounting error");
  if (phase != 2) {
    ARDAE_TRY(launch_fill(zero, (size_t)B * P.nd, 0.f, st));
    ARDAE_TRY(encode_trunk(P, K, params, packed, x, B, W, st));
    ARDAE_TRY(encode_stack(P, K, params, packed, zero, B, 1, W.rb, t0, z0_out, false, st));    // encode(x, std=0): the draw is multiplied by 0
  }
  if (phase != 1) ARDAE_TRY(encode_stack(P, K, params, packed, noise, B, nz, W.rb, W.t, z_out, false, st));   // forward_hidden(x, nz); W.rb from phase 1
  return 0;
}

int ardae_model_encode_hidden_raw(const ardae_model_desc* d, const float* params, const float* packed, const float* x, const float* raw0, int B,
                                  float* workspace, size_t workspace_floats_, float* z0_out, float* hidden_out, void* stream) {
  ARDAE_TRY(model_common(d, params, packed, x, B, 1, workspace, workspace_floats_, 0));
  ARDAE_CHECK_ARG(d->kind == 6 && (d->flags & ARDAE_MODEL_CLIPPED), "model_encode_hidden_raw: the clipped aux-resconv class only (kind 6 + ARDAE_MODEL_CLIPPED)");
  ARDAE_CHECK_ARG(z0_out || hidden_out, "model_encode_hidden_raw: nothing to return");
  return res_model_encode(*d, params, packed, x, nullptr, B, 1, workspace, workspace_floats_, z0_out, hidden_out, (hipStream_t)stream, raw0);
}
int ardae_model_encode_hidden(const ardae_model_desc* d, const float* params, const float* packed, const float* x, int B, float* workspace,
                              size_t workspace_floats_, float* z0_out, float* hidden_out, void* stream) {
  ARDAE_TRY(model_common(d, params, packed, x, B, 1, workspace, workspace_floats_, 0));
  ARDAE_CHECK_ARG((d->kind == 3 || d->kind == 7) || d->kind == 4 || d->kind == 6, "model_encode_hidden: the hidden1a context exists for the aux models only (kinds 3, 4, 6, 7)");
  ARDAE_CHECK_ARG(hidden_out, "model_encode_hidden: hidden_out is NULL");
  hipStream_t st = (hipStream_t)stream;
  if (d->kind == 6) return res_model_encode(*d, params, packed, x, nullptr, B, 1, workspace, workspace_floats_, z0_out, hidden_out, st);   // z0_out may be NULL
  if (d->kind == 4) ARDAE_TRY(auxconv_model_encode(*d, params, packed, x, nullptr, B, 1, workspace, workspace_floats_, z0_out, hidden_out, st));
  else ARDAE_TRY(aux_model_encode(*d, params, packed, x, nullptr, B, 1, workspace, workspace_floats_, z0_out, hidden_out, st));
  return 0;
}

int ardae_model_decode(const ardae_model_desc* d, const float* params, const float* packed, const float* z, int R, float* workspace,
                       size_t workspace_floats_, float* out0, float* out1, void* stream) {
  ARDAE_TRY(desc_ok(d));
  ARDAE_CHECK_ARG(params && packed && z && workspace && out0 && R > 0, "model_decode: bad arguments");
  if ((d->kind == 5 || d->kind == 6)) {
    ARDAE_CHECK_ARG(workspace_floats_ >= res_model_workspace_floats(*d, R, 1, 2), "model_decode: workspace too small");
    return res_model_decode(*d, params, packed, z, R, workspace, workspace_floats_, out0, (hipStream_t)stream);
  }
  if (d->kind == 2) {
    ARDAE_CHECK_ARG(workspace_floats_ >= conv_model_workspace_floats(*d, R, 1, 2), "model_decode: workspace too small");
    return conv_model_decode(*d, params, packed, z, R, workspace, workspace_floats_, out0, (hipStream_t)stream);
  }
  if (d->kind == 4) {
    ARDAE_CHECK_ARG(workspace_floats_ >= auxconv_model_workspace_floats(*d, R, 1, 2), "model_decode: workspace too small");
    return auxconv_model_decode(*d, params, packed, z, R, workspace, workspace_floats_, out0, (hipStream_t)stream);
  }
  if ((d->kind == 3 || d->kind == 7)) {
    ARDAE_CHECK_ARG(workspace_floats_ >= aux_model_workspace_floats(*d, R, 1, 2), "model_decode: workspace too small");
    ARDAE_TRY(aux_model_decode(*d, params, packed, z, R, workspace, workspace_floats_, out0, (hipStream_t)stream, out1));
    return 0;
  }
  const ModelLayout P(*d);
  const ModelPacked K(P);
  ARDAE_CHECK_ARG(P.kind == 0 || out1, "model_decode: the Gaussian decoder needs out1 (logvar)");
  ARDAE_CHECK_ARG(workspace_floats_ >= P.dec.size() * al64((size_t)R * P.h), "model_decode: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  Bump ws(workspace, workspace_floats_);
  const int h = P.h;
  const float* cur = z;
  for (size_t l = 1; l <= P.dec.size(); ++l) {
    float* nxt = ws.take((size_t)R * h);
    LinArgs A{}; A.bias = params + P.dec[l - 1].b; A.Y = nxt; A.ldY = h;
    ARDAE_TRY(lin1(EPI_ACT, P.act, R, h, cur, l == 1 ? P.zd : h, P.dec[l - 1].in, packed + K.dec_f[l - 1], A, st));
    cur = nxt;
  }
  for (size_t k = 0; k < P.heads.size(); ++k) {
    LinArgs A{}; A.bias = params + P.heads[k].b; A.Y = k == 0 ? out0 : out1; A.ldY = P.D;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.D, cur, h, h, packed + K.head_f[k], A, st));
  }
  return 0;
}

int ardae_model_loss_rows(const ardae_model_desc* d, const float* out0, const float* out1, const float* x, const float* z, int rows,
                          int nz, float* recon_row, float* prior_row, void* stream) {
  ARDAE_TRY(desc_ok(d));
  return launch_vae_loss(d->kind == 1 || d->kind == 7 ? 1 : 0, out0, out1, x, z, rows, nz, d->input_dim, d->z_dim, 1.f, 0, 0.f, nullptr, recon_row, prior_row, nullptr,
                         nullptr, nullptr, (hipStream_t)stream);
}

int ardae_model_vae_forward(const ardae_model_desc* d, const float* params, const float* packed, const float* x, const float* noise,
                            int B, int nz, float beta, float* workspace, size_t workspace_floats_, float* z_out, float* losses,
                            void* stream) {
  ARDAE_TRY(model_common(d, params, packed, x, B, nz, workspace, workspace_floats_, 1));
  ARDAE_CHECK_ARG(noise && z_out && losses, "model_vae_forward: null pointer argument");
  hipStream_t st = (hipStream_t)stream;
  if ((d->kind == 5 || d->kind == 6)) return res_model_vae_forward(*d, params, packed, x, noise, B, nz, beta, workspace, workspace_floats_, z_out, losses, st);
  if ((d->kind == 3 || d->kind == 7)) {
    ARDAE_TRY(aux_model_vae_forward(*d, params, packed, x, noise, B, nz, beta, workspace, workspace_floats_, z_out, losses, st));
    return 0;
  }
  if (d->kind == 4) return auxconv_model_vae_forward(*d, params, packed, x, noise, B, nz, beta, workspace, workspace_floats_, z_out, losses, st);
  if (d->kind == 2) return conv_model_vae_forward(*d, params, packed, x, noise, B, nz, beta, workspace, workspace_floats_, z_out, losses, st);
  const ModelLayout P(*d);
  const ModelPacked K(P);
  Bump ws(workspace, workspace_floats_);
  ModelWs W;
  carve(P, ws, B, nz, 1, W);
  ARDAE_CHECK_ARG(ws.ok, "model_vae_forward: internal workspace accounting error");
  const int R = B * nz, h = P.h, act = P.act;
  ARDAE_TRY(encode_fwd(P, K, params, packed, x, noise, B, nz, W, z_out, true, st));
  for (size_t l = 1; l <= P.dec.size(); ++l) {
    LinArgs A{}; A.bias = params + P.dec[l - 1].b; A.Y = W.dcd[l]; A.ldY = h;
    ARDAE_TRY(lin1(EPI_ACT, act, R, h, l == 1 ? W.z : W.dcd[l - 1], l == 1 ? P.zd : h, P.dec[l - 1].in, packed + K.dec_f[l - 1], A, st));
  }
  for (size_t k = 0; k < P.heads.size(); ++k) {
    LinArgs A{}; A.bias = params + P.heads[k].b; A.Y = W.o[k]; A.ldY = P.D;
    ARDAE_TRY(lin1(EPI_ACT, ACT_NONE, R, P.D, W.dcd[P.dec.size()], h, h, packed + K.head_f[k], A, st));
  }
  ARDAE_TRY(launch_vae_loss(P.kind, W.o[0], P.kind == 1 ? W.o[1] : nullptr, x, W.z, R, nz, P.D, P.zd, beta, 0, 0.f, nullptr, W.rec_row,
                            W.pri_row, nullptr, nullptr, nullptr, st));
  return launch_vae_loss_finalize(W.rec_row, W.pri_row, R, beta, losses, st);
}

// phases: 1 = loss gradients + decoder backward up to dz (needs nothing from the cDAE), 2 = entropy seed + sampler backward +
// weight gradients, 3 = both.  With phases == 3 the (pre-scaled) seed enters through the loss kernel; with phases == 2 it is
// added to dz as seed_scale * dz_extra.
static int vae_backward_impl(const ardae_model_desc* d, const float* params, const float* packed, const float* x, const float* noise,
                             int B, int nz, float beta, float dloss, const float* dz_extra, float seed_scale, float* workspace,
                             size_t workspace_floats_, float* grads, float grads_beta, int phases, hipStream_t st) {
  const ModelLayout P(*d);
  const ModelPacked K(P);
  Bump ws(workspace, workspace_floats_);
  ModelWs W;
  carve(P, ws, B, nz, 1, W);
  const int R = B * nz, h = P.h, act = P.act;
  const size_t ns = P.stack.size(), ndec = P.dec.size(), ninp = P.inp.size(), nh = P.heads.size();
  const float gscale = dloss / (float)R;
  if (phases & 1) {
  ARDAE_TRY(launch_vae_loss(P.kind, W.o[0], P.kind == 1 ? W.o[1] : nullptr, x, W.z, R, nz, P.D, P.zd, beta, 1, gscale,
                            phases == 3 ? dz_extra : nullptr, W.rec_row, W.pri_row, W.dox[0], P.kind == 1 ? W.dox[1] : nullptr, W.dzq, st));
  // decoder backward
  {
    LinArgs A{}; A.S = W.dcd[ndec]; A.ldS = h; A.Y = W.ddec[ndec]; A.ldY = h; A.M = R; A.Nout = h; A.act = act; A.nsrc = (int)nh;
    for (size_t k = 0; k < nh; ++k) { A.src[k].x = W.dox[k]; A.src[k].ld = P.D; A.src[k].K = P.D; A.src[k].wp = packed + K.head_b[k]; }
    ARDAE_TRY(launch_linear(A, EPI_DACT, st));
  }
  for (size_t l = ndec; l >= 2; --l) {
    LinArgs A{}; A.S = W.dcd[l - 1]; A.ldS = h; A.Y = W.ddec[l - 1]; A.ldY = h;
    ARDAE_TRY(lin1(EPI_DACT, act, R, h, W.ddec[l], h, h, packed + K.dec_b[l - 1], A, st));
  }
  {  // dz = ddec_1 . D_1 + (prior + injected seed)      (act NONE: act' == 1, S is only a placeholder)
    LinArgs A{}; A.S = W.dzq; A.ldS = P.zd; A.Q = W.dzq; A.ldQ = P.zd; A.Y = W.dz; A.ldY = P.zd;
    ARDAE_TRY(lin1(EPI_DACT, ACT_NONE, R, P.zd, W.ddec[1], h, h, packed + K.dec_b[0], A, st));
  }
  }
  if (!(phases & 2)) return 0;
  if (phases == 2 && dz_extra) ARDAE_TRY(launch_axpy(dz_extra, (int64_t)R * P.zd, seed_scale, W.dz, st));   // + the entropy seed
  // sampler backward
  for (size_t i = ns - 1; i >= 1; --i) {
    LinArgs A{}; A.S = W.t[i]; A.ldS = h; A.Y = W.dt[i]; A.ldY = h;
    const float* src = (i == ns - 1) ? W.dz : W.dt[i + 1];
    const int kk = P.stack[i].out;
    ARDAE_TRY(lin1(EPI_DACT, act, R, h, src, kk, kk, packed + K.sh_b[i], A, st));
  }
  ARDAE_TRY(launch_segment_sum(W.dt[1], h, B, nz, h, 1.0f, W.drb, h, st));
  {
    LinArgs A{}; A.S = W.e[ninp]; A.ldS = h; A.Y = W.de[ninp]; A.ldY = h;
    ARDAE_TRY(lin1(EPI_DACT, act, B, h, W.drb, h, h, packed + K.sh_b[0], A, st));
  }
  for (size_t l = ninp; l >= 2; --l) {
    LinArgs A{}; A.S = W.e[l - 1]; A.ldS = h; A.Y = W.de[l - 1]; A.ldY = h;
    ARDAE_TRY(lin1(EPI_DACT, act, B, h, W.de[l], h, h, packed + K.inp_b[l - 1], A, st));
  }
  // weight gradients: one batched launch (problem order must match wgrad_scratch)
  std::vector<int> splits;
  wgrad_scratch(P, B, R, &splits);
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
  const float* x_in = P.kind == 0 ? W.x2 : x;
  for (size_t k = 0; k < nh; ++k) push(R, P.D, h, W.dox[k], W.dcd[ndec], h, grads + P.heads[k].w, h, grads + P.heads[k].b);
  for (size_t l = 1; l <= ndec; ++l)
    push(R, h, P.dec[l - 1].in, W.ddec[l], l == 1 ? W.z : W.dcd[l - 1], l == 1 ? P.zd : h, grads + P.dec[l - 1].w, P.dec[l - 1].in,
         grads + P.dec[l - 1].b);
  for (size_t i = 0; i < ns; ++i) {
    const Lin& L = P.stack[i];
    const float* G = (i == ns - 1) ? W.dz : W.dt[i + 1];
    if (i == 0) push(B, L.out, h, W.drb, W.e[ninp], h, grads + L.w, L.in, nullptr);        // per-image hidden part
    else push(R, L.out, h, G, W.t[i], h, grads + L.w, L.in, P.stack_noise[i] ? nullptr : grads + L.b);
    if (P.stack_noise[i]) push(R, L.out, P.nd, G, noise, P.nd, grads + L.w + h, L.in, grads + L.b);   // noise part (+ bias)
  }
  for (size_t l = 1; l <= ninp; ++l)
    push(B, h, P.inp[l - 1].in, W.de[l], l == 1 ? x_in : W.e[l - 1], l == 1 ? P.D : h, grads + P.inp[l - 1].w, P.inp[l - 1].in,
         grads + P.inp[l - 1].b);
  ARDAE_CHECK_ARG(ws.ok, "model_vae_backward: internal workspace accounting error");
  return launch_wgrad_batch(probs.data(), (int)probs.size(), st);
}

int ardae_model_vae_backward(const ardae_model_desc* d, const float* params, const float* packed, const float* x, const float* noise,
                             int B, int nz, float beta, float dloss, const float* dz_extra, float* workspace,
                             size_t workspace_floats_, float* grads, float grads_beta, void* stream) {
  ARDAE_TRY(model_common(d, params, packed, x, B, nz, workspace, workspace_floats_, 1));
  ARDAE_CHECK_ARG(noise && grads, "model_vae_backward: null pointer argument");
  hipStream_t st = (hipStream_t)stream;
  if ((d->kind == 5 || d->kind == 6)) {
    return res_model_vae_backward(*d, params, packed, x, noise, B, nz, beta, dloss, dz_extra, workspace, workspace_floats_, grads, grads_beta, st);
  }
  if (d->kind == 2) {
    return conv_model_vae_backward(*d, params, packed, x, noise, B, nz, beta, dloss, dz_extra, workspace, workspace_floats_, grads, grads_beta, st);
  }
  if (d->kind == 4) {
    return auxconv_model_vae_backward(*d, params, packed, x, noise, B, nz, beta, dloss, dz_extra, workspace, workspace_floats_, grads, grads_beta, st);
  }
  if ((d->kind == 3 || d->kind == 7)) {
    return aux_model_vae_backward(*d, params, packed, x, noise, B, nz, beta, dloss, dz_extra, workspace, workspace_floats_, grads, grads_beta, st);
  }
  return vae_backward_impl(d, params, packed, x, noise, B, nz, beta, dloss, dz_extra, 1.f, workspace, workspace_floats_, grads, grads_beta, 3, st);
}

int ardae_model_vae_backward_decoder(const ardae_model_desc* d, const float* params, const float* packed, const float* x,
                                     const float* noise, int B, int nz, float beta, float dloss, float* workspace,
                                     size_t workspace_floats_, void* stream) {
  ARDAE_TRY(model_common(d, params, packed, x, B, nz, workspace, workspace_floats_, 1));
  ARDAE_CHECK_ARG(noise, "model_vae_backward_decoder: null pointer argument");
  ARDAE_CHECK_ARG(d->kind < 2, "model_vae_backward_decoder: the conv model and the aux model have no split backward (use ardae_model_vae_backward)");
  return vae_backward_impl(d, params, packed, x, noise, B, nz, beta, dloss, nullptr, 0.f, workspace, workspace_floats_, nullptr, 0.f, 1,
                           (hipStream_t)stream);
}

int ardae_model_vae_backward_sampler(const ardae_model_desc* d, const float* params, const float* packed, const float* x,
                                     const float* noise, int B, int nz, const float* dz_extra, float seed_scale, float* workspace,
                                     size_t workspace_floats_, float* grads, float grads_beta, void* stream) {
  ARDAE_TRY(model_common(d, params, packed, x, B, nz, workspace, workspace_floats_, 1));
  ARDAE_CHECK_ARG(noise && grads, "model_vae_backward_sampler: null pointer argument");
  ARDAE_CHECK_ARG(d->kind < 2, "model_vae_backward_sampler: the conv model and the aux model have no split backward (use ardae_model_vae_backward)");
  return vae_backward_impl(d, params, packed, x, noise, B, nz, 0.f, 0.f, dz_extra, seed_scale, workspace, workspace_floats_, grads,
                           grads_beta, 2, (hipStream_t)stream);
}

}  // extern "C"
