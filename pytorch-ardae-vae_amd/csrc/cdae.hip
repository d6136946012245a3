// Conditional AR-DAE update on gfx950: forward, score pass, DAE loss, double backward and weight gradients,
// orchestrated on the host from the K1 (linear.hip) / K6w (wgrad.hip) kernels.
//
// Maths: SURVEY.md Appendix A (closed form of models/graddae/mlp.py:400-444 under autograd), notation below.
//   rows i=1..N (N = B*S), image b(i) = i / S
//   inp_encode : a_0 = xbar, a_l = sp(A_l a_{l-1} + b_l)                      l = 1..L        (N rows)
//   ctx_encode : c_0 = ctx,  c_l = sp(C_l c_{l-1} + bc_l)                     l = 1..L        (B rows, ONCE per image)
//   energy MLP : h_1 = sp(W1a a_L + [W1c c_L + d_1](b) + sigma w1s), h_l = sp(W_l h_{l-1} + d_l), E = w.h_L + d_f
//   score      : e_L = -w (.) s(h_L); e_{l-1} = (e_l W_l) (.) s(h_{l-1}); r_L = (e_1 W1a) (.) s(a_L);
//                r_{l-1} = (r_l A_l) (.) s(a_{l-1});  g = r_1 A_1          [s(.) = softplus' rebuilt from the saved output]
//   loss       : rho = sigma g + eps; loss = sum rho^2 / (N z); gbar = 2 sigma rho / (N z)
//   forward-mode chain (the create_graph half of the double backward):
//                rb_1 = gbar A_1^T; tau_l = rb_l (.) s(a_l); pbar_l = rb_l (.) r_l (.) (1 - s(a_l)); rb_{l+1} = tau_l A_{l+1}^T
//                eb_1 = tau_L W1a^T; tau'_l = eb_l (.) s(h_l); qbar_l = eb_l (.) e_l (.) (1 - s(h_l)); eb_{l+1} = tau'_l W_{l+1}^T
//   ordinary backward seeded by qbar/pbar:
//                qhat_L = qbar_L; qhat_{l-1} = qbar_{l-1} + (qhat_l W_l) (.) s(h_{l-1});
//                phat_L = pbar_L + (qhat_1 W1a) (.) s(a_L); phat_{l-1} = pbar_{l-1} + (phat_l A_l) (.) s(a_{l-1})
//                ctx branch: Qsum_b = sum_{i in b} qhat_1[i]  (reduce over S BEFORE the ctx chain), chat_L = (Qsum W1c) (.) s(c_L) ...
//   weight gradients (one batched launch): see the problem list in cdae_loss_grads_impl().
#include <vector>

#include "ardae_hip.h"
#include "common.h"
#include "elementwise.h"
#include "linear.h"
#include "wgrad.h"

namespace ardae {
namespace {

// collect pack requests; flushed with one launch by PACK_FLUSH
#define PACK_PUSH(W_, ldw_, nout_, k_, tr_, out_) pack_items__.push_back(PackItem{W_, ldw_, nout_, k_, (tr_) ? 1 : 0, out_})
#define PACK_FLUSH(st_) ARDAE_TRY(launch_pack_batch(pack_items__.data(), (int)pack_items__.size(), st_))

struct Lin {
  size_t w, b;   // offsets (floats) into the flat parameter buffer
  int out, in;
};

struct CdaeLayout {
  int kind, z, c, h, L, act;
  std::vector<Lin> ctx, inp, neg;   // ctx/inp: L linears (L-1 hidden + fc); neg: L hidden + fc (= neglogprob / dae)
  size_t total = 0;
  int out_dim() const { return kind == 0 ? 1 : z; }

  explicit CdaeLayout(const ardae_cdae_desc& d) : kind(d.kind), z(d.input_dim), c(d.context_dim), h(d.h_dim), L(d.n_layers), act(d.act) {
    size_t off = 0;
    auto add = [&](std::vector<Lin>& v, int out, int in) {
      Lin l; l.out = out; l.in = in; l.w = off; off += (size_t)out * in; l.b = off; off += out;
      v.push_back(l);
    };
    for (int l = 0; l < L; ++l) add(ctx, h, l == 0 ? c : h);
    for (int l = 0; l < L; ++l) add(inp, h, l == 0 ? z : h);
    for (int l = 0; l < L; ++l) add(neg, h, l == 0 ? 2 * h + 1 : h);
    add(neg, out_dim(), h);
    total = off;
  }
};

// offsets into the packed-weight buffer
struct PackedLayout {
  std::vector<size_t> ctx_f, ctx_b, inp_f, inp_b, neg_f, neg_b;   // neg_*[0] unused (W1 is split below)
  size_t w1a_f, w1a_b, w1c_f, w1c_b, w1s;
  size_t fc_f, fc_b;   // res kind only (dae.fc [z,h])
  size_t total = 0;
  explicit PackedLayout(const CdaeLayout& P) {
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 63) & ~size_t(63); return o; };
    for (int l = 0; l < P.L; ++l) {
      ctx_f.push_back(take(packed_floats(P.ctx[l].out, P.ctx[l].in)));
      ctx_b.push_back(take(packed_floats(P.ctx[l].in, P.ctx[l].out)));
      inp_f.push_back(take(packed_floats(P.inp[l].out, P.inp[l].in)));
      inp_b.push_back(take(packed_floats(P.inp[l].in, P.inp[l].out)));
    }
    neg_f.assign(P.L, 0); neg_b.assign(P.L, 0);
    w1a_f = take(packed_floats(P.h, P.h)); w1a_b = take(packed_floats(P.h, P.h));
    w1c_f = take(packed_floats(P.h, P.h)); w1c_b = take(packed_floats(P.h, P.h));
    w1s = take(P.h);
    for (int l = 1; l < P.L; ++l) {
      neg_f[l] = take(packed_floats(P.h, P.h));
      neg_b[l] = take(packed_floats(P.h, P.h));
    }
    fc_f = fc_b = 0;
    if (P.kind == 1) {
      fc_f = take(packed_floats(P.z, P.h));
      fc_b = take(packed_floats(P.h, P.z));
    }
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

int desc_ok(const ardae_cdae_desc* d) {
  ARDAE_CHECK_ARG(d != nullptr, "cdae: desc is NULL");
  ARDAE_CHECK_ARG(d->kind == 0 || d->kind == 1, "cdae: kind must be 0 (mlp-grad) or 1 (mlp-res)");
  ARDAE_CHECK_ARG(d->input_dim >= 1 && d->context_dim >= 1 && d->h_dim >= 1 && d->n_layers >= 1, "cdae: bad dimensions");
  ARDAE_CHECK_ARG(d->n_layers <= 6, "cdae: n_layers <= 6 supported (3L+1 gradient problems per batch)");
  // every activation of get_nonlinear_func; with a piecewise linear one mlp-grad's second-order terms vanish, as
  // they do under autograd in the reference
  ARDAE_CHECK_ARG(d->act > ACT_NONE && d->act <= ACT_LAST, "cdae: unknown activation %d", d->act);
  return 0;
}

size_t wgrad_scratch(const CdaeLayout& P, int B, int S, std::vector<int>* splits_out) {
  // problem order must match cdae_loss_grads_impl
  const int N = B * S;
  const int nprob = 3 * P.L + 1;
  size_t tot = 0;
  std::vector<int> sp;
  auto one = [&](int M, int O, int I) {
    const int s = wgrad_splits(M, O, I, nprob);
    sp.push_back(s);
    tot += (((size_t)s * O * I) + 63) & ~size_t(63);
    tot += (((size_t)s * 2 * O) + 63) & ~size_t(63);
  };
  for (int l = 0; l < P.L; ++l) one(N, P.h, P.inp[l].in);         // inp A_l
  one(N, P.h, P.h);                                               // W1a (+ d_1, w1s)
  one(B, P.h, P.h);                                               // W1c
  for (int l = 1; l < P.L; ++l) one(N, P.h, P.h);                 // W_l
  if (P.kind == 1) one(N, P.z, P.h);                              // dae.fc
  for (int l = 0; l < P.L; ++l) one(B, P.h, P.ctx[l].in);         // ctx C_l
  if (splits_out) *splits_out = sp;
  return tot;
}

size_t workspace_floats(const CdaeLayout& P, int B, int S, bool need_grads) {
  const size_t N = (size_t)B * S, h = P.h;
  auto al = [](size_t n) { return (n + 63) & ~size_t(63); };
  size_t t = 0;
  t += (size_t)(P.L) * al((size_t)B * h) + al((size_t)B * h);            // c_l, cb
  t += (size_t)(need_grads ? 8 : 4) * P.L * al(N * h);                   // a,hh,e,r (+ tau,taup,pbar,qbar)
  t += 2 * al(N * P.z);                                                  // gbar, g
  t += al((size_t)linear_row_tiles((int)N, P.z) * linear_col_panels((int)N, P.z));
  t += al((size_t)LINEAR_SMALL_CHAIN_COUNTER_WORDS * ((N + 15) / 16));                                          // row-block counters of the per-image chain launch (score pass)
  if (need_grads) {
    t += al((size_t)B * h) + (size_t)P.L * al((size_t)B * h);            // Qsum, chat_l
    t += al((size_t)linear_row_tiles((int)N, P.h) * h);                  // colsum of tau'_L
    t += wgrad_scratch(P, B, S, nullptr);
  }
  return t;
}

int cdae_pack_impl(const CdaeLayout& P, const PackedLayout& K, const float* params, float* packed, hipStream_t st) {
  std::vector<PackItem> pack_items__;
  for (int l = 0; l < P.L; ++l) {
    PACK_PUSH(params + P.ctx[l].w, P.ctx[l].in, P.ctx[l].out, P.ctx[l].in, false, packed + K.ctx_f[l]);
    PACK_PUSH(params + P.ctx[l].w, P.ctx[l].in, P.ctx[l].in, P.ctx[l].out, true, packed + K.ctx_b[l]);
    PACK_PUSH(params + P.inp[l].w, P.inp[l].in, P.inp[l].out, P.inp[l].in, false, packed + K.inp_f[l]);
    PACK_PUSH(params + P.inp[l].w, P.inp[l].in, P.inp[l].in, P.inp[l].out, true, packed + K.inp_b[l]);
  }
  const float* W1 = params + P.neg[0].w;
  const int ld1 = 2 * P.h + 1;
  PACK_PUSH(W1, ld1, P.h, P.h, false, packed + K.w1a_f);
  PACK_PUSH(W1, ld1, P.h, P.h, true, packed + K.w1a_b);
  PACK_PUSH(W1 + P.h, ld1, P.h, P.h, false, packed + K.w1c_f);
  PACK_PUSH(W1 + P.h, ld1, P.h, P.h, true, packed + K.w1c_b);
  ARDAE_TRY(launch_gather_strided(W1 + 2 * P.h, ld1, P.h, packed + K.w1s, st));
  for (int l = 1; l < P.L; ++l) {
    PACK_PUSH(params + P.neg[l].w, P.h, P.h, P.h, false, packed + K.neg_f[l]);
    PACK_PUSH(params + P.neg[l].w, P.h, P.h, P.h, true, packed + K.neg_b[l]);
  }
  if (P.kind == 1) {
    PACK_PUSH(params + P.neg[P.L].w, P.h, P.z, P.h, false, packed + K.fc_f);
    PACK_PUSH(params + P.neg[P.L].w, P.h, P.h, P.z, true, packed + K.fc_b);
  }
  PACK_FLUSH(st);
  return 0;
}

// one fused Linear launch with a single source
LinArgs lin_args(int act, int M, int Nout, const float* x, int ldx, int K, const float* wp, LinArgs a) {
  a.M = M; a.Nout = Nout; a.nsrc = 1; a.act = act;
  a.src[0].x = x; a.src[0].ld = ldx; a.src[0].K = K; a.src[0].wp = wp;
  return a;
}
int lin(int epi, int act, int M, int Nout, const float* x, int ldx, int K, const float* wp, LinArgs a, hipStream_t st) {
  return launch_linear(lin_args(act, M, Nout, x, ldx, K, wp, a), epi, st);
}

// A run of consecutive N-row layers of one epilogue kind, each reading its predecessor's output: the longest prefixes that
// qualify go out as ONE launch each (linear_chain.hip: the small-shard regime), the rest layer by layer.
struct LayerRun {
  int epi;
  hipStream_t st;
  std::vector<LinArgs> v;
  LayerRun(int epi_, hipStream_t st_) : epi(epi_), st(st_) {}
  void add(int act, int M, int Nout, const float* x, int ldx, int K, const float* wp, LinArgs a) { v.push_back(lin_args(act, M, Nout, x, ldx, K, wp, a)); }
  int flush() {
    size_t i = 0;
    while (i < v.size()) {
      size_t n = v.size() - i;
      while (n >= 2 && !linear_chain_eligible(v.data() + i, (int)n, epi)) --n;
      if (n >= 2) {
        ARDAE_TRY(launch_linear_chain(v.data() + i, (int)n, epi, st));
        i += n;
        continue;
      }
      // many tiles per workgroup: the same run layer-major in the weight-stationary kernel (one launch, the slab loaded once per layer)
      size_t m = std::min<size_t>(v.size() - i, 6);
      while (m >= 2 && !linear_wide_layers_eligible(v.data() + i, (int)m, epi)) --m;
      if (m >= 2) {
        ARDAE_TRY(launch_linear_wide_layers(v.data() + i, (int)m, epi, st));
        i += m;
      } else {
        ARDAE_TRY(launch_linear(v[i], epi, st));
        i += 1;
      }
    }
    v.clear();
    return 0;
  }
};

// where cdae_impl's workspace layout puts a_1 [N, h] (behind c_1..c_L and the per-image bias block, [B, h] each)
float* cdae_a1_slot(const CdaeLayout& P, int B, float* workspace) {
  Bump ws(workspace, ~size_t(0));
  for (int l = 0; l <= P.L; ++l) ws.take((size_t)B * P.h);
  return ws.take(1);
}

int cdae_impl(const ardae_cdae_desc* d, const float* params, const float* packed, const float* xbar, const float* sigma,
              const float* eps, const float* ctx, int B, int S, float* workspace, size_t ws_floats, float* loss, float* grads,
              float* score_out, bool need_grads, hipStream_t st, bool a1_ready = false) {
  ARDAE_TRY(desc_ok(d));
  ARDAE_CHECK_ARG(params && packed && xbar && sigma && ctx && workspace, "cdae: null pointer argument");
  ARDAE_CHECK_ARG(B > 0 && S > 0 && (int64_t)B * S < (int64_t)1 << 30, "cdae: bad batch (B=%d, S=%d)", B, S);
  ARDAE_CHECK_ARG(!need_grads || (eps && loss && grads), "cdae: loss/grads/eps must be given");
  ARDAE_CHECK_ARG(need_grads || score_out, "cdae: score_out is NULL");
  const CdaeLayout P(*d);
  const PackedLayout K(P);
  const int N = B * S, h = P.h, L = P.L, z = P.z, act = P.act;
  ARDAE_CHECK_ARG(ws_floats >= workspace_floats(P, B, S, need_grads), "cdae: workspace too small (%zu < %zu floats)", ws_floats,
                  workspace_floats(P, B, S, need_grads));
  Bump ws(workspace, ws_floats);
  const size_t Bh = (size_t)B * h, Nh = (size_t)N * h;
  std::vector<float*> cL(L + 1), a(L + 1), hh(L + 1), e(L + 1), r(L + 1), tau(L + 1), taup(L + 1), pbar(L + 1), qbar(L + 1), chat(L + 1);
  for (int l = 1; l <= L; ++l) cL[l] = ws.take(Bh);
  float* cb = ws.take(Bh);
  for (int l = 1; l <= L; ++l) { a[l] = ws.take(Nh); hh[l] = ws.take(Nh); e[l] = ws.take(Nh); r[l] = ws.take(Nh); }
  if (need_grads)
    for (int l = 1; l <= L; ++l) { tau[l] = ws.take(Nh); taup[l] = ws.take(Nh); pbar[l] = ws.take(Nh); qbar[l] = ws.take(Nh); }
  float* gbar = ws.take((size_t)N * z);
  float* gbuf = ws.take((size_t)N * z);
  const int ltiles = linear_row_tiles(N, z) * linear_col_panels(N, z);
  float* tile_loss = ws.take(ltiles);
  float* chain_cnt = ws.take((size_t)LINEAR_SMALL_CHAIN_COUNTER_WORDS * ((N + 15) / 16));
  float* g = score_out ? score_out : gbuf;

  const float* W1s = packed + K.w1s;
  // ------------------------------------------------------------------ forward: ctx (B rows, once per image) and inp (N rows)
  auto ctx_layer = [&](int l) { LinArgs A{}; A.bias = params + P.ctx[l - 1].b; A.Y = cL[l]; A.ldY = h;
                                return lin_args(act, B, h, l == 1 ? ctx : cL[l - 1], l == 1 ? P.c : h, P.ctx[l - 1].in, packed + K.ctx_f[l - 1], A); };
  auto inp_layer = [&](int l) { LinArgs A{}; A.bias = params + P.inp[l - 1].b; A.Y = a[l]; A.ldY = h;
                                return lin_args(act, N, h, l == 1 ? xbar : a[l - 1], l == 1 ? z : h, P.inp[l - 1].in, packed + K.inp_f[l - 1], A); };
  const float* wfc0 = params + P.neg[L].w;
  if (!need_grads && P.kind == 0 && N == B && linear_small_eligible(inp_layer(1), EPI_ACT)) {
    // The sigma = 0 score pass of the VAE update (glogprob on B rows, models/graddae/mlp.py:446-483): 4 L + 2 per-image problems, every one
    // a link of a dependent chain - ONE launch walks them level by level (linear_small_chain_kernel): [ctx_l | inp_l] l = 1..L, the
    // per-image bias of the first energy layer, the energy layers, the score layers, g.
    std::vector<LinArgs> pr; std::vector<int> ep, lv;
    int level = 0;
    auto add = [&](const LinArgs& A, int epi, int lev) { pr.push_back(A); ep.push_back(epi); lv.push_back(lev); };
    for (int l = 1; l <= L; ++l, ++level) { add(ctx_layer(l), EPI_ACT, level); add(inp_layer(l), EPI_ACT, level); }
    { LinArgs A{}; A.bias = params + P.neg[0].b; A.Y = cb; A.ldY = h; add(lin_args(ACT_NONE, B, h, cL[L], h, h, packed + K.w1c_f, A), EPI_ACT, level++); }
    for (int l = 1; l <= L; ++l) {
      LinArgs A{}; A.Y = hh[l]; A.ldY = h;
      if (l == 1) { A.rowbias = cb; A.rowbias_ld = h; A.rows_per_group = S; A.rowscale = sigma; A.rowscale_w = W1s; } else A.bias = params + P.neg[l - 1].b;
      if (l == L) { A.Y2 = e[L]; A.ldY2 = h; A.R = wfc0; }
      add(lin_args(act, N, h, l == 1 ? a[L] : hh[l - 1], h, h, l == 1 ? packed + K.w1a_f : packed + K.neg_f[l - 1], A), EPI_ACT, level++);
    }
    for (int l = L; l >= 2; --l) { LinArgs A{}; A.S = hh[l - 1]; A.ldS = h; A.Y = e[l - 1]; A.ldY = h; add(lin_args(act, N, h, e[l], h, h, packed + K.neg_b[l - 1], A), EPI_DACT, level++); }
    { LinArgs A{}; A.S = a[L]; A.ldS = h; A.Y = r[L]; A.ldY = h; add(lin_args(act, N, h, e[1], h, h, packed + K.w1a_b, A), EPI_DACT, level++); }
    for (int l = L; l >= 2; --l) { LinArgs A{}; A.S = a[l - 1]; A.ldS = h; A.Y = r[l - 1]; A.ldY = h; add(lin_args(act, N, h, r[l], h, h, packed + K.inp_b[l - 1], A), EPI_DACT, level++); }
    { LinArgs A{}; A.Y = g; A.ldY = z; add(lin_args(ACT_NONE, N, z, r[1], h, h, packed + K.inp_b[0], A), EPI_ACT, level++); }
    return launch_linear_small_chain(pr.data(), ep.data(), lv.data(), (int)pr.size(), chain_cnt, st);
  }
  ARDAE_CHECK_ARG(!a1_ready || a[1] == cdae_a1_slot(P, B, workspace), "cdae: a_1 slot moved (workspace layout and cdae_a1_slot disagree)");
  const bool few_rows = !a1_ready && linear_small_eligible(inp_layer(1), EPI_ACT);
  auto ctx_bias = [&]() {  // per-image bias of the first energy layer: cb = W1c c_L + d_1
    LinArgs A{}; A.bias = params + P.neg[0].b; A.Y = cb; A.ldY = h;
    return lin(EPI_ACT, ACT_NONE, B, h, cL[L], h, h, packed + K.w1c_f, A, st);
  };
  LayerRun run(EPI_ACT, st);      // the forward N-row layers: A_1 (2) .. A_L, then W_1 .. W_L
  if (few_rows) {
    // few rows: the two encoders are independent chains of per-image launches - level l of both in ONE launch
    for (int l = 1; l <= L; ++l) ARDAE_TRY(launch_linear_pair(ctx_layer(l), inp_layer(l), EPI_ACT, st));
    ARDAE_TRY(ctx_bias());
  } else {
    // the whole per-image branch FIRST (context encoder, then the bias it contributes to the first energy layer): the N-row forward
    // layers of both networks then form ONE run (round 4: one multi-layer launch instead of two with a per-image launch between them)
    for (int l = 1; l <= L; ++l) ARDAE_TRY(launch_linear(ctx_layer(l), EPI_ACT, st));
    ARDAE_TRY(ctx_bias());
    for (int l = a1_ready ? 2 : 1; l <= L; ++l) run.v.push_back(inp_layer(l));   // a1_ready: the perturbation kernel has written a_1
  }
  const float* wfc = params + P.neg[L].w;   // grad kind: w [1,h]
  {
    for (int l = 1; l <= L; ++l) {
      LinArgs A{}; A.Y = hh[l]; A.ldY = h;
      if (l == 1) {
        A.rowbias = cb; A.rowbias_ld = h; A.rows_per_group = S; A.rowscale = sigma; A.rowscale_w = W1s;
      } else {
        A.bias = params + P.neg[l - 1].b;
      }
      if (P.kind == 0 && l == L) { A.Y2 = e[L]; A.ldY2 = h; A.R = wfc; }   // e_L = -w (.) s(h_L)
      run.add(act, N, h, l == 1 ? a[L] : hh[l - 1], h, h, l == 1 ? packed + K.w1a_f : packed + K.neg_f[l - 1], A);
    }
    ARDAE_TRY(run.flush());
  }
  const float inv_nz = 1.0f / ((float)N * (float)z);
  if (P.kind == 0) {
    // ---------------------------------------------------------------- score pass (input-gradient of the energy)
    {
      LayerRun run(EPI_DACT, st);
      for (int l = L; l >= 2; --l) {
        LinArgs A{}; A.S = hh[l - 1]; A.ldS = h; A.Y = e[l - 1]; A.ldY = h;
        run.add(act, N, h, e[l], h, h, packed + K.neg_b[l - 1], A);
      }
      {
        LinArgs A{}; A.S = a[L]; A.ldS = h; A.Y = r[L]; A.ldY = h;
        run.add(act, N, h, e[1], h, h, packed + K.w1a_b, A);
      }
      for (int l = L; l >= 2; --l) {
        LinArgs A{}; A.S = a[l - 1]; A.ldS = h; A.Y = r[l - 1]; A.ldY = h;
        run.add(act, N, h, r[l], h, h, packed + K.inp_b[l - 1], A);
      }
      ARDAE_TRY(run.flush());
    }
    if (!need_grads) {   // glogprob: g = r_1 A_1
      LinArgs A{}; A.Y = g; A.ldY = z;
      return lin(EPI_ACT, ACT_NONE, N, z, r[1], h, h, packed + K.inp_b[0], A, st);
    }
    LinArgs A{}; A.sigma = sigma; A.eps = eps; A.ldeps = z; A.scale = inv_nz; A.Y = g; A.ldY = z; A.Y2 = gbar; A.ldY2 = z;
    A.tile_loss = tile_loss;
    ARDAE_TRY(lin(EPI_DAE_LOSS, ACT_NONE, N, z, r[1], h, h, packed + K.inp_b[0], A, st));
  } else {
    if (!need_grads) {
      LinArgs A{}; A.bias = params + P.neg[L].b; A.Y = g; A.ldY = z;
      return lin(EPI_ACT, ACT_NONE, N, z, hh[L], h, h, packed + K.fc_f, A, st);
    }
    LinArgs A{}; A.bias = params + P.neg[L].b; A.sigma = sigma; A.eps = eps; A.ldeps = z; A.scale = inv_nz; A.Y = g; A.ldY = z;
    A.Y2 = gbar; A.ldY2 = z; A.tile_loss = tile_loss;
    ARDAE_TRY(lin(EPI_DAE_LOSS, ACT_NONE, N, z, hh[L], h, h, packed + K.fc_f, A, st));
  }
  ARDAE_TRY(launch_sum_scale(tile_loss, ltiles, inv_nz, loss, st));

  // -------------------------------------------------------------------- backward
  float* Qsum = ws.take(Bh);
  for (int l = 1; l <= L; ++l) chat[l] = ws.take(Bh);
  const int ctiles = linear_row_tiles(N, h);
  float* cs_taup = ws.take((size_t)ctiles * h);
  std::vector<float*>&qhat = qbar, &phat = pbar;   // in-place: qhat_l overwrites qbar_l, phat_l overwrites pbar_l
  if (P.kind == 0) {
    // forward-mode chain through the score pass
    {
      LayerRun run(EPI_CHAIN, st);
      for (int l = 1; l <= L; ++l) {
        LinArgs A{}; A.S = a[l]; A.ldS = h; A.R = r[l]; A.ldR = h; A.Y = tau[l]; A.ldY = h; A.Y2 = pbar[l]; A.ldY2 = h;
        if (l == 1) run.add(act, N, h, gbar, z, z, packed + K.inp_f[0], A);
        else run.add(act, N, h, tau[l - 1], h, h, packed + K.inp_f[l - 1], A);
      }
      for (int l = 1; l <= L; ++l) {
        LinArgs A{}; A.S = hh[l]; A.ldS = h; A.R = e[l]; A.ldR = h; A.Y = taup[l]; A.ldY = h; A.Y2 = qbar[l]; A.ldY2 = h;
        if (l == L) A.colsum = cs_taup;
        run.add(act, N, h, l == 1 ? tau[L] : taup[l - 1], h, h, l == 1 ? packed + K.w1a_f : packed + K.neg_f[l - 1], A);
      }
      ARDAE_TRY(run.flush());
    }
    // wbar = -colsum(tau'_L)  -> grads of neglogprob.fc.weight [1,h]
    ARDAE_TRY(launch_segment_sum(cs_taup, h, 1, ctiles, h, -1.0f, grads + P.neg[L].w, h, st));
    // ordinary backward of the energy chain, seeded ONLY by the qbar_l
    {
      LayerRun run(EPI_DACT, st);
      for (int l = L; l >= 2; --l) {
        LinArgs A{}; A.S = hh[l - 1]; A.ldS = h; A.Q = qbar[l - 1]; A.ldQ = h; A.Y = qhat[l - 1]; A.ldY = h;
        run.add(act, N, h, qhat[l], h, h, packed + K.neg_b[l - 1], A);
      }
      {
        LinArgs A{}; A.S = a[L]; A.ldS = h; A.Q = pbar[L]; A.ldQ = h; A.Y = phat[L]; A.ldY = h;
        run.add(act, N, h, qhat[1], h, h, packed + K.w1a_b, A);
      }
      for (int l = L; l >= 2; --l) {
        LinArgs A{}; A.S = a[l - 1]; A.ldS = h; A.Q = pbar[l - 1]; A.ldQ = h; A.Y = phat[l - 1]; A.ldY = h;
        run.add(act, N, h, phat[l], h, h, packed + K.inp_b[l - 1], A);
      }
      ARDAE_TRY(run.flush());
    }
  } else {
    // direct-score variant: a single ordinary backward from gbar
    {
      LinArgs A{}; A.S = hh[L]; A.ldS = h; A.Y = qhat[L]; A.ldY = h;
      ARDAE_TRY(lin(EPI_DACT, act, N, h, gbar, z, z, packed + K.fc_b, A, st));
    }
    for (int l = L; l >= 2; --l) {
      LinArgs A{}; A.S = hh[l - 1]; A.ldS = h; A.Y = qhat[l - 1]; A.ldY = h;
      ARDAE_TRY(lin(EPI_DACT, act, N, h, qhat[l], h, h, packed + K.neg_b[l - 1], A, st));
    }
    {
      LinArgs A{}; A.S = a[L]; A.ldS = h; A.Y = phat[L]; A.ldY = h;
      ARDAE_TRY(lin(EPI_DACT, act, N, h, qhat[1], h, h, packed + K.w1a_b, A, st));
    }
    for (int l = L; l >= 2; --l) {
      LinArgs A{}; A.S = a[l - 1]; A.ldS = h; A.Y = phat[l - 1]; A.ldY = h;
      ARDAE_TRY(lin(EPI_DACT, act, N, h, phat[l], h, h, packed + K.inp_b[l - 1], A, st));
    }
  }
  // ctx branch: reduce over the S samples of each image first, then B-row back-prop
  ARDAE_TRY(launch_segment_sum(qhat[1], h, B, S, h, 1.0f, Qsum, h, st));
  {
    LinArgs A{}; A.S = cL[L]; A.ldS = h; A.Y = chat[L]; A.ldY = h;
    ARDAE_TRY(lin(EPI_DACT, act, B, h, Qsum, h, h, packed + K.w1c_b, A, st));
  }
  for (int l = L; l >= 2; --l) {
    LinArgs A{}; A.S = cL[l - 1]; A.ldS = h; A.Y = chat[l - 1]; A.ldY = h;
    ARDAE_TRY(lin(EPI_DACT, act, B, h, chat[l], h, h, packed + K.ctx_b[l - 1], A, st));
  }

  // -------------------------------------------------------------------- weight gradients: one batched launch
  std::vector<int> splits;
  wgrad_scratch(P, B, S, &splits);
  std::vector<WgradProblem> probs;
  auto push = [&](int M, int O, int I, const float* G0, const float* X0, int ldX0, const float* G1, const float* X1, int ldX1,
                  int bias_pair, const float* rowscale, float* out, int ldout, float* out_bias, float* out_rs, int ld_rs) {
    WgradProblem p;
    memset(&p, 0, sizeof(p));
    p.M = M; p.O = O; p.I = I;
    p.npairs = G1 ? 2 : 1;
    p.G[0] = G0; p.ldG[0] = O; p.X[0] = X0; p.ldX[0] = ldX0;
    p.G[1] = G1; p.ldG[1] = O; p.X[1] = X1; p.ldX[1] = ldX1;
    p.bias_pair = bias_pair; p.rowscale = rowscale;
    p.splits = splits[probs.size()];
    p.partial = ws.take((size_t)p.splits * O * I);
    p.partial_vec = ws.take((size_t)p.splits * 2 * O);
    p.out = out; p.ldout = ldout; p.out_bias = out_bias; p.out_rowscale = out_rs; p.ld_rowscale = ld_rs; p.beta = 0.f;
    probs.push_back(p);
  };
  const int ld1 = 2 * h + 1;
  float* gW1 = grads + P.neg[0].w;
  if (P.kind == 0) {
    for (int l = 1; l <= L; ++l)   // inp A_l: r_l (x) tau_{l-1}  +  phat_l (x) a_{l-1}   (tau_0 = gbar, a_0 = xbar)
      push(N, h, P.inp[l - 1].in, r[l], l == 1 ? gbar : tau[l - 1], l == 1 ? z : h, phat[l], l == 1 ? xbar : a[l - 1], l == 1 ? z : h, 1,
           nullptr, grads + P.inp[l - 1].w, P.inp[l - 1].in, grads + P.inp[l - 1].b, nullptr, 0);
    push(N, h, h, e[1], tau[L], h, qhat[1], a[L], h, 1, sigma, gW1, ld1, grads + P.neg[0].b, gW1 + 2 * h, ld1);        // W1a, d_1, w1s
    push(B, h, h, Qsum, cL[L], h, nullptr, nullptr, 0, -1, nullptr, gW1 + h, ld1, nullptr, nullptr, 0);                 // W1c
    for (int l = 2; l <= L; ++l)   // W_l: e_l (x) tau'_{l-1} + qhat_l (x) h_{l-1}
      push(N, h, h, e[l], taup[l - 1], h, qhat[l], hh[l - 1], h, 1, nullptr, grads + P.neg[l - 1].w, h, grads + P.neg[l - 1].b, nullptr, 0);
  } else {
    for (int l = 1; l <= L; ++l)
      push(N, h, P.inp[l - 1].in, phat[l], l == 1 ? xbar : a[l - 1], l == 1 ? z : h, nullptr, nullptr, 0, 0, nullptr,
           grads + P.inp[l - 1].w, P.inp[l - 1].in, grads + P.inp[l - 1].b, nullptr, 0);
    push(N, h, h, qhat[1], a[L], h, nullptr, nullptr, 0, 0, sigma, gW1, ld1, grads + P.neg[0].b, gW1 + 2 * h, ld1);
    push(B, h, h, Qsum, cL[L], h, nullptr, nullptr, 0, -1, nullptr, gW1 + h, ld1, nullptr, nullptr, 0);
    for (int l = 2; l <= L; ++l)
      push(N, h, h, qhat[l], hh[l - 1], h, nullptr, nullptr, 0, 0, nullptr, grads + P.neg[l - 1].w, h, grads + P.neg[l - 1].b, nullptr, 0);
    push(N, z, h, gbar, hh[L], h, nullptr, nullptr, 0, 0, nullptr, grads + P.neg[L].w, h, grads + P.neg[L].b, nullptr, 0);   // dae.fc
  }
  for (int l = 1; l <= L; ++l)     // ctx C_l: chat_l (x) c_{l-1}
    push(B, h, P.ctx[l - 1].in, chat[l], l == 1 ? ctx : cL[l - 1], l == 1 ? P.c : h, nullptr, nullptr, 0, 0, nullptr,
         grads + P.ctx[l - 1].w, P.ctx[l - 1].in, grads + P.ctx[l - 1].b, nullptr, 0);
  ARDAE_CHECK_ARG(ws.ok, "cdae: internal workspace accounting error");
  return launch_wgrad_batch(probs.data(), (int)probs.size(), st);
}

}  // namespace
}  // namespace ardae

using namespace ardae;

extern "C" {

size_t ardae_cdae_param_floats(const ardae_cdae_desc* d) {
  if (desc_ok(d) != 0) return 0;
  return CdaeLayout(*d).total;
}
size_t ardae_cdae_packed_floats(const ardae_cdae_desc* d) {
  if (desc_ok(d) != 0) return 0;
  return PackedLayout(CdaeLayout(*d)).total;
}
size_t ardae_cdae_workspace_floats(const ardae_cdae_desc* d, int B, int S, int need_grads) {
  if (desc_ok(d) != 0 || B <= 0 || S <= 0) return 0;
  return workspace_floats(CdaeLayout(*d), B, S, need_grads != 0);
}
int ardae_cdae_pack(const ardae_cdae_desc* d, const float* params, float* packed, void* stream) {
  ARDAE_TRY(desc_ok(d));
  ARDAE_CHECK_ARG(params && packed, "cdae_pack: null pointer");
  const CdaeLayout P(*d);
  return cdae_pack_impl(P, PackedLayout(P), params, packed, (hipStream_t)stream);
}
int ardae_cdae_loss_grads(const ardae_cdae_desc* d, const float* params, const float* packed, const float* xbar,
                          const float* sigma, const float* eps, const float* ctx, int B, int S, float* workspace,
                          size_t workspace_floats, float* loss, float* grads, float* score_out, void* stream) {
  return cdae_impl(d, params, packed, xbar, sigma, eps, ctx, B, S, workspace, workspace_floats, loss, grads, score_out, true,
                   (hipStream_t)stream);
}
int ardae_cdae_perturb_fused_ok(const ardae_cdae_desc* d, int nz, int nstd) {
  if (desc_ok(d) != 0) return 0;
  const CdaeLayout P(*d);
  return nstd == 1 && P.inp[0].in == P.z && latent_perturb_draw_fwd_ok(nz, P.z, P.h, P.act) ? 1 : 0;
}
int ardae_cdae_perturb_loss_grads(const ardae_cdae_desc* d, const float* params, const float* packed, const float* latent, const float* z0,
                                  const float* ctx, int B, int nz, float std_scale, float delta, uint64_t seed, uint64_t offset_xi,
                                  uint64_t offset_eps, const void* state, uint64_t first_row, float* xbar, float* sigma, float* eps_out,
                                  float* std_b, float* workspace, size_t ws_floats, float* loss, float* grads, void* stream) {
  ARDAE_TRY(desc_ok(d));
  ARDAE_CHECK_ARG(ardae_cdae_perturb_fused_ok(d, nz, 1), "cdae_perturb_loss_grads: shape not eligible (nz=%d): use ardae_latent_perturb* + ardae_cdae_loss_grads", nz);
  ARDAE_CHECK_ARG(params && packed && workspace && B > 0, "cdae_perturb_loss_grads: null pointer argument");
  const CdaeLayout P(*d);
  const PackedLayout K(P);
  ARDAE_CHECK_ARG(ws_floats >= workspace_floats(P, B, nz, true), "cdae_perturb_loss_grads: workspace too small");
  ARDAE_TRY(launch_latent_perturb_draw_fwd(latent, z0, B, nz, P.z, std_scale, delta, seed, offset_xi, offset_eps, state, first_row, xbar, sigma,
                                           eps_out, std_b, packed + K.inp_f[0], params + P.inp[0].b, P.h, P.act, cdae_a1_slot(P, B, workspace),
                                           (hipStream_t)stream));
  return cdae_impl(d, params, packed, xbar, sigma, eps_out, ctx, B, nz, workspace, ws_floats, loss, grads, nullptr, true, (hipStream_t)stream,
                   true);
}
int ardae_cdae_score(const ardae_cdae_desc* d, const float* params, const float* packed, const float* x, const float* sigma,
                     const float* ctx, int B, int S, float* workspace, size_t workspace_floats, float* score_out, void* stream) {
  return cdae_impl(d, params, packed, x, sigma, nullptr, ctx, B, S, workspace, workspace_floats, nullptr, nullptr, score_out, false,
                   (hipStream_t)stream);
}

}  // extern "C"
