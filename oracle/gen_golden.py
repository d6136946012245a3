"""Pin the oracle against the reference and write the golden fixtures  --  TEST INFRASTRUCTURE.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box):

    python oracle/gen_golden.py            # validates + (re)writes tests/golden/*.npz

What it does, per case:
  1. builds the reference's own classes (models.MNISTIPVAE / ToyIPVAE / MLPGradCARDAE /
     MLPResCARDAE, utils.Adam, torch.optim.RMSprop) with inert stubs for the plotting
     imports the reference pulls in at module top (torchvision, seaborn),
  2. loads parameters from the oracle's deterministic initialiser into them,
  3. runs the loop body of ivae_ardae.py:713-846 on the reference objects with a fixed torch
     seed, and replays the same seed to capture the noise tensors in the reference's draw order,
  4. runs the oracle (oracle/ardae_oracle.py) on the same parameters + captured noise and
     asserts agreement (fp32: rtol 2e-5 on losses, 1e-4 relative L2 on grads; fp64 case tighter),
  5. stores the REFERENCE's outputs as the fixture.
"""
import os
import sys
import types
import math
import copy

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ardae_oracle as O  # noqa: E402

REF = "/root/reference"
GOLDEN = os.path.join(ROOT, "tests", "golden")


# --------------------------------------------------------------------------- #
def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    noop = lambda *a, **k: None
    if "torchvision" not in sys.modules:
        tv = stub("torchvision")
        tv.utils = stub("torchvision.utils", make_grid=noop, save_image=noop)
        tv.datasets = stub("torchvision.datasets", MNIST=object)
        tv.transforms = stub("torchvision.transforms")
    if "seaborn" not in sys.modules:
        stub("seaborn", set=noop, set_style=noop, set_palette=noop, color_palette=noop, scatterplot=noop)
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    import models as net      # noqa
    import utils as rutils    # noqa
    return net, rutils


def build_reference(net, mc, cc, pm, pc, dtype):
    if mc.kind == "conv":
        model = net.ConvIPVAE(input_height=28, input_channels=1, z_dim=mc.z_dim, noise_dim=mc.noise_dim, nonlinearity=mc.nonlin)
    elif mc.kind == "auxmnist":   # ivae_ardae.py:455-466 with --model-clip-z0-logvar / --model-clip-z-logvar none
        model = net.MNISTAuxIPVAE(input_dim=mc.input_dim, noise_dim=mc.noise_dim, h_dim=mc.h_dim, num_hidden_layers=mc.n_layers,
                                  nonlinearity=mc.nonlin, enc_type="simple", z_dim=mc.z_dim, clip_z0_logvar=mc.clip_z0, clip_z_logvar=mc.clip_z)
    elif mc.kind == "auxconv":   # ivae_ardae.py:467-478
        model = net.MNISTConvAuxIPVAE(input_height=28, input_channels=1, z0_dim=mc.noise_dim, z_dim=mc.z_dim, nonlinearity=mc.nonlin)
    elif mc.kind == "resconv":   # ivae_ardae.py:359-370 (--model resconvct-res)
        model = net.ResConvIPVAE(input_height=28, input_channels=1, z_dim=mc.z_dim, h_dim=mc.h_dim, num_hidden_layers=mc.n_layers,
                                 noise_dim=mc.noise_dim, nonlinearity=mc.nonlin, do_center=mc.do_center, enc_type=mc.enc_type)
    elif mc.kind == "auxresconv":   # ivae_ardae.py:493-505 (--model auxresconvct); :507-534 (--model auxresconv-clip / auxresconvct-clip)
        model = (net.MNISTResConvAuxIPVAEClipped if mc.clipped else net.MNISTResConvAuxIPVAE)(input_height=28, input_channels=1, z_dim=mc.z_dim, c_dim=mc.h_dim, z0_dim=mc.noise_dim,
                                         nonlinearity=mc.nonlin, do_center=mc.do_center)
    elif mc.kind == "auxtoy":   # ivae_ardae.py:443-454 (--model auxmlp) with --model-clip-z0-logvar / --model-clip-z-logvar none
        model = net.ToyAuxIPVAE(input_dim=mc.input_dim, noise_dim=mc.noise_dim, h_dim=mc.h_dim, num_hidden_layers=mc.n_layers,
                                nonlinearity=mc.nonlin, enc_type="simple", z_dim=mc.z_dim, clip_z0_logvar=mc.clip_z0, clip_z_logvar=mc.clip_z)
    elif mc.kind == "mnist":
        model = net.MNISTIPVAE(input_dim=mc.input_dim, noise_dim=mc.noise_dim, h_dim=mc.h_dim,
                               num_hidden_layers=mc.n_layers, nonlinearity=mc.nonlin, enc_type="concat", z_dim=mc.z_dim)
    else:
        model = net.ToyIPVAE(input_dim=mc.input_dim, noise_dim=mc.noise_dim, h_dim=mc.h_dim,
                             num_hidden_layers=mc.n_layers, nonlinearity=mc.nonlin, enc_type="concat", z_dim=mc.z_dim)
    ctor = net.MLPGradCARDAE if cc.kind == "grad" else net.MLPResCARDAE
    cdae = ctor(input_dim=cc.input_dim, context_dim=cc.context_dim, std=1., h_dim=cc.h_dim,
                num_hidden_layers=cc.n_layers, nonlinearity=cc.nonlin, noise_type="gaussian",
                enc_ctx=True, enc_input=True)
    model = model.to(dtype)
    cdae = cdae.to(dtype)
    # names and shapes must coincide with the oracle's spec (this is App. B of SURVEY.md)
    assert [(n, tuple(p.shape)) for n, p in model.named_parameters()] == O.model_param_spec(mc)
    assert [(n, tuple(p.shape)) for n, p in cdae.named_parameters()] == O.cdae_param_spec(cc)
    model.load_state_dict({k: v.clone() for k, v in pm.items()})
    cdae.load_state_dict({k: v.clone() for k, v in pc.items()})
    return model, cdae


def ref_step(rutils, model, cdae, m_opt, c_opt, tc, x_cdae, x_vae, seed):
    """Loop body of ivae_ardae.py:713-846 on the reference's objects (num_cdae_updates=1, lt0)."""
    out = {}
    torch.manual_seed(seed)
    model.train(); cdae.train()
    c_opt.zero_grad()
    B = x_cdae.size(0)
    if tc.ctx_type == "hidden1a":   # ivae_ardae.py:737-739
        context = model.encode.forward_hidden(x_cdae, std=0).detach().unsqueeze(1)
    elif tc.ctx_type == "data":     # ivae_ardae.py:730-734
        context = x_cdae.unsqueeze(1)
        if tc.ctx_center:
            context = (2 * context - 1).view(B, 1, -1)
    else:
        context = model.encode(x_cdae, std=0).detach()
    latent_mean = model.encode(x_cdae, std=0).detach()
    latent = model.forward_hidden(x_cdae, nz=tc.nz_cdae).detach()
    u = tc.std_scale * (latent - latent_mean)
    std_qz = torch.std(u, dim=1, keepdim=True)
    std = tc.delta * torch.mean(std_qz, dim=2, keepdim=True)
    stdmat = std * torch.randn(B, tc.nz_cdae * tc.nstd, 1, dtype=u.dtype)
    u_exp = u.unsqueeze(2).expand(B, tc.nz_cdae, tc.nstd, u.size(-1)).reshape(B, tc.nz_cdae * tc.nstd, u.size(-1))
    _, closs = cdae(u_exp, context, std=stdmat, scale=tc.std_scale)
    closs.backward()
    out["z0"] = latent_mean.clone(); out["latent"] = latent.clone(); out["std"] = std.clone()
    out["u_rows"] = u_exp.reshape(-1, u.size(-1)).clone(); out["sigma_rows"] = stdmat.reshape(-1).clone()
    out["cdae_loss"] = closs.detach().clone()
    out["cdae_grads"] = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in cdae.named_parameters()}
    c_opt.step()
    out["cdae_params_after"] = {n: p.detach().clone() for n, p in cdae.named_parameters()}

    model.train(); cdae.eval()
    m_opt.zero_grad()
    B = x_vae.size(0)
    _, _, latent, mloss, rec, pri = model(x_vae, beta=tc.beta, eta=0., lmbd=0., nz=tc.nz_model)
    mloss.backward(retain_graph=True)
    if tc.ctx_type == "hidden1a":   # ivae_ardae.py:815-817
        context = model.encode.forward_hidden(x_vae, std=0).detach().unsqueeze(1)
    elif tc.ctx_type == "data":     # ivae_ardae.py:809-813
        context = x_vae.unsqueeze(1)
        if tc.ctx_center:
            context = (2 * context - 1).view(B, 1, -1)
    else:
        context = model.encode(x_vae, std=0).detach()
    latent_mean = model.encode(x_vae, std=0).detach()
    lsm = tc.std_scale * (latent - latent_mean).detach()
    stdmat = torch.zeros(B, tc.nz_model, 1, dtype=lsm.dtype)
    g = cdae.glogprob(lsm, context, std=stdmat, scale=tc.std_scale).detach()
    (tc.std_scale * (latent - latent_mean)).backward(tc.beta * g.detach() / float(B * tc.nz_model))
    out["model_loss"] = mloss.detach().clone(); out["recon"] = rec.clone(); out["prior"] = pri.clone()
    out["score"] = g.clone(); out["vae_latent"] = latent.detach().clone()
    out["model_grads"] = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    m_opt.step()
    out["model_params_after"] = {n: p.detach().clone() for n, p in model.named_parameters()}
    return out


def replay_noise(mc, tc, B_c, B_v, seed, dtype):
    """Re-draw, with the same seed and call sizes, what ref_step consumed (SURVEY 8 a-R)."""
    torch.manual_seed(seed)
    if mc.kind in O.AUX_KINDS:
        # one Encoder._forward call (ivae/auxmnist.py:110-116) draws eps0 [R, noise_dim] and eps [R, 1, z]; the two reparam modules
        # it runs then draw samples nobody uses (AuxEncoder.forward: randn_like [B, noise_dim], vae/auxmnist.py:66;
        # SimpleEncoder._forward_all: randn_like [R, z], :189)
        def fwd(B, nz):
            if mc.kind == "auxtoy":     # the model-level calls pass q = int(sqrt(nz)) to Encoder._forward (ivae/auxtoy.py:215,230): q z0's x q z's
                q = math.isqrt(nz)
                e0 = torch.randn(B * q, mc.noise_dim)
                e = torch.randn(B * q, q, mc.z_dim)
                torch.randn(B, mc.noise_dim, dtype=dtype); torch.randn(B * q, mc.z_dim, dtype=dtype)
                return e0, e.reshape(B * q * q, mc.z_dim)
            e0 = torch.randn(B * nz, mc.noise_dim)
            e = torch.randn(B * nz, 1, mc.z_dim)
            torch.randn(B, mc.noise_dim, dtype=dtype); torch.randn(B * nz, mc.z_dim, dtype=dtype)   # randn_like(std): the model's dtype
            return e0, e.reshape(B * nz, mc.z_dim)
        n = {}
        ctx_raw = fwd(B_c, 1)[0]                             # context: forward_hidden(std=0)
        z0_raw = fwd(B_c, 1)[0]                              # latent_mean: encode(std=0)
        n["sampler"], n["sampler_z"] = fwd(B_c, tc.nz_cdae)
        n["sigma"] = torch.randn(B_c, tc.nz_cdae * tc.nstd, 1, dtype=dtype)
        n["eps"] = torch.randn(B_c * tc.nz_cdae * tc.nstd, mc.z_dim, dtype=dtype)
        n["vae"], n["vae_z"] = fwd(B_v, tc.nz_model)
        if mc.clipped:      # the std = 0 calls keep an unscaled eps0 (ivae/auxresconv2.py:91): these draws are used
            n["ctx_raw"], n["z0_raw"] = ctx_raw, z0_raw
            torch.rand(B_v * tc.nz_model, mc.input_dim, dtype=dtype)     # the decoder's unused relaxed-Bernoulli sample (models/reparam.py:113)
            n["vctx_raw"] = fwd(B_v, 1)[0]                   # ivae_ardae.py:815-817
            n["vz0_raw"] = fwd(B_v, 1)[0]                    # :826
        return {k: v.to(dtype) for k, v in n.items()}
    if tc.ctx_type != "data":
        torch.randn(B_c, mc.noise_dim)                   # context encode (x0); --cdae-ctx-type data takes the image: no draw
    torch.randn(B_c, mc.noise_dim)                       # latent_mean encode (x0)
    n = {}
    n["sampler"] = torch.randn(B_c * tc.nz_cdae, mc.noise_dim)
    n["sigma"] = torch.randn(B_c, tc.nz_cdae * tc.nstd, 1, dtype=dtype)
    n["eps"] = torch.randn(B_c * tc.nz_cdae * tc.nstd, mc.z_dim, dtype=dtype)   # randn_like(input)
    n["vae"] = torch.randn(B_v * tc.nz_model, mc.noise_dim)
    # decoder's unused sample: rand_like(logit) (mnist) / randn_like(std) (toy); then 2 x encode(std=0)
    return {k: v.to(dtype) for k, v in n.items()}


def rel_l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def synth_x(mc, B, seed):
    g = torch.Generator().manual_seed(seed)
    if mc.kind in ("mnist", "conv", "auxmnist", "auxconv", "resconv", "auxresconv"):
        p = (torch.rand(mc.input_dim, generator=g) < 0.2).float() * 0.6 + 0.03
        return torch.bernoulli(p.expand(B, -1), generator=g)
    mu = (torch.randint(0, 5, (B, mc.input_dim), generator=g).float() - 2) * 2
    return mu + math.sqrt(0.1) * torch.randn(B, mc.input_dim, generator=g)


# --------------------------------------------------------------------------- #
def run_case(net, rutils, name, mc, cc, tc, B, steps, dtype, store_full, pseed=0, tol32=1e-4):
    tol = tol32 if dtype == torch.float32 else 1e-10
    pm = O.init_params(O.model_param_spec(mc), pseed, O.model_init_special(mc), dtype)
    pc = O.init_params(O.cdae_param_spec(cc), pseed + 1, None, dtype)
    # tame the N(0,1) head so fp32 comparisons are meaningful at tiny nz as well (values still O(1..10))
    model, cdae = build_reference(net, mc, cc, pm, pc, dtype)
    def make_opt(kind, params, lr, beta1):     # ivae_ardae.py:545-556 (model), :612-622 (cDAE); the model's RMSprop takes d_momentum too (:553)
        if kind == "sgd":
            return torch.optim.SGD(params, lr=lr)
        if kind in ("adam", "amsgrad"):
            return rutils.Adam(params, lr=lr, betas=(beta1, 0.999), amsgrad=kind == "amsgrad")
        return torch.optim.RMSprop(params, lr=lr, momentum=tc.d_momentum)
    m_opt = make_opt(tc.m_optimizer, model.parameters(), tc.m_lr, tc.m_beta1)
    c_opt = make_opt(tc.d_optimizer, list(cdae.parameters()), tc.d_lr, tc.d_beta1)
    st_m, st_c = {}, {}
    fx = {"meta_B": np.int64(B), "meta_steps": np.int64(steps), "meta_pseed": np.int64(pseed)}
    if store_full:
        for k, v in pm.items():
            fx[f"pm/{k}"] = v.numpy().copy()
        for k, v in pc.items():
            fx[f"pc/{k}"] = v.numpy().copy()
    worst = 0.0
    for t in range(steps):
        xc = synth_x(mc, B, 1000 + 2 * t).to(dtype)
        xv = synth_x(mc, B, 1001 + 2 * t).to(dtype)
        seed = 4242 + t
        ref = ref_step(rutils, model, cdae, m_opt, c_opt, tc, xc, xv, seed)
        noise = replay_noise(mc, tc, B, B, seed, dtype)
        # ---- oracle on identical inputs --------------------------------------
        closs, gc, std = O.cdae_update_grads(mc, cc, tc, pm, pc, xc, noise)
        errs = {"cdae_loss": abs(float(closs) - float(ref["cdae_loss"])) / abs(float(ref["cdae_loss"])),
                "std": rel_l2(std, ref["std"])}
        for n_, g_ in ref["cdae_grads"].items():
            if g_ is None:
                assert gc[n_] is None, n_
            else:
                errs["gc/" + n_] = rel_l2(gc[n_], g_)
        with torch.no_grad():
            O.optimizer_step(tc.d_optimizer, pc, gc, st_c, tc.d_lr, tc.d_beta1, tc.d_momentum)
        for n_, p_ in ref["cdae_params_after"].items():
            errs["pc/" + n_] = rel_l2(pc[n_], p_)
        mloss, rec, pri, g, gm = O.vae_update_grads(mc, cc, tc, pm, pc, xv, noise)
        errs["model_loss"] = abs(float(mloss) - float(ref["model_loss"])) / abs(float(ref["model_loss"]))
        errs["recon"] = abs(float(rec) - float(ref["recon"])) / abs(float(ref["recon"]))
        errs["prior"] = abs(float(pri) - float(ref["prior"])) / abs(float(ref["prior"]))
        errs["score"] = rel_l2(g, ref["score"])
        for n_, g_ in ref["model_grads"].items():
            errs["gm/" + n_] = rel_l2(gm[n_], g_)
        with torch.no_grad():
            O.optimizer_step(tc.m_optimizer, pm, gm, st_m, tc.m_lr, tc.m_beta1, tc.d_momentum)
        for n_, p_ in ref["model_params_after"].items():
            errs["pm/" + n_] = rel_l2(pm[n_], p_)
        bad = {k: v for k, v in errs.items() if not (v <= tol)}
        worst = max(worst, max(errs.values()))
        assert not bad, f"{name} step {t}: oracle != reference: {bad}"
        # keep the two trajectories locked together: continue from the REFERENCE's parameters
        pm = {k: v.clone() for k, v in ref["model_params_after"].items()}
        pc = {k: v.clone() for k, v in ref["cdae_params_after"].items()}
        # ---- fixture ----------------------------------------------------------
        pre = f"s{t}/"
        fx[pre + "x_cdae"] = xc.numpy(); fx[pre + "x_vae"] = xv.numpy()
        for k, v in noise.items():
            fx[pre + "noise/" + k] = v.numpy()
        for k in ("cdae_loss", "model_loss", "recon", "prior", "std", "score", "z0", "sigma_rows"):
            fx[pre + k] = ref[k].numpy()
        # the exact cDAE inputs the reference saw (u = 1e4 (z - z0) amplifies last-bit differences of the sampler
        # between machines, so tests feed these instead of recomputing them)
        fx[pre + "xbar"] = (ref["u_rows"] + ref["sigma_rows"][:, None] * noise["eps"]).numpy()
        if store_full:
            fx[pre + "latent"] = ref["latent"].numpy()
            fx[pre + "vae_latent"] = ref["vae_latent"].numpy()
        for grp in ("cdae_grads", "model_grads", "cdae_params_after", "model_params_after"):
            for k, v in ref[grp].items():
                if v is None:
                    fx[pre + grp + "/" + k + "/none"] = np.int64(1)
                elif store_full:
                    fx[pre + grp + "/" + k] = v.numpy()
                else:   # summaries only: L2 norm, sum, first 8 elements
                    f = v.flatten().double()
                    fx[pre + grp + "/" + k + "/norm"] = np.float64(f.norm())
                    fx[pre + grp + "/" + k + "/sum"] = np.float64(f.sum())
                    fx[pre + grp + "/" + k + "/head"] = v.flatten()[:8].numpy()
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"[golden] {name}: {steps} step(s), oracle-vs-reference worst rel err {worst:.2e} -> {path} "
          f"({os.path.getsize(path) / 1024:.0f} KiB)")


def run_iwae_case(net, name, mc, B, k, dtype, store_params=True):
    """models/ivae/mnist.py:378-437 with the per-image draws captured by replaying the seed."""
    pm = O.init_params(O.model_param_spec(mc), 7, O.model_init_special(mc), dtype)
    cc = O.CdaeCfg(input_dim=mc.z_dim, context_dim=2 * mc.h_dim if mc.kind in ("auxmnist", "auxconv", "auxtoy") else mc.z_dim, h_dim=32, n_layers=2)
    if mc.kind == "auxresconv":
        cc = O.CdaeCfg(input_dim=mc.z_dim, context_dim=mc.h_dim, h_dim=32, n_layers=2)
    pc = O.init_params(O.cdae_param_spec(cc), 8, None, dtype)
    model, _ = build_reference(net, mc, cc, pm, pc, dtype)
    x = synth_x(mc, B, 55).to(dtype)
    torch.manual_seed(99)
    model.eval()
    with torch.no_grad():
        ref = model.logprob(x, sample_size=k)
    torch.manual_seed(99)
    if mc.kind == "auxtoy":      # Encoder._forward(nz = k): k z0's x k z's per image (ivae/auxtoy.py:313)
        e0 = torch.randn(B * k, mc.noise_dim)
        e = torch.randn(B * k, k, mc.z_dim)
        torch.randn(B, mc.noise_dim, dtype=dtype); torch.randn(B * k, mc.z_dim, dtype=dtype)
        enc = (e0.to(dtype).reshape(B, k, -1), e.to(dtype).reshape(B, k * k, -1))
    elif mc.kind in O.AUX_KINDS:   # ONE Encoder._forward(nz=k) call for all images (ivae/auxmnist.py:314), then the per-image proposals
        e0 = torch.randn(B * k, mc.noise_dim)
        e = torch.randn(B * k, 1, mc.z_dim)
        torch.randn(B, mc.noise_dim, dtype=dtype); torch.randn(B * k, mc.z_dim, dtype=dtype)     # the two unused reparam samples
        enc = (e0.to(dtype).reshape(B, k, -1), e.to(dtype).reshape(B, k, -1))
    else:
        enc = torch.stack([torch.randn(k, mc.noise_dim) for _ in range(B)]).to(dtype)
    # MultivariateNormal.rsample(torch.Size([1, k])) draws standard normals of shape [1, k, z]
    prop = torch.stack([torch.randn(1, k, mc.z_dim, dtype=dtype)[0] for _ in range(B)])
    mine = O.iwae_logprob(mc, pm, x, k, enc, prop)
    err = abs(float(mine) - float(ref)) / abs(float(ref))
    assert err < (1e-4 if dtype == torch.float32 else 1e-9), (float(mine), float(ref))
    fx = {"x": x.numpy(), "prop_noise": prop.numpy(), "logprob": ref.numpy(), "meta_k": np.int64(k)}
    if mc.kind in O.AUX_KINDS:
        fx["enc_noise"], fx["enc_noise_z"] = enc[0].numpy(), enc[1].numpy()
    else:
        fx["enc_noise"] = enc.numpy()
    if store_params:
        for kk, v in pm.items():
            fx["pm/" + kk] = v.numpy()
    else:
        fx["meta_pseed"] = np.int64(7)       # parameters: O.init_params(spec, 7, special) (large model: regenerated, not stored)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"[golden] {name}: IWAE-{k} ref {float(ref):.6f} oracle {float(mine):.6f} rel err {err:.2e}")


def run_ckpt_case(net, rutils, name, mc, cc, tc, B, k_steps, dtype=torch.float32):
    """what the reference's own objects put in a checkpoint (ivae_ardae.py:1117-1139: `model.state_dict()`,
    `model_optimizer.state_dict()` of utils.Adam, `cdae.state_dict()`, `cdae_optimizer.state_dict()` of torch.optim.RMSprop)
    after k_steps steps, as plain arrays + the exact key layout, and the reference's parameters after step k_steps + 1 run
    from that state.  The test rebuilds the dicts, loads them into the fused engine (ArdaeEngine.load_checkpoints) and must
    land on the same step k_steps + 1."""
    import json
    pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc), dtype)
    pc = O.init_params(O.cdae_param_spec(cc), 1, None, dtype)
    model, cdae = build_reference(net, mc, cc, pm, pc, dtype)
    def make_opt(kind, params, lr, beta1):     # ivae_ardae.py:545-556 (model), :612-622 (cDAE); the model's RMSprop takes d_momentum too (:553)
        if kind == "sgd":
            return torch.optim.SGD(params, lr=lr)
        if kind in ("adam", "amsgrad"):
            return rutils.Adam(params, lr=lr, betas=(beta1, 0.999), amsgrad=kind == "amsgrad")
        return torch.optim.RMSprop(params, lr=lr, momentum=tc.d_momentum)
    m_opt = make_opt(tc.m_optimizer, model.parameters(), tc.m_lr, tc.m_beta1)
    c_opt = make_opt(tc.d_optimizer, list(cdae.parameters()), tc.d_lr, tc.d_beta1)
    for t in range(k_steps):
        ref_step(rutils, model, cdae, m_opt, c_opt, tc, synth_x(mc, B, 1000 + 2 * t).to(dtype), synth_x(mc, B, 1001 + 2 * t).to(dtype), 4242 + t)
    fx = {"meta_B": np.int64(B), "meta_k": np.int64(k_steps)}
    layout = {}
    for tag, sd in (("model", model.state_dict()), ("cdae", cdae.state_dict())):
        layout[tag + "_state_dict"] = list(sd.keys())
        for k_, v in sd.items():
            fx[f"{tag}_sd/{k_}"] = v.detach().numpy().copy()
    for tag, osd in (("m_opt", m_opt.state_dict()), ("c_opt", c_opt.state_dict())):
        groups = []
        for g in osd["param_groups"]:
            groups.append({k_: (list(v) if isinstance(v, (tuple, list)) else v) for k_, v in g.items()})
        layout[tag + "_param_groups"] = groups
        layout[tag + "_state_keys"] = {str(i): sorted(st.keys()) for i, st in osd["state"].items()}
        for i, st in osd["state"].items():
            for k_, v in st.items():
                if torch.is_tensor(v):
                    fx[f"{tag}/{i}/{k_}"] = v.detach().numpy().copy()
                    layout.setdefault(tag + "_tensor_fields", {}).setdefault(str(i), []).append(k_)
                elif v is not None:
                    fx[f"{tag}/{i}/{k_}"] = np.float64(v)
    fx["layout_json"] = np.array(json.dumps(layout))
    t = k_steps
    xc, xv = synth_x(mc, B, 1000 + 2 * t).to(dtype), synth_x(mc, B, 1001 + 2 * t).to(dtype)
    seed = 4242 + t
    ref = ref_step(rutils, model, cdae, m_opt, c_opt, tc, xc, xv, seed)
    noise = replay_noise(mc, tc, B, B, seed, dtype)
    fx["x_cdae"], fx["x_vae"] = xc.numpy(), xv.numpy()
    for k_, v in noise.items():
        fx["noise/" + k_] = v.numpy()
    for k_ in ("cdae_loss", "model_loss", "recon", "prior"):
        fx[k_] = ref[k_].numpy()
    for grp in ("cdae_params_after", "model_params_after"):
        for k_, v in ref[grp].items():
            fx[grp + "/" + k_] = v.numpy()
    # the optimiser states after step k + 1 as well (what the engine must write back)
    for tag, osd in (("m_opt_after", m_opt.state_dict()), ("c_opt_after", c_opt.state_dict())):
        for i, st in osd["state"].items():
            for k_, v in st.items():
                if torch.is_tensor(v):
                    fx[f"{tag}/{i}/{k_}"] = v.detach().numpy().copy()
                elif v is not None:
                    fx[f"{tag}/{i}/{k_}"] = np.float64(v)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **fx)
    print(f"[golden] {name}: reference checkpoint after {k_steps} steps + step {k_steps + 1} -> {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    torch.set_num_threads(8)
    net, rutils = import_reference()
    os.makedirs(GOLDEN, exist_ok=True)
    f32, f64 = torch.float32, torch.float64
    only = set(sys.argv[1:])                                  # optional: names of the fixtures to (re)generate
    if only:
        real_run_case, real_iwae = run_case, run_iwae_case
        globals()["run_case"] = lambda net_, ru_, name, *a, **k: real_run_case(net_, ru_, name, *a, **k) if name in only else None
        globals()["run_iwae_case"] = lambda net_, name, *a, **k: real_iwae(net_, name, *a, **k) if name in only else None
        real_ckpt = run_ckpt_case
        globals()["run_ckpt_case"] = lambda net_, ru_, name, *a, **k: real_ckpt(net_, ru_, name, *a, **k) if name in only else None
    tiny_m = O.ModelCfg("mnist", input_dim=24, noise_dim=10, h_dim=64, z_dim=8, n_layers=2, nonlin="softplus")
    tiny_c = O.CdaeCfg("grad", input_dim=8, context_dim=8, h_dim=64, n_layers=3)
    tc = O.TrainCfg(nz_cdae=8)
    # float64 pin of the maths (tight) and float32 fixtures (what the kernels are compared with)
    run_case(net, rutils, "tiny_mnist_grad_f64", tiny_m, tiny_c, tc, B=4, steps=1, dtype=f64, store_full=True)
    run_case(net, rutils, "tiny_mnist_grad", tiny_m, tiny_c, tc, B=4, steps=3, dtype=f32, store_full=True)
    tiny_r = O.CdaeCfg("res", input_dim=8, context_dim=8, h_dim=64, n_layers=3)
    run_case(net, rutils, "tiny_mnist_res", tiny_m, tiny_r, tc, B=4, steps=2, dtype=f32, store_full=True)
    toy_m = O.ModelCfg("toy", input_dim=2, noise_dim=10, h_dim=64, z_dim=2, n_layers=2, nonlin="relu")
    toy_c = O.CdaeCfg("grad", input_dim=2, context_dim=2, h_dim=64, n_layers=3)
    run_case(net, rutils, "tiny_toy_grad", toy_m, toy_c, tc, B=4, steps=2, dtype=f32, store_full=True)
    # the other activations of get_nonlinear_func (utils/models.py:14-32): tanh is the DEFAULT of the reference's model / cDAE classes,
    # relu the default of --model-nonlin / --cdae-nonlin (mlp-grad with a piecewise linear activation: second-order terms vanish)
    for nm, mk, mnl, ck, cnl in (("tiny_toy_tanh", "toy", "tanh", "grad", "tanh"), ("tiny_mnist_elu", "mnist", "elu", "grad", "elu"),
                                 ("tiny_mnist_leaky", "mnist", "leaky_relu", "res", "leaky_relu"), ("tiny_toy_relu_relu", "toy", "relu", "grad", "relu"),
                                 ("tiny_mnist_tanh_res", "mnist", "tanh", "res", "tanh"),
                                 # swish (utils/models.py:8-10): not monotonic - the HIP path keeps the branch in the stored output's lowest bit
                                 ("tiny_mnist_swish", "mnist", "swish", "grad", "swish"), ("tiny_toy_swish_res", "toy", "swish", "res", "swish")):
        m_ = (O.ModelCfg("toy", input_dim=2, noise_dim=10, h_dim=64, z_dim=2, n_layers=2, nonlin=mnl) if mk == "toy"
              else O.ModelCfg("mnist", input_dim=24, noise_dim=10, h_dim=64, z_dim=8, n_layers=2, nonlin=mnl))
        c_ = O.CdaeCfg(ck, input_dim=m_.z_dim, context_dim=m_.z_dim, h_dim=64, n_layers=3, nonlin=cnl)
        run_case(net, rutils, nm, m_, c_, tc, B=4, steps=2, dtype=f32, store_full=True)
    # --cdae-ctx-type data: the (centred) image itself is the context (ivae_ardae.py:730-734,809-813)
    run_case(net, rutils, "tiny_mnist_ctxdata", tiny_m, O.CdaeCfg("grad", input_dim=8, context_dim=24, h_dim=64, n_layers=3),
             O.TrainCfg(nz_cdae=8, ctx_type="data"), B=4, steps=2, dtype=f32, store_full=True)
    run_case(net, rutils, "tiny_toy_ctxdata", toy_m, O.CdaeCfg("res", input_dim=2, context_dim=2, h_dim=64, n_layers=3),
             O.TrainCfg(nz_cdae=8, ctx_type="data", ctx_center=False), B=4, steps=2, dtype=f32, store_full=True)
    # --m-optimizer / --d-optimizer beyond the recipes' adam / rmsprop pair (ivae_ardae.py:545-556,612-622): argparse's own default pair
    # (adam, adam), amsgrad + sgd, and rmsprop for the model (which takes --d-momentum, :553) with amsgrad for the cDAE; 3 steps each
    for nm, mo, do in (("tiny_mnist_opt_adam_adam", "adam", "adam"), ("tiny_mnist_opt_amsgrad_sgd", "amsgrad", "sgd"),
                       ("tiny_mnist_opt_rmsprop_amsgrad", "rmsprop", "amsgrad"), ("tiny_mnist_opt_sgd_rmsprop", "sgd", "rmsprop")):
        run_case(net, rutils, nm, tiny_m, tiny_c, O.TrainCfg(nz_cdae=8, m_optimizer=mo, d_optimizer=do, d_beta1=0.7, m_lr=2e-4, d_lr=3e-4),
                 B=4, steps=3, dtype=f32, store_full=True)
    # --train-nstd-cdae 3: every sample row meets three noise levels (ivae_ardae.py:759-767)
    run_case(net, rutils, "tiny_mnist_nstd3", tiny_m, tiny_c, O.TrainCfg(nz_cdae=8, nstd=3), B=4, steps=2, dtype=f32, store_full=True)
    # full-width networks of BASELINE configs #2 / #1 at a small batch; parameters regenerated from the seed
    cfg2_m = O.ModelCfg("mnist", 784, 100, 256, 32, 2, "softplus")
    cfg2_c = O.CdaeCfg("grad", 32, 32, 256, 3)
    run_case(net, rutils, "cfg2_b8_nz16", cfg2_m, cfg2_c, O.TrainCfg(nz_cdae=16), B=8, steps=2, dtype=f32, store_full=False)
    cfg1_m = O.ModelCfg("toy", 2, 10, 256, 2, 2, "relu")
    cfg1_c = O.CdaeCfg("grad", 2, 2, 256, 3)
    run_case(net, rutils, "cfg1_b8_nz16", cfg1_m, cfg1_c, O.TrainCfg(nz_cdae=16), B=8, steps=2, dtype=f32, store_full=False)
    # BASELINE config #4 model (ConvIPVAE) with a small cDAE; the conv trunk / transposed-conv decoder are fixed-size
    conv_m = O.ModelCfg("conv", 784, 100, 800, 32, 1, "softplus")
    conv_c = O.CdaeCfg("grad", 32, 32, 64, 2)
    run_case(net, rutils, "conv_b4_nz8", conv_m, conv_c, O.TrainCfg(nz_cdae=8), B=4, steps=2, dtype=f32, store_full=False)
    run_iwae_case(net, "iwae_tiny", tiny_m, B=3, k=16, dtype=f64)
    # hierarchical (aux) sampler of the shipped "hierarchical mlp" recipe (run_vae_dbmnist.sh: --model auxmnist --cdae-ctx-type hidden1a):
    # oracle pin only so far - the HIP path for this family is the next row to build
    aux_m = O.ModelCfg("auxmnist", input_dim=24, noise_dim=10, h_dim=48, z_dim=8, n_layers=2, nonlin="softplus")
    aux_c = O.CdaeCfg("grad", input_dim=8, context_dim=96, h_dim=64, n_layers=3)
    aux_t = O.TrainCfg(nz_cdae=8, ctx_type="hidden1a")
    run_case(net, rutils, "tiny_auxmnist_grad_f64", aux_m, aux_c, aux_t, B=4, steps=1, dtype=f64, store_full=True)
    run_case(net, rutils, "tiny_auxmnist_grad", aux_m, aux_c, aux_t, B=4, steps=2, dtype=f32, store_full=True)
    run_iwae_case(net, "iwae_tiny_auxmnist", aux_m, B=3, k=16, dtype=f64)
    # clip_z0_logvar / clip_z_logvar of the hierarchical MLP classes (NormalDistribution.clip_logvar, models/reparam.py:17-41; argparse only
    # offers 'none', the constructors take every choice): one soft clip and one bounded one per fixture
    import dataclasses
    run_case(net, rutils, "tiny_auxmnist_clip", dataclasses.replace(aux_m, clip_z0="spm4", clip_z="2tanh"), aux_c, aux_t, B=4, steps=2, dtype=f32, store_full=True)
    # hierarchical conv model (--model auxconv, run_vae_dbmnist.sh "hierarchical conv"): oracle pin (its HIP path is not built yet)
    auxc_m = O.ModelCfg("auxconv", 784, 100, 800, 32, 1, "softplus")
    auxc_c = O.CdaeCfg("grad", 32, 1600, 64, 2)
    run_case(net, rutils, "auxconv_b4_nz8", auxc_m, auxc_c, O.TrainCfg(nz_cdae=8, ctx_type="hidden1a"), B=4, steps=2, dtype=f32, store_full=False)
    run_iwae_case(net, "iwae_auxconv", auxc_m, B=2, k=64, dtype=f64, store_params=False)
    # the shipped "implicit resconv" / "hierarchical resconv" recipes' model families (run_vae_dbmnist.sh: --model resconvct-res /
    # auxresconvct, --model-nonlin elu, --cdae mlp-res, --std-scale 100, Adam betas (0.9, 0.999), RMSprop momentum 0.9, lr 1e-3 / 1e-4):
    # weight-normalised residual conv trunk + decoder at their fixed sizes, a small mlp-res cDAE
    res_m = O.ModelCfg("resconv", 784, 100, 512, 32, 1, "elu")
    res_c = O.CdaeCfg("res", 32, 32, 64, 2)
    res_t = O.TrainCfg(nz_cdae=8, std_scale=100., m_lr=1e-3, m_beta1=0.9, d_momentum=0.9)
    run_case(net, rutils, "resconv_b4_nz8_f64", res_m, res_c, res_t, B=4, steps=1, dtype=f64, store_full=False)
    # (fp32: the first Adam step at lr 1e-3 is sign-like, two fp32 evaluations of a 45-layer backward differ by ~1e-4 in a few updates)
    run_case(net, rutils, "resconv_b4_nz8", res_m, res_c, res_t, B=4, steps=2, dtype=f32, store_full=False, tol32=5e-4)
    ares_m = O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu")
    ares_c = O.CdaeCfg("res", 32, 450, 64, 2)
    ares_t = O.TrainCfg(nz_cdae=8, std_scale=100., m_lr=1e-3, m_beta1=0.9, d_momentum=0.9, ctx_type="hidden1a")
    run_case(net, rutils, "auxresconv_b4_nz8_f64", ares_m, ares_c, ares_t, B=4, steps=1, dtype=f64, store_full=False)
    run_case(net, rutils, "auxresconv_b4_nz8", ares_m, ares_c, ares_t, B=4, steps=2, dtype=f32, store_full=False, tol32=5e-4)
    # --model resconv-res / auxresconv (ivae_ardae.py:347-358,479-492): the same families with do_center=False (the trunk sees x, not 2x - 1)
    run_case(net, rutils, "resconv_nocenter_b4_nz8", O.ModelCfg("resconv", 784, 100, 512, 32, 1, "elu", do_center=False), res_c, res_t,
             B=4, steps=2, dtype=f32, store_full=False, tol32=5e-4)
    run_case(net, rutils, "auxresconv_nocenter_b4_nz8", O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu", do_center=False), ares_c, ares_t,
             B=4, steps=2, dtype=f32, store_full=False, tol32=5e-4)
    # the other sampler heads of ResConvIPVAE (ivae_ardae.py:323-346,371-442: --model resconv / resconvct 'mlp', -res2 'res-mlp', -res3
    # 'res-wn-mlp-lin', -res4 'res-mlp-lin'), centred and not, one and two hidden layers (two: ResLinear blocks with identity skips)
    for nm, et, nl, ctr in (("resconv_mlp_b4_nz8", "mlp", 1, True), ("resconv_mlp2_nocenter_b4_nz8", "mlp", 2, False),
                            ("resconv_res2_b4_nz8", "res-mlp", 1, False), ("resconv_res2x2_b4_nz8", "res-mlp", 2, True),
                            ("resconv_res3_b4_nz8", "res-wn-mlp-lin", 1, True), ("resconv_res3x2_b4_nz8", "res-wn-mlp-lin", 2, False),
                            ("resconv_res4_b4_nz8", "res-mlp-lin", 1, False), ("resconv_resx2_b4_nz8", "res-wn-mlp", 2, True)):
        run_case(net, rutils, nm, O.ModelCfg("resconv", 784, 100, 512, 32, nl, "elu", do_center=ctr, enc_type=et), res_c, res_t,
                 B=4, steps=2, dtype=f32, store_full=False, tol32=5e-4)
    run_case(net, rutils, "resconv_res2_b4_nz8_f64", O.ModelCfg("resconv", 784, 100, 512, 32, 1, "elu", do_center=False, enc_type="res-mlp"), res_c, res_t,
             B=4, steps=1, dtype=f64, store_full=False)
    run_iwae_case(net, "iwae_resconv_mlp", O.ModelCfg("resconv", 784, 100, 512, 32, 1, "elu", enc_type="mlp"), B=2, k=64, dtype=f64, store_params=False)
    run_iwae_case(net, "iwae_resconv", res_m, B=2, k=64, dtype=f64, store_params=False)
    run_iwae_case(net, "iwae_auxresconv", ares_m, B=2, k=64, dtype=f64, store_params=False)
    # --model auxmlp (ivae_ardae.py:443-454): ToyAuxIPVAE - the toy problem's hierarchical sampler, q z0's x q z's per image (nz_cdae = q^2),
    # Gaussian decoder, tanh, --cdae-ctx-type hidden1a
    atoy_m = O.ModelCfg("auxtoy", 2, 2, 32, 2, 2, "tanh")
    atoy_c = O.CdaeCfg("grad", input_dim=2, context_dim=64, h_dim=64, n_layers=3, nonlin="softplus")
    atoy_t = O.TrainCfg(nz_cdae=16, ctx_type="hidden1a")
    run_case(net, rutils, "tiny_auxtoy_grad_f64", atoy_m, atoy_c, atoy_t, B=4, steps=1, dtype=f64, store_full=True)
    run_case(net, rutils, "tiny_auxtoy_grad", atoy_m, atoy_c, atoy_t, B=4, steps=3, dtype=f32, store_full=True)
    run_iwae_case(net, "iwae_tiny_auxtoy", atoy_m, B=3, k=8, dtype=f64)
    run_case(net, rutils, "tiny_auxtoy_clip", dataclasses.replace(atoy_m, clip_z0="hard", clip_z="softplus"), atoy_c, atoy_t, B=4, steps=2, dtype=f32, store_full=True)
    # --model auxresconv-clip / auxresconvct-clip (ivae_ardae.py:507-534): MNISTResConvAuxIPVAEClipped - unclipped log-variances, z0 keeps an
    # unscaled eps0 (min_std = 1), so the std = 0 calls of the loop are random draws and their eps0 are part of the fixture's noise
    clip_m = O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu", clipped=True)
    run_case(net, rutils, "auxresconv_clip_b4_nz8_f64", clip_m, ares_c, ares_t, B=4, steps=1, dtype=f64, store_full=False)
    run_case(net, rutils, "auxresconv_clip_b4_nz8", clip_m, ares_c, ares_t, B=4, steps=2, dtype=f32, store_full=False, tol32=5e-4)
    run_case(net, rutils, "auxresconv_clip_nocenter_b4_nz8", O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu", do_center=False, clipped=True), ares_c,
             ares_t, B=4, steps=2, dtype=f32, store_full=False, tol32=5e-4)
    run_iwae_case(net, "iwae_auxresconv_clip", clip_m, B=2, k=64, dtype=f64, store_params=False)
    # reference-written checkpoint (model / cDAE state_dict + utils.Adam / torch.optim.RMSprop state_dict) and the step after it
    run_ckpt_case(net, rutils, "ckpt_tiny_mnist_grad", tiny_m, tiny_c, tc, B=4, k_steps=2)


if __name__ == "__main__":
    main()
