"""Whole-step parity: the fused engine and the drop-in module surface against the reference's golden trajectories.

Tolerances: losses 1e-4 relative (north star); recon/prior 2e-5; gradients / parameter updates as in test_cdae_gpu.py
(the reference's own fp32 path is ~1e-4 from float64 on this ill-conditioned loss, so updates are compared at 2e-3).
"""
import os

import numpy as np
import pytest
import torch

import ardae_amd as net
from oracle import ardae_oracle as O

pytestmark = pytest.mark.gpu

CASES = {
    "tiny_mnist_grad": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 8, 64, 3), 8, True),
    "tiny_mnist_res": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("res", 8, 8, 64, 3), 8, True),
    "tiny_toy_grad": (O.ModelCfg("toy", 2, 10, 64, 2, 2, "relu"), O.CdaeCfg("grad", 2, 2, 64, 3), 8, True),
    "cfg2_b8_nz16": (O.ModelCfg("mnist", 784, 100, 256, 32, 2, "softplus"), O.CdaeCfg("grad", 32, 32, 256, 3), 16, False),
    "cfg1_b8_nz16": (O.ModelCfg("toy", 2, 10, 256, 2, 2, "relu"), O.CdaeCfg("grad", 2, 2, 256, 3), 16, False),
    "conv_b4_nz8": (O.ModelCfg("conv", 784, 100, 800, 32, 1, "softplus"), O.CdaeCfg("grad", 32, 32, 64, 2), 8, False),   # cfg #4 model
    # the shipped "hierarchical mlp" recipe's model family: aux sampler + hidden1a context (run_vae_dbmnist.sh --model auxmnist)
    "tiny_auxmnist_grad": (O.ModelCfg("auxmnist", 24, 10, 48, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 96, 64, 3), 8, True),
    # --model auxmlp (ToyAuxIPVAE, ivae_ardae.py:443-454): q z0's x q z's per image (nz_cdae 16 = 4 x 4), Gaussian decoder, tanh
    "tiny_auxtoy_grad": (O.ModelCfg("auxtoy", 2, 2, 32, 2, 2, "tanh"), O.CdaeCfg("grad", 2, 64, 64, 3), 16, True),
    # clip_z0_logvar / clip_z_logvar of the two hierarchical MLP classes (NormalDistribution.clip_logvar, models/reparam.py:17-41; round 4)
    "tiny_auxmnist_clip": (O.ModelCfg("auxmnist", 24, 10, 48, 8, 2, "softplus", clip_z0="spm4", clip_z="2tanh"), O.CdaeCfg("grad", 8, 96, 64, 3), 8, True),
    "tiny_auxtoy_clip": (O.ModelCfg("auxtoy", 2, 2, 32, 2, 2, "tanh", clip_z0="hard", clip_z="softplus"), O.CdaeCfg("grad", 2, 64, 64, 3), 16, True),
    # the shipped "hierarchical conv" recipe's model family (run_vae_dbmnist.sh --model auxconv, hidden1a context of 1600 columns)
    "auxconv_b4_nz8": (O.ModelCfg("auxconv", 784, 100, 800, 32, 1, "softplus"), O.CdaeCfg("grad", 32, 1600, 64, 2), 8, False),
    # the shipped "implicit resconv" / "hierarchical resconv" recipes' model families (--model resconvct-res / auxresconvct, ELU,
    # mlp-res cDAE, --std-scale 100, Adam (0.9, 0.999) lr 1e-3, RMSprop momentum 0.9; the second with the hidden1a context of 450 columns)
    "resconv_b4_nz8": (O.ModelCfg("resconv", 784, 100, 512, 32, 1, "elu"), O.CdaeCfg("res", 32, 32, 64, 2), 8, False),
    "auxresconv_b4_nz8": (O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu"), O.CdaeCfg("res", 32, 450, 64, 2), 8, False),
}
# the other activations of get_nonlinear_func: tanh (class default of the reference's models / cDAEs), relu in mlp-grad (the default
# of --cdae-nonlin), elu, leaky_relu - on the generic kernels
CASES.update({
    "tiny_toy_tanh": (O.ModelCfg("toy", 2, 10, 64, 2, 2, "tanh"), O.CdaeCfg("grad", 2, 2, 64, 3, "tanh"), 8, True),
    "tiny_mnist_elu": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "elu"), O.CdaeCfg("grad", 8, 8, 64, 3, "elu"), 8, True),
    "tiny_mnist_leaky": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "leaky_relu"), O.CdaeCfg("res", 8, 8, 64, 3, "leaky_relu"), 8, True),
    "tiny_toy_relu_relu": (O.ModelCfg("toy", 2, 10, 64, 2, 2, "relu"), O.CdaeCfg("grad", 2, 2, 64, 3, "relu"), 8, True),
    "tiny_mnist_tanh_res": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "tanh"), O.CdaeCfg("res", 8, 8, 64, 3, "tanh"), 8, True),
    # swish (utils/models.py:8-10): the seventh and last name of get_nonlinear_func (round 4)
    "tiny_mnist_swish": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "swish"), O.CdaeCfg("grad", 8, 8, 64, 3, "swish"), 8, True),
    "tiny_toy_swish_res": (O.ModelCfg("toy", 2, 10, 64, 2, 2, "swish"), O.CdaeCfg("res", 2, 2, 64, 3, "swish"), 8, True),
})
# --cdae-ctx-type data (ivae_ardae.py:730-734,809-813): the image itself (centred for the MNIST family) is the cDAE's context
CASES.update({
    "tiny_mnist_ctxdata": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 24, 64, 3), 8, True),
    "tiny_toy_ctxdata": (O.ModelCfg("toy", 2, 10, 64, 2, 2, "relu"), O.CdaeCfg("res", 2, 2, 64, 3), 8, True),
})
CASES.update({nm: (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 8, 64, 3), 8, True)
              for nm in ("tiny_mnist_opt_adam_adam", "tiny_mnist_opt_amsgrad_sgd", "tiny_mnist_opt_rmsprop_amsgrad", "tiny_mnist_opt_sgd_rmsprop")})
CASES["tiny_mnist_nstd3"] = (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 8, 64, 3), 8, True)
# --model resconv-res / auxresconv: do_center=False (ivae_ardae.py:347-358,479-492)
CASES["resconv_nocenter_b4_nz8"] = (O.ModelCfg("resconv", 784, 100, 512, 32, 1, "elu", do_center=False), O.CdaeCfg("res", 32, 32, 64, 2), 8, False)
CASES["auxresconv_nocenter_b4_nz8"] = (O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu", do_center=False), O.CdaeCfg("res", 32, 450, 64, 2), 8, False)
# the other sampler heads of ResConvIPVAE (ivae_ardae.py:323-346,371-442): --model resconv / resconvct ('mlp'), -res2 ('res-mlp'), -res3
# ('res-wn-mlp-lin'), -res4 ('res-mlp-lin'), one and two hidden layers (two: ResLinear blocks with identity skips), centred and not
RESCONV_HEAD_CASES = {"resconv_mlp_b4_nz8": ("mlp", 1, True), "resconv_mlp2_nocenter_b4_nz8": ("mlp", 2, False), "resconv_res2_b4_nz8": ("res-mlp", 1, False),
                      "resconv_res2x2_b4_nz8": ("res-mlp", 2, True), "resconv_res3_b4_nz8": ("res-wn-mlp-lin", 1, True),
                      "resconv_res3x2_b4_nz8": ("res-wn-mlp-lin", 2, False), "resconv_res4_b4_nz8": ("res-mlp-lin", 1, False),
                      "resconv_resx2_b4_nz8": ("res-wn-mlp", 2, True)}
for _nm, (_et, _nl, _ctr) in RESCONV_HEAD_CASES.items():
    CASES[_nm] = (O.ModelCfg("resconv", 784, 100, 512, 32, _nl, "elu", do_center=_ctr, enc_type=_et), O.CdaeCfg("res", 32, 32, 64, 2), 8, False)
# --model auxresconv-clip / auxresconvct-clip (ivae_ardae.py:507-534): MNISTResConvAuxIPVAEClipped - unclipped log-variances, z0 keeps an unscaled eps0,
# so the std = 0 calls of the loop are random draws whose eps0 the fixtures carry (noise/ctx_raw, z0_raw, vctx_raw, vz0_raw)
CASES["auxresconv_clip_b4_nz8"] = (O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu", clipped=True), O.CdaeCfg("res", 32, 450, 64, 2), 8, False)
CASES["auxresconv_clip_nocenter_b4_nz8"] = (O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu", do_center=False, clipped=True), O.CdaeCfg("res", 32, 450, 64, 2), 8, False)
RES_RECIPE = dict(std_scale=100., m_lr=1e-3, m_beta1=0.9, d_momentum=0.9)     # run_vae_dbmnist.sh, the two resconv lines


def build(mc, cc):
    if mc.kind == "resconv":
        model = net.ResConvIPVAE(input_height=28, input_channels=1, z_dim=mc.z_dim, h_dim=mc.h_dim, num_hidden_layers=mc.n_layers, noise_dim=mc.noise_dim,
                                 nonlinearity=mc.nonlin, do_center=mc.do_center, enc_type=mc.enc_type)
    elif mc.kind == "auxresconv":
        model = (net.MNISTResConvAuxIPVAEClipped if mc.clipped else net.MNISTResConvAuxIPVAE)(
            input_height=28, input_channels=1, z_dim=mc.z_dim, c_dim=mc.h_dim, z0_dim=mc.noise_dim, nonlinearity=mc.nonlin, do_center=mc.do_center)
    elif mc.kind == "auxconv":
        model = net.MNISTConvAuxIPVAE(input_height=28, input_channels=1, z0_dim=mc.noise_dim, z_dim=mc.z_dim, nonlinearity=mc.nonlin)
    elif mc.kind == "auxtoy":
        model = net.ToyAuxIPVAE(input_dim=mc.input_dim, noise_dim=mc.noise_dim, h_dim=mc.h_dim, num_hidden_layers=mc.n_layers,
                                nonlinearity=mc.nonlin, enc_type="simple", z_dim=mc.z_dim, clip_z0_logvar=mc.clip_z0, clip_z_logvar=mc.clip_z)
    elif mc.kind == "auxmnist":
        model = net.MNISTAuxIPVAE(input_dim=mc.input_dim, noise_dim=mc.noise_dim, h_dim=mc.h_dim, num_hidden_layers=mc.n_layers,
                                  nonlinearity=mc.nonlin, enc_type="simple", z_dim=mc.z_dim, clip_z0_logvar=mc.clip_z0, clip_z_logvar=mc.clip_z)
    elif mc.kind == "conv":
        model = net.ConvIPVAE(input_height=28, input_channels=1, z_dim=mc.z_dim, noise_dim=mc.noise_dim, nonlinearity=mc.nonlin)
    else:
        ctor = net.MNISTIPVAE if mc.kind == "mnist" else net.ToyIPVAE
        model = ctor(input_dim=mc.input_dim, noise_dim=mc.noise_dim, h_dim=mc.h_dim, num_hidden_layers=mc.n_layers,
                     nonlinearity=mc.nonlin, enc_type="concat", z_dim=mc.z_dim)
    cctor = net.MLPGradCARDAE if cc.kind == "grad" else net.MLPResCARDAE
    cdae = cctor(input_dim=cc.input_dim, context_dim=cc.context_dim, std=1., h_dim=cc.h_dim, num_hidden_layers=cc.n_layers,
                 nonlinearity=cc.nonlin, noise_type="gaussian", enc_ctx=True, enc_input=True)
    return model, cdae


def load_case(golden_dir, name):
    mc, cc, nz, full = CASES[name]
    fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    if full:
        pm = {n: torch.tensor(fx["pm/" + n]) for n, _ in O.model_param_spec(mc)}
        pc = {n: torch.tensor(fx["pc/" + n]) for n, _ in O.cdae_param_spec(cc)}
    else:
        pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc))
        pc = O.init_params(O.cdae_param_spec(cc), 1)
    return mc, cc, nz, full, fx, pm, pc


def rel(a, b):
    return abs(float(a) - float(b)) / abs(float(b))


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def assert_update_close(after, before, ref_after, what, sgd=False):
    """RMSprop / Adam updates are sign-like in their first steps (|update| ~ lr whatever |grad| is), so an element whose
    gradient is at the fp32 noise floor may flip: compare robustly (median element error + a loose L2), and check the
    optimiser arithmetic itself bit-tightly in test_optimizer_kernels_* with identical gradients."""
    upd, ref = (after - before).double().cpu(), (ref_after - before).double().cpu()
    # both sides round the NEW PARAMETER to fp32: an update of a few ulps of the parameter (plain SGD with a small gradient) carries
    # that rounding as an absolute error
    ulp = 2.0 ** -23 * before.double().cpu().abs()
    err = ((upd - ref).abs() - 2 * ulp).clamp(min=0) / (ref.abs() + 1e-12)
    assert float(err.median()) < 1e-3, what
    if sgd:     # the update IS the gradient: small entries carry the gradient's own fp32 cancellation noise, so the norm decides
        assert rel_l2(upd, ref) < 5e-3, what     # gradient tolerance of the cDAE tests (2e-3) + the fp32 rounding of the updated parameters
        return
    assert float((err > 1e-2).double().mean()) < 2e-2, what
    assert rel_l2(upd, ref) < 5e-2, what


def noise_of(fx, t, dev, blocks=False):
    n = {k: torch.tensor(fx[f"s{t}/noise/{k}"]) for k in ("sampler", "sigma", "eps", "vae")}
    for k in ("sampler", "vae"):        # aux models: the second draw of a sampler call sits beside the first, rows [eps0 | eps]
        if f"s{t}/noise/{k}_z" in fx:
            ez = torch.tensor(fx[f"s{t}/noise/{k}_z"])
            if blocks:                          # ToyAuxIPVAE: eps0 [B q, nd] and eps [B q q, z] - two blocks, one after the other
                n[k] = torch.cat([n[k].reshape(-1), ez.reshape(-1)])
            else:
                n[k] = torch.cat([n[k], ez], 1)
    for k in ("ctx_raw", "z0_raw", "vctx_raw", "vz0_raw"):      # the clipped class: unscaled eps0 of the std = 0 calls
        if f"s{t}/noise/{k}" in fx:
            n[k] = torch.tensor(fx[f"s{t}/noise/{k}"])
    return {k: v.to(dev).contiguous() for k, v in n.items()}


CTX_DATA = {"tiny_mnist_ctxdata": True, "tiny_toy_ctxdata": False}      # --cdae-ctx-type data fixtures: centred (2x - 1) or not
# --m-optimizer / --d-optimizer pairs other than the recipes' adam / rmsprop (ivae_ardae.py:545-556,612-622), as oracle/gen_golden.py ran them
OPT_PAIRS = {"tiny_mnist_opt_adam_adam": ("adam", "adam"), "tiny_mnist_opt_amsgrad_sgd": ("amsgrad", "sgd"),
             "tiny_mnist_opt_rmsprop_amsgrad": ("rmsprop", "amsgrad"), "tiny_mnist_opt_sgd_rmsprop": ("sgd", "rmsprop")}
OPT_KW = dict(d_beta1=0.7, m_lr=2e-4, d_lr=3e-4)


def train_config(mc, nz, name=None, **kw):
    if mc.kind in ("resconv", "auxresconv"):
        kw = dict(RES_RECIPE, **kw)
    if name in CTX_DATA:
        return net.TrainConfig(nz_cdae=nz, cdae_ctx_type="data", ctx_data_center=CTX_DATA[name], **kw)
    if name == "tiny_mnist_nstd3":          # --train-nstd-cdae 3
        return net.TrainConfig(nz_cdae=nz, nstd_cdae=3, **kw)
    if name in OPT_PAIRS:
        return net.TrainConfig(nz_cdae=nz, m_optimizer=OPT_PAIRS[name][0], d_optimizer=OPT_PAIRS[name][1], **dict(OPT_KW, **kw))
    return net.TrainConfig(nz_cdae=nz, cdae_ctx_type="hidden1a" if mc.kind in O.AUX_KINDS else "lt0", **kw)


@pytest.mark.parametrize("name", list(CASES))
def test_engine_trajectory_golden(golden_dir, name):
    mc, cc, nz, full, fx, pm, pc = load_case(golden_dir, name)
    model, cdae = build(mc, cc)
    assert [k for k in model.state_dict()] == [n for n, _ in O.model_param_spec(mc)]      # reference checkpoint keys
    assert [k for k in cdae.state_dict()] == [n for n, _ in O.cdae_param_spec(cc)]
    model.load_state_dict(pm); cdae.load_state_dict(pc)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    B, steps = int(fx["meta_B"]), int(fx["meta_steps"])
    eng = net.ArdaeEngine(model, cdae, train_config(mc, nz, name), batch_size=B)
    for t in range(steps):
        pre = f"s{t}/"
        xc, xv = torch.tensor(fx[pre + "x_cdae"]).cuda(), torch.tensor(fx[pre + "x_vae"]).cuda()
        before_c, before_m = cdae.flat_params().clone(), model.flat_params().clone()
        eng.step(xc, xv, noise=noise_of(fx, t, "cuda", blocks=mc.kind == "auxtoy"))
        s = eng.stats()
        # sampler + latent statistics
        assert rel_l2(eng.z0, fx[pre + "z0"].reshape(B, -1)) < 1e-5
        assert rel_l2(eng.std_b, fx[pre + "std"].reshape(-1)) < 2e-4
        # losses (north star: 1e-4 relative)
        assert rel(s["cdae_loss"], fx[pre + "cdae_loss"]) < 1e-4
        assert rel(s["model_loss"], fx[pre + "model_loss"]) < 1e-4
        assert rel(s["recon"], fx[pre + "recon"]) < 2e-5
        assert rel(s["prior"], fx[pre + "prior"]) < 2e-5
        # parameter updates of both optimisers
        if full:
            ref_c = torch.cat([torch.tensor(fx[pre + "cdae_params_after/" + n]).reshape(-1) for n, _ in O.cdae_param_spec(cc)])
            ref_m = torch.cat([torch.tensor(fx[pre + "model_params_after/" + n]).reshape(-1) for n, _ in O.model_param_spec(mc)])
            assert_update_close(cdae.flat_params().cpu(), before_c.cpu(), ref_c, "cdae update", sgd=OPT_PAIRS.get(name, ("", ""))[1] == "sgd")
            assert_update_close(model.flat_params().cpu(), before_m.cpu(), ref_m, "model update", sgd=OPT_PAIRS.get(name, ("", ""))[0] == "sgd")
            if cc.kind == "grad":     # neglogprob.fc.bias: no gradient in the reference -> never touched
                assert float(cdae.flat_params()[-1]) == float(before_c[-1])
            # continue from the reference's parameters, as the fixture chain does
            with torch.no_grad():
                cdae.flat_params().copy_(ref_c.cuda()); model.flat_params().copy_(ref_m.cuda())
            eng.repack()
        else:
            for grp, spec, flat in (("cdae_params_after", O.cdae_param_spec(cc), cdae.flat_params()),
                                    ("model_params_after", O.model_param_spec(mc), model.flat_params())):
                off = 0
                for n, shp in spec:
                    k = int(np.prod(shp))
                    key = pre + grp + "/" + n
                    if key + "/norm" in fx:
                        assert abs(float(flat[off:off + k].double().norm()) - float(fx[key + "/norm"])) / float(fx[key + "/norm"]) < 1e-4, n
                    off += k
            break   # summaries-only fixtures: parameters cannot be re-synchronised, so only the first step is comparable


@pytest.mark.parametrize("kind", ["grad", "res"])
def test_engine_step_production_kernels_vs_oracle(kind):
    """Config #2 network (784 / 100 / h 256 / z 32, cDAE h 256 L 3) on 128 images x 256 samples = 32768 rows: the size from
    which every N-row launch runs on the production kernels (software-pipelined linear, 256x256 / 256x32 weight
    gradients) instead of the generic ones the small fixtures exercise.  One full step (cDAE update + VAE update) against
    the oracle run live on the CPU with the same parameters, images and noise."""
    mc = O.ModelCfg("mnist", 784, 100, 256, 32, 2, "softplus")
    cc = O.CdaeCfg(kind, 32, 32, 256, 3)
    tc = O.TrainCfg(nz_cdae=256)
    B = 128
    pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc))
    pc = O.init_params(O.cdae_param_spec(cc), 1)
    gen = torch.Generator().manual_seed(11)
    x1 = torch.bernoulli(torch.full((B, 784), 0.2), generator=gen)
    x2 = torch.bernoulli(torch.full((B, 784), 0.2), generator=gen)
    noise = O.draw_step_noise(mc, tc, B, gen)
    model, cdae = build(mc, cc)
    model.load_state_dict(pm); cdae.load_state_dict(pc)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    before_c, before_m = cdae.flat_params().clone().cpu(), model.flat_params().clone().cpu()
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=256), batch_size=B)
    eng.step(x1.cuda(), x2.cuda(), noise={k: v.cuda().contiguous() for k, v in noise.items()})
    got = eng.stats()
    rm, rc = {k: v.clone() for k, v in pm.items()}, {k: v.clone() for k, v in pc.items()}
    ref = O.train_step(mc, cc, tc, rm, rc, {}, {}, x1, x2, noise)
    for k in ("cdae_loss", "model_loss"):
        assert rel(got[k], ref[k]) < 1e-4, k
    for k in ("recon", "prior"):
        assert rel(got[k], ref[k]) < 2e-5, k
    ref_c = torch.cat([rc[n].reshape(-1) for n, _ in O.cdae_param_spec(cc)])
    ref_m = torch.cat([rm[n].reshape(-1) for n, _ in O.model_param_spec(mc)])
    assert_update_close(cdae.flat_params().cpu(), before_c, ref_c, "cdae update")
    assert_update_close(model.flat_params().cpu(), before_m, ref_m, "model update")


@pytest.mark.parametrize("name", ["tiny_mnist_grad", "tiny_toy_grad", "tiny_mnist_res", "tiny_auxmnist_grad", "tiny_auxtoy_grad", "tiny_auxmnist_clip", "tiny_auxtoy_clip",
                                  "tiny_toy_tanh", "tiny_mnist_elu",
                                  "tiny_mnist_leaky", "tiny_toy_relu_relu", "tiny_mnist_tanh_res", "tiny_mnist_swish", "tiny_toy_swish_res"])
def test_vae_phase_grads_golden(golden_dir, name):
    mc, cc, nz, full, fx, pm, pc = load_case(golden_dir, name)
    model, cdae = build(mc, cc)
    model.load_state_dict(pm); cdae.load_state_dict(pc)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    # the VAE phase of step 0 runs AFTER the cDAE update: load the reference's post-update cDAE parameters
    with torch.no_grad():
        cdae.flat_params().copy_(torch.cat([torch.tensor(fx["s0/cdae_params_after/" + n]).reshape(-1) for n, _ in O.cdae_param_spec(cc)]).cuda())
    B = int(fx["meta_B"])
    eng = net.ArdaeEngine(model, cdae, train_config(mc, nz), batch_size=B)
    eng.vae_phase(torch.tensor(fx["s0/x_vae"]).cuda(), noise=noise_of(fx, 0, "cuda", blocks=mc.kind == "auxtoy"), apply_update=False)
    torch.cuda.synchronize()
    assert rel_l2(eng.zv, fx["s0/vae_latent"].reshape(B, -1)) < 1e-5
    # score at sigma=0 (glogprob); with the split backward the seed buffer holds g itself (the factor s*beta/(B nz) is applied
    # when it is added to dL/dz), with the single-call backward g * s*beta/(B nz)
    g = eng.g.cpu() if eng.split_backward else eng.g.cpu() / (1e4 * 1.0 / B)
    assert rel_l2(g, fx["s0/score"].reshape(B, -1)) < 2e-3
    off = 0
    for n, shp in O.model_param_spec(mc):
        k = int(np.prod(shp))
        assert rel_l2(eng.grads_m[off:off + k].cpu(), torch.tensor(fx["s0/model_grads/" + n]).reshape(-1)) < 2e-3, n
        off += k


def test_module_surface_drop_in_loop(golden_dir):
    """The reference's own loop body (ivae_ardae.py:713-846) written against this package's modules + optimisers."""
    mc, cc, nz, full, fx, pm, pc = load_case(golden_dir, "tiny_mnist_grad")
    model, cdae = build(mc, cc)
    model.load_state_dict(pm); cdae.load_state_dict(pc)
    device = torch.device("cuda")
    model, cdae = model.to(device), cdae.to(device)
    model_optimizer = net.Adam(model.parameters(), lr=1e-4, betas=(0.5, 0.999))
    cdae_optimizer = net.RMSprop(cdae.parameters(), lr=1e-4, momentum=0.5)
    std_scale, delta, beta, nz_model = 1e4, 0.1, 1.0, 1
    noise = noise_of(fx, 0, device)
    x = torch.tensor(fx["s0/x_cdae"]).to(device)
    B = x.size(0)
    # ---- update cdae
    model.train(); cdae.train()
    cdae_optimizer.zero_grad()
    context = model.encode(x, std=0).detach()
    latent_mean = model.encode(x, std=0).detach()
    latent = model.forward_hidden(x, nz=nz, noise=noise["sampler"]).detach()
    latent_sub_mean = std_scale * (latent - latent_mean)
    std_qz = torch.std(latent_sub_mean, dim=1, keepdim=True)
    std = delta * torch.mean(std_qz, dim=2, keepdim=True)
    stdmat = std * noise["sigma"]
    _, cdae_loss = cdae(latent_sub_mean, context, std=stdmat, scale=std_scale, eps=noise["eps"])
    cdae_loss.backward()
    assert rel(cdae_loss.item(), fx["s0/cdae_loss"]) < 1e-4
    assert dict(cdae.named_parameters())["neglogprob.fc.bias"].grad is None
    cdae_optimizer.step()
    ref_c = torch.cat([torch.tensor(fx["s0/cdae_params_after/" + n]).reshape(-1) for n, _ in O.cdae_param_spec(cc)])
    before_c = torch.cat([pc[n].reshape(-1) for n, _ in O.cdae_param_spec(cc)])
    assert_update_close(cdae.flat_params().cpu(), before_c, ref_c, "cdae update")
    # ---- update model
    model.train(); cdae.eval()
    model_optimizer.zero_grad()
    xv = torch.tensor(fx["s0/x_vae"]).to(device)
    output, _, latent, model_loss, recon_loss, prior_loss = model(xv, beta=beta, eta=0., lmbd=0., nz=nz_model, noise=noise["vae"])
    model_loss.backward(retain_graph=True)
    context = model.encode(xv, std=0).detach()
    latent_mean = model.encode(xv, std=0).detach()
    latent_sub_mean = std_scale * (latent - latent_mean).detach()
    stdmat = torch.zeros(B, nz_model, 1, device=device).fill_(0)
    grad = cdae.glogprob(latent_sub_mean, context, std=stdmat, scale=std_scale).detach()
    (std_scale * (latent - latent_mean)).backward(beta * grad.detach() / float(B * nz_model))
    assert rel(model_loss.item(), fx["s0/model_loss"]) < 1e-4
    assert rel(recon_loss.item(), fx["s0/recon"]) < 2e-5 and rel(prior_loss.item(), fx["s0/prior"]) < 2e-5
    for n, p in model.named_parameters():
        assert rel_l2(p.grad.cpu(), torch.tensor(fx["s0/model_grads/" + n])) < 2e-3, n
    model_optimizer.step()
    ref_m = torch.cat([torch.tensor(fx["s0/model_params_after/" + n]).reshape(-1) for n, _ in O.model_param_spec(mc)])
    before_m = torch.cat([pm[n].reshape(-1) for n, _ in O.model_param_spec(mc)])
    assert_update_close(model.flat_params().cpu(), before_m, ref_m, "model update")
    # error behaviour mirrors the reference
    with pytest.raises(AssertionError):
        cdae(latent_sub_mean.view(-1, mc.z_dim), context)                      # graddae/mlp.py:402
    with pytest.raises(NotImplementedError):
        model(xv, lmbd=1.0)                                                     # ivae/mnist.py:288-290


def test_conv_vae_phase_grads_vs_oracle(golden_dir):
    """ConvIPVAE (BASELINE config #4 model): every conv / transposed-conv / fc gradient of the VAE phase against the
    oracle (itself pinned to the reference's ConvIPVAE at 6e-7) on the fixture's inputs."""
    mc, cc, nz, full, fx, pm, pc = load_case(golden_dir, "conv_b4_nz8")
    model, cdae = build(mc, cc)
    model.load_state_dict(pm); cdae.load_state_dict(pc)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    B = int(fx["meta_B"])
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=nz), batch_size=B)
    noise = noise_of(fx, 0, "cuda")
    xv = torch.tensor(fx["s0/x_vae"])
    eng.vae_phase(xv.cuda(), noise=noise, apply_update=False)
    mloss, rec, pri, g, gm = O.vae_update_grads(mc, cc, O.TrainCfg(nz_cdae=nz), pm, pc, xv, {k: v.cpu() for k, v in noise.items()})
    s = eng.stats()
    assert rel(s["model_loss"], mloss) < 1e-4 and rel(s["recon"], rec) < 2e-5 and rel(s["prior"], pri) < 2e-5
    off = 0
    for n, shp in O.model_param_spec(mc):
        k = int(np.prod(shp))
        assert rel_l2(eng.grads_m[off:off + k].cpu(), gm[n].reshape(-1)) < 2e-3, n
        off += k
    # decoder-only entry point (IWAE evaluator) agrees with the oracle's decoder
    z = torch.randn(6, mc.z_dim)
    (logit,) = model.decode_params(z.cuda())
    assert rel_l2(logit, O.decode(mc, pm, z)[0]) < 1e-5


@pytest.mark.parametrize("name", ["tiny_mnist_grad", "tiny_toy_grad"])
def test_optimizer_kernels_golden(golden_dir, name):
    """utils.Adam / torch RMSprop arithmetic with the reference's own gradients: 3 chained steps, tight tolerance."""
    mc, cc, nz, full, fx, pm, pc = load_case(golden_dir, name)
    model, cdae = build(mc, cc)
    model.load_state_dict(pm); cdae.load_state_dict(pc)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    m_opt = net.Adam(model.parameters(), lr=1e-4, betas=(0.5, 0.999))
    c_opt = net.RMSprop(cdae.parameters(), lr=1e-4, momentum=0.5)
    for t in range(int(fx["meta_steps"])):
        pre = f"s{t}/"
        for mod, grp, opt, after in ((cdae, "cdae_grads", c_opt, "cdae_params_after"), (model, "model_grads", m_opt, "model_params_after")):
            before = {n: p.detach().clone() for n, p in mod.named_parameters()}
            for n, p in mod.named_parameters():
                key = pre + grp + "/" + n
                p.grad = None if key + "/none" in fx else torch.tensor(fx[key]).cuda()
            opt.step()
            for n, p in mod.named_parameters():
                ref = torch.tensor(fx[pre + after + "/" + n])
                # within 2 ulp of the reference's fp32 result (fma contraction differs between the two code generators)
                assert bool(((p.detach().cpu() - ref).abs() <= 2.4e-7 * ref.abs() + 1e-9).all()), (t, n)
                assert float((p.detach().cpu() - before[n].cpu()).abs().max()) > 0 or key + "/none" in fx
            # the fixture chain continues from the reference's values (already equal up to the tolerance above)
            with torch.no_grad():
                for n, p in mod.named_parameters():
                    p.copy_(torch.tensor(fx[pre + after + "/" + n]).cuda())
    st = m_opt.state_dict()["state"][0]
    assert set(st.keys()) == {"step", "exp_avg", "exp_avg_sq"} and st["step"] == int(fx["meta_steps"])
    stc = c_opt.state_dict()["state"]
    assert set(stc[0].keys()) == {"step", "square_avg", "momentum_buffer"}
    assert (len(stc) == len(list(cdae.parameters())) - 1) == (cc.kind == "grad")      # no state for the grad-less bias


def test_philox_normal_moments():
    net.manual_seed(123)
    a = net.rng.normal((1 << 20,), "cuda")
    b = net.rng.normal((1 << 20,), "cuda")
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    assert abs(float((a * b).mean())) < 5e-3                                     # consecutive draws are independent
    assert abs(float((a ** 4).mean()) - 3) < 0.05
    # (round 4: Box-Muller on the hardware log2 / sin / cos units) the shape of the distribution, not only its first moments
    assert abs(float((a ** 3).mean())) < 0.02 and abs(float((a ** 6).mean()) - 15) < 0.6
    for thr, p in ((1.0, 0.317311), (1.959964, 0.05), (3.0, 0.0026998), (4.0, 6.334e-5)):
        assert abs(float((a.abs() > thr).float().mean()) - p) < 4 * (p / (1 << 20)) ** 0.5 + 1e-6, thr
    assert float(a.abs().max()) < 6.5 and torch.isfinite(a).all()
    c = torch.stack([a[0::2], a[1::2]])                                      # the two outputs of one Box-Muller pair are uncorrelated
    assert abs(float((c[0] * c[1]).mean())) < 5e-3
    net.manual_seed(123)
    assert torch.equal(a, net.rng.normal((1 << 20,), "cuda"))                    # reproducible from (seed, offset)


def test_iwae_logprob_golden(golden_dir):
    """model.logprob (IWAE-k, full-covariance Gaussian proposal) against the reference's value with injected draws."""
    fx = dict(np.load(os.path.join(golden_dir, "iwae_tiny.npz")))
    mc = O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus")
    model, _ = build(mc, O.CdaeCfg("grad", 8, 8, 32, 2))
    model.load_state_dict({n: torch.tensor(fx["pm/" + n]).float() for n, _ in O.model_param_spec(mc)})
    model = model.to("cuda")
    k = int(fx["meta_k"])
    got = model.logprob(torch.tensor(fx["x"]).float().cuda(), sample_size=k, enc_noise=torch.tensor(fx["enc_noise"]).float().cuda(),
                        prop_noise=torch.tensor(fx["prop_noise"]).float().cuda())
    ref = float(fx["logprob"])
    assert abs(float(got) - ref) < 1e-4 * abs(ref)          # fixture is float64; the device path is fp32
    with pytest.raises(AssertionError):
        model.logprob(torch.tensor(fx["x"]).float().cuda(), sample_size=8)      # needs sample_size >= 2 z_dim (ivae/mnist.py:382)
    _, mean, z = model.generate(5)
    assert mean.shape == (5, 24) and z.shape == (5, 8) and bool((mean >= 0).all() and (mean <= 1).all())


def test_engine_checkpoint_roundtrip_reference_format(tmp_path):
    """the fused engine writes / reads the reference's checkpoint dicts ('state_dict' + 'optimizer' in
    torch.optim layout, ivae_ardae.py:931-950).  A resumed engine continues bit-identically (parameters, optimiser state,
    noise stream); the optimiser dicts load into the drop-in optimisers; the files survive torch.load(weights_only=True)."""
    mc = O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus")
    cc = O.CdaeCfg("grad", 8, 8, 64, 3)
    B = 8

    def fresh(seed_params):
        model, cdae = build(mc, cc)
        if seed_params:
            model.load_state_dict(O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc)))
            cdae.load_state_dict(O.init_params(O.cdae_param_spec(cc), 1))
        model, cdae = model.to("cuda"), cdae.to("cuda")
        return model, cdae, net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=16), batch_size=B)

    g = torch.Generator().manual_seed(3)
    xs = [torch.bernoulli(torch.full((B, 24), 0.3), generator=g).cuda() for _ in range(6)]
    net.manual_seed(123)
    model, cdae, eng = fresh(True)
    eng.step(xs[0], xs[1]); eng.step(xs[2], xs[3])
    mck, cck = eng.model_checkpoint(), eng.cdae_checkpoint()
    # same top-level keys as the reference's files (plus the engine's RNG bookkeeping), tensor-only payload
    torch.save(mck, tmp_path / "model.pth.tar"); torch.save(cck, tmp_path / "cdae.pth.tar")
    mck2 = torch.load(tmp_path / "model.pth.tar", weights_only=True); cck2 = torch.load(tmp_path / "cdae.pth.tar", weights_only=True)
    assert list(mck2["state_dict"]) == [n for n, _ in O.model_param_spec(mc)]
    assert set(cck2["optimizer"]["state"]) == set(range(len(O.cdae_param_spec(cc)) - 1))      # no state for neglogprob.fc.bias
    eng.step(xs[4], xs[5])
    want_m, want_c = model.flat_params().clone(), cdae.flat_params().clone()

    model_b, cdae_b, eng_b = fresh(False)
    eng_b.load_checkpoints(mck2, cck2)
    assert eng_b.step_count == 2
    eng_b.step(xs[4], xs[5])
    assert torch.equal(model_b.flat_params(), want_m) and torch.equal(cdae_b.flat_params(), want_c)

    # a file written BEFORE the device step block described the coming step (no 'state_version': the block holds the step just done -
    # t == step_count and that step's Philox offsets): on load the block is advanced once, and the run continues on the same bits
    import copy
    old_layout = copy.deepcopy(mck2)
    del old_layout["engine"]["state_version"]
    st = old_layout["engine"]["step_state"]                      # [rng offset, Adam t, (step size, sqrt bc2) as one packed int64, -]
    st[0] -= net.ArdaeEngine.RNG_STRIDE; st[1] -= 1              # what the older layout stored after two steps
    model_o, cdae_o, eng_o = fresh(False)
    eng_o.load_checkpoints(old_layout, cck2)
    assert eng_o.step_count == 2 and int(eng_o.state[1]) == 3 and int(eng_o.state[0]) == int(mck2["engine"]["step_state"][0])
    eng_o.step(xs[4], xs[5])
    assert torch.equal(model_o.flat_params(), want_m) and torch.equal(cdae_o.flat_params(), want_c)

    # interchangeable with the drop-in optimisers (and hence with the reference's utils.Adam / torch RMSprop layouts)
    model_c, cdae_c = build(mc, cc)
    model_c, cdae_c = model_c.to("cuda"), cdae_c.to("cuda")
    m_opt = net.Adam(model_c.parameters(), lr=1e-4, betas=(0.5, 0.999)); c_opt = net.RMSprop(cdae_c.parameters(), lr=1e-4, momentum=0.5)
    m_opt.load_state_dict(mck2["optimizer"]); c_opt.load_state_dict(cck2["optimizer"])
    st = m_opt.state_dict()["state"]
    assert int(st[0]["step"]) == 2 and torch.equal(st[0]["exp_avg"].cpu(), mck2["optimizer"]["state"][0]["exp_avg"].cpu())
    assert m_opt.state_dict()["param_groups"][0]["betas"] == (0.5, 0.999) or list(m_opt.state_dict()["param_groups"][0]["betas"]) == [0.5, 0.999]


def test_on_device_data_helpers():
    """dynamic binarisation (datasets/mnist.py:36-40) and the 25-Gaussians set (datasets/toy.py:193-227) drawn on
    the device from the library's Philox stream."""
    net.manual_seed(5)
    x, label = net.data.gaussians25(25 * 4000)
    assert x.shape == (100000, 2) and label.shape == (100000,) and x.is_cuda
    lin = torch.linspace(-4, 4, 5)
    for comp in (0, 1, 7, 24):
        pts = x[label == comp].cpu()
        assert pts.shape[0] == 4000
        want = torch.tensor([lin[comp % 5], lin[comp // 5]])               # x varies fastest, as np.meshgrid(x, y) does
        assert (pts.mean(0) - want).abs().max() < 0.03                      # 6 sigma of the sample mean
        assert (pts.var(0) - 0.1).abs().max() < 0.012
    with pytest.raises(ValueError):
        net.data.gaussians25(1001)
    probs = torch.rand(64, 784, device="cuda")
    xb = net.data.dynamic_binarize(probs)
    assert set(torch.unique(xb).tolist()) <= {0.0, 1.0}
    assert abs(float(xb.mean()) - float(probs.mean())) < 0.01
    strong = net.data.dynamic_binarize((probs > 0.5).float())               # probabilities 0 / 1 are reproduced exactly
    assert torch.equal(strong, (probs > 0.5).float())
    assert not torch.equal(net.data.dynamic_binarize(probs), xb)            # a fresh draw every call


@pytest.mark.parametrize("B,nz,z", [(5, 256, 32), (3, 512, 32), (4, 64, 32), (3, 70, 5), (2, 1024, 32), (6, 256, 2), (3, 625, 32), (2, 100, 16)])
def test_latent_perturb_kernels(B, nz, z):
    """Latent statistics + perturbation (ivae_ardae.py:753-767, graddae/mlp.py:21-23) against the oracle's restatement: the
    register-resident kernel (z a power of two, up to 96 values per thread, ragged last pass masked) and the generic one."""
    import ctypes
    from ardae_amd import _lib as L
    g = torch.Generator().manual_seed(B * 1000 + nz + z)
    z0 = torch.randn(B, z, generator=g)
    latent = z0[:, None, :] + 0.05 * torch.randn(B, nz, z, generator=g)
    xi, eps = torch.randn(B, nz, 1, generator=g), torch.randn(B * nz, z, generator=g)
    u, std = O.latent_stats(latent, z0.view(B, 1, z), 1e4, 0.1)
    sigma_ref = (std * xi).reshape(-1)
    xbar_ref = u.reshape(-1, z) + sigma_ref[:, None] * eps
    d = lambda t: t.contiguous().cuda()
    lat_d, z0_d, xi_d, eps_d = d(latent), d(z0), d(xi.reshape(-1)), d(eps)
    xbar, sigma, std_b = torch.empty(B * nz, z, device="cuda"), torch.empty(B * nz, device="cuda"), torch.empty(B, device="cuda")
    L.check(L.lib().ardae_latent_perturb(L.ptr(lat_d), L.ptr(z0_d), L.ptr(xi_d), L.ptr(eps_d), B, nz, z, 1e4, 0.1, L.ptr(xbar), L.ptr(sigma),
                                         L.ptr(std_b), L.stream_ptr()), "ardae_latent_perturb")
    torch.cuda.synchronize()
    assert rel_l2(std_b, std.reshape(-1)) < 1e-5
    assert rel_l2(sigma, sigma_ref) < 1e-5
    assert rel_l2(xbar, xbar_ref) < 1e-5


@pytest.mark.parametrize("B,nz,z,rank", [(5, 256, 32, 0), (8, 256, 32, 3), (4, 64, 32, 1), (6, 512, 16, 2), (3, 128, 8, 0), (64, 256, 32, 7)])
def test_latent_perturb_with_in_kernel_draws_equals_separate_draws(B, nz, z, rank):
    """North star: 'fused per-sample Gaussian-perturb + sigma-scaling' with the draws made in the kernel (ivae_ardae.py:761,
    graddae/mlp.py:21-23).  ardae_latent_perturb_draw generates the Philox counters of xi and eps itself; for the same (seed,
    offset, element) keying - including a rank's row offset into the global draws and the step state's base offset - it must
    give BIT-IDENTICAL xbar / sigma / std_b / eps to two ardae_philox_normal_at launches followed by ardae_latent_perturb."""
    import ctypes
    from ardae_amd import _lib as L
    lib = L.lib()
    assert lib.ardae_latent_perturb_draw_ok(nz, 1, z) == 1
    g = torch.Generator().manual_seed(B + nz + z)
    z0 = torch.randn(B, z, generator=g).cuda()
    latent = (z0[:, None, :].cpu() + 0.05 * torch.randn(B, nz, z, generator=g)).cuda().contiguous()
    seed, k_xi, k_eps = 0xC0FFEE, 4, 5
    state = torch.zeros(4, dtype=torch.int64, device="cuda")
    L.check(lib.ardae_step_state_advance(ctypes.c_void_p(state.data_ptr()), ctypes.c_uint64(48), 1e-4, 0.5, 0.999, L.stream_ptr()))
    first = rank * B * nz
    xi, eps = torch.empty(B * nz, device="cuda"), torch.empty(B * nz, z, device="cuda")
    for t, k, f in ((xi, k_xi, first), (eps, k_eps, first * z)):
        L.check(lib.ardae_philox_normal_at(L.ptr(t), t.numel(), ctypes.c_uint64(seed), ctypes.c_uint64(k), ctypes.c_void_p(state.data_ptr()),
                                           ctypes.c_uint64(f), L.stream_ptr()))
    new = lambda *s: torch.full(s, float("nan"), device="cuda")
    xbar, sigma, std_b = new(B * nz, z), new(B * nz), new(B)
    L.check(lib.ardae_latent_perturb(L.ptr(latent), L.ptr(z0), L.ptr(xi), L.ptr(eps), B, nz, z, 1e4, 0.1, L.ptr(xbar), L.ptr(sigma), L.ptr(std_b),
                                     L.stream_ptr()))
    xbar2, sigma2, std_b2, eps2 = new(B * nz, z), new(B * nz), new(B), new(B * nz, z)
    L.check(lib.ardae_latent_perturb_draw(L.ptr(latent), L.ptr(z0), B, nz, z, 1e4, 0.1, ctypes.c_uint64(seed), ctypes.c_uint64(k_xi), ctypes.c_uint64(k_eps),
                                          ctypes.c_void_p(state.data_ptr()), ctypes.c_uint64(first), L.ptr(xbar2), L.ptr(sigma2), L.ptr(eps2), L.ptr(std_b2),
                                          L.stream_ptr()), "ardae_latent_perturb_draw")
    torch.cuda.synchronize()
    assert torch.equal(eps2, eps) and torch.equal(std_b2, std_b) and torch.equal(sigma2, sigma) and torch.equal(xbar2, xbar)
    # moments of the in-kernel draw (B nz z normals)
    if eps2.numel() >= 100_000:
        assert abs(float(eps2.mean())) < 4.0 / eps2.numel() ** 0.5 and abs(float(eps2.var()) - 1.0) < 0.02
    assert lib.ardae_latent_perturb_draw_ok(625, 1, 32) == 0 and lib.ardae_latent_perturb_draw_ok(256, 3, 32) == 0     # ragged counters / nstd: separate draws


@pytest.mark.parametrize("graph", [False, True])
def test_engine_two_cdae_updates_per_step(graph):
    """--num-cdae-updates 2 (the shipped mnist-conv / auxresconvct recipes, run_vae_dbmnist.sh): two cDAE updates on their own
    batches, then the VAE update against the twice-updated cDAE (ivae_ardae.py:713-846), against the oracle run the same way.
    graph=False injects the oracle's draws (tight comparison); graph=True checks that the captured two-update step replays
    with fresh batches and noise (own Philox stream, so only finiteness / movement / bit-identity of two engines is checked)."""
    mc = O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus")
    cc = O.CdaeCfg("grad", 8, 8, 64, 3)
    tc = O.TrainCfg(nz_cdae=8)
    B = 4
    pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc))
    pc = O.init_params(O.cdae_param_spec(cc), 1)
    gen = torch.Generator().manual_seed(21)
    xs = [torch.bernoulli(torch.full((B, 24), 0.3), generator=gen) for _ in range(3)]
    noises = [O.draw_step_noise(mc, tc, B, gen) for _ in range(2)]

    def make():
        model, cdae = build(mc, cc)
        model.load_state_dict(pm); cdae.load_state_dict(pc)
        model, cdae = model.to("cuda"), cdae.to("cuda")
        return model, cdae, net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=8, num_cdae_updates=2), batch_size=B, graph=graph)

    if not graph:
        model, cdae, eng = make()
        dn = [{k: v.cuda().contiguous() for k, v in n.items()} for n in noises]
        eng.step([xs[0].cuda(), xs[1].cuda()], xs[2].cuda(), noise=dn)
        got = eng.stats()
        rpm, rpc = {k: v.clone() for k, v in pm.items()}, {k: v.clone() for k, v in pc.items()}
        st_m, st_c = {}, {}
        for i in range(2):
            closs, gc, _ = O.cdae_update_grads(mc, cc, tc, rpm, rpc, xs[i], noises[i])
            with torch.no_grad():
                O.rmsprop_step(rpc, gc, st_c, tc.d_lr, tc.d_momentum)
        mloss, rec, pri, _, gm = O.vae_update_grads(mc, cc, tc, rpm, rpc, xs[2], noises[1])
        with torch.no_grad():
            O.adam_ref_step(rpm, gm, st_m, tc.m_lr, tc.m_beta1)
        assert rel(got["cdae_loss"], closs) < 1e-4 and rel(got["model_loss"], mloss) < 1e-4
        assert rel(got["recon"], rec) < 2e-5 and rel(got["prior"], pri) < 2e-5
        flat = lambda p, spec: torch.cat([p[n].reshape(-1) for n, _ in spec])

        def updates_agree(after, before, ref_after, what):
            # sign-like first optimiser steps (|update| ~ lr whatever |grad| is): elements whose gradient sits at the fp32 noise
            # floor may flip, and the second cDAE update / the VAE update inherit the first one's flips - so: the typical element
            # agrees tightly and only a small fraction disagrees at all (cf. assert_update_close, whose L2 bound is for one update)
            upd, ref = (after - before).double(), (ref_after - before).double()
            err = (upd - ref).abs() / (ref.abs() + 1e-12)
            assert float(err.median()) < 1e-3, what
            assert float((err > 1e-2).double().mean()) < 2e-2, what
        updates_agree(cdae.flat_params().cpu()[:-1], flat(pc, O.cdae_param_spec(cc))[:-1], flat(rpc, O.cdae_param_spec(cc))[:-1], "cdae after two updates")
        updates_agree(model.flat_params().cpu(), flat(pm, O.model_param_spec(mc)), flat(rpm, O.model_param_spec(mc)), "model")
        return
    outs = []
    for _ in range(2):
        net.manual_seed(5)
        model, cdae, eng = make()
        for t in range(4):                       # eager warm step, capture, two replays - with changing batches
            eng.step([xs[t % 3].cuda(), xs[(t + 1) % 3].cuda()], xs[(t + 2) % 3].cuda())
        torch.cuda.synchronize()
        assert eng._graph is not None and len(eng._xc) == 2
        st = eng.stats()
        assert all(v == v for v in st.values())
        outs.append((model.flat_params().clone(), cdae.flat_params().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][1].cpu(), torch.cat([pc[n].reshape(-1) for n, _ in O.cdae_param_spec(cc)]))


def test_engine_clipped_aux_resconv_own_draws_replay_and_refusals():
    """--model auxresconv-clip / auxresconvct-clip with the engine's OWN noise: every phase opens with two std = 0 calls that are random
    draws for this class (z0 = mu0 + eps0, ivae/auxresconv2.py:91) - four extra Philox draws per step (offsets 9 .. 12), made inside the
    captured graphs.  Two engines from one seed end bit-identical after eager + captured + replayed steps, replay == eager, the draws move
    from step to step; and what the class is not built for is refused by name."""
    mc = O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu", clipped=True)
    cc = O.CdaeCfg("res", 32, 450, 64, 2)
    pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc)); pc = O.init_params(O.cdae_param_spec(cc), 1)
    g = torch.Generator().manual_seed(9)
    xs = [torch.bernoulli(torch.full((4, 784), 0.2), generator=g) for _ in range(3)]

    def run(graph):
        net.manual_seed(77)
        model, cdae = build(mc, cc)
        model.load_state_dict(pm); cdae.load_state_dict(pc)
        model, cdae = model.to("cuda"), cdae.to("cuda")
        eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=8, cdae_ctx_type="hidden1a", **RES_RECIPE), batch_size=4, graph=graph)
        raws = []
        for t in range(4):
            eng.step(xs[t % 3].cuda(), xs[(t + 1) % 3].cuda())
            raws.append(torch.cat([eng.raw_c.flatten(), eng.raw_v.flatten()]).clone())
        torch.cuda.synchronize()
        assert all(v == v for v in eng.stats().values())
        return model.flat_params().clone(), cdae.flat_params().clone(), raws
    a, b, e = run(True), run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])                      # deterministic
    assert torch.equal(a[0], e[0]) and torch.equal(a[1], e[1])                      # replay == eager
    assert not torch.equal(a[2][2], a[2][3]) and float(a[2][3].abs().max()) > 0     # fresh draws in every replay
    assert abs(float(a[2][3].mean())) < 0.2 and abs(float(a[2][3].std()) - 1) < 0.2
    model, cdae = build(mc, O.CdaeCfg("res", 32, 32, 64, 2))
    with pytest.raises(NotImplementedError, match="hidden1a"):
        net.ArdaeEngine(model.to("cuda"), cdae.to("cuda"), net.TrainConfig(nz_cdae=8, cdae_ctx_type="lt0"), batch_size=4)
    with pytest.raises(NotImplementedError, match="std must be"):
        model.encode(xs[0].cuda(), std=0.5)


def test_engine_toy_aux_own_draws_replay():
    """--model auxmlp with the engine's own noise under graph replay (nz_cdae 16 = 4 z0's x 4 z's per image; the sampler's draw is one flat
    [eps0 block | eps block] buffer): two engines from one seed end bit-identical, replay == eager, a non-square nz_cdae is refused."""
    mc = O.ModelCfg("auxtoy", 2, 2, 32, 2, 2, "tanh")
    cc = O.CdaeCfg("grad", 2, 64, 64, 3)
    pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc)); pc = O.init_params(O.cdae_param_spec(cc), 1)
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(8, 2, generator=g) for _ in range(3)]

    def run(graph, nz=16):
        net.manual_seed(5)
        model, cdae = build(mc, cc)
        model.load_state_dict(pm); cdae.load_state_dict(pc)
        model, cdae = model.to("cuda"), cdae.to("cuda")
        eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=nz, cdae_ctx_type="hidden1a"), batch_size=8, graph=graph)
        for t in range(4):
            eng.step(xs[t % 3].cuda(), xs[(t + 1) % 3].cuda())
        torch.cuda.synchronize()
        assert all(v == v for v in eng.stats().values())
        return model.flat_params().clone(), cdae.flat_params().clone()
    a, b, e = run(True), run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[0], e[0]) and torch.equal(a[1], e[1])
    assert not torch.equal(a[0].cpu(), torch.cat([pm[n].reshape(-1) for n, _ in O.model_param_spec(mc)]))
    with pytest.raises(ValueError, match="square"):
        run(False, nz=12)


def test_engine_beta_annealing_under_graph_mode():
    """--beta-annealing (utils/msc.py:53-55, ivae_ardae.py:800): beta is a kernel argument frozen in a captured graph, so the
    engine launches eagerly while beta moves and captures once it has settled - with the same parameters, bit for bit, as an
    engine that never uses graphs."""
    mc = O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus")
    cc = O.CdaeCfg("grad", 8, 8, 64, 3)
    pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc))
    pc = O.init_params(O.cdae_param_spec(cc), 1)
    gen = torch.Generator().manual_seed(3)
    xs = [torch.bernoulli(torch.full((4, 24), 0.3), generator=gen).cuda() for _ in range(8)]
    betas = [net.annealing_func(0.1, 1.0, 3, t) for t in range(8)]          # 0.1, 0.4, 0.7, 1.0, 1.0, ...
    assert betas[0] == pytest.approx(0.1) and betas[3] == betas[7] == pytest.approx(1.0)
    outs = []
    for graph in (True, False):
        net.manual_seed(17)
        model, cdae = build(mc, cc)
        model.load_state_dict(pm); cdae.load_state_dict(pc)
        model, cdae = model.to("cuda"), cdae.to("cuda")
        eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=8), batch_size=4, graph=graph)
        captured_at = None
        for t in range(8):
            eng.step(xs[t], xs[(t + 3) % 8], beta=betas[t])
            if captured_at is None and eng._graph is not None:
                captured_at = t
        torch.cuda.synchronize()
        if graph:
            assert captured_at is not None and captured_at >= 5 and betas[captured_at] == betas[7]     # never while beta was moving
        outs.append((model.flat_params().clone(), cdae.flat_params().clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_aux_model_module_surface(golden_dir):
    """MNISTAuxIPVAE through the reference's call surface (ivae_ardae.py:737-749,801-834): the hidden1a context, the latent mean,
    the sampler with a pair of draws, and the loss / gradients of model(...) + latent.backward(seed) against the oracle."""
    mc, cc, nz, full, fx, pm, pc = load_case(golden_dir, "tiny_auxmnist_grad")
    model, _ = build(mc, cc)
    model.load_state_dict(pm)
    model = model.to("cuda")
    x = torch.tensor(fx["s0/x_cdae"])
    B = x.size(0)
    hid = model.encode.forward_hidden(x.cuda(), std=0)
    ref_hid = O.cdae_context(mc, O.TrainCfg(ctx_type="hidden1a"), pm, x)
    assert hid.shape == (B, 2 * mc.h_dim) and rel_l2(hid, ref_hid) < 1e-5
    z0 = model.encode(x.cuda(), std=0)
    assert z0.shape == (B, 1, mc.z_dim) and rel_l2(z0.reshape(B, -1), fx["s0/z0"].reshape(B, -1)) < 1e-5
    e0, e = torch.tensor(fx["s0/noise/sampler"]), torch.tensor(fx["s0/noise/sampler_z"])
    z = model.forward_hidden(x.cuda(), nz=nz, noise=(e0.cuda(), e.cuda()))
    assert z.shape == (B, nz, mc.z_dim) and rel_l2(z.reshape(B * nz, -1), O.encode(mc, pm, x, (e0, e), nz).reshape(B * nz, -1)) < 1e-5
    # loss + both backward calls of the VAE phase, with an arbitrary seed
    xv = torch.tensor(fx["s0/x_vae"])
    nv = (torch.tensor(fx["s0/noise/vae"]), torch.tensor(fx["s0/noise/vae_z"]))
    seed = torch.tensor(fx["s0/score"]).reshape(B, 1, mc.z_dim) * 0.01
    _, _, latent, loss, rec, pri = model(xv.cuda(), beta=1.0, eta=0., lmbd=0., nz=1, noise=(nv[0].cuda(), nv[1].cuda()))
    loss.backward(retain_graph=True)
    latent.backward(seed.cuda())
    preq = {k: v.clone().requires_grad_(True) for k, v in pm.items()}
    zr, lr_, recr, prir, _ = O.vae_forward(mc, preq, xv, nv, 1.0, 1)
    (lr_ + (zr * seed).sum()).backward()
    assert rel(loss.item(), lr_.item()) < 1e-4 and rel(rec.item(), recr.item()) < 2e-5 and rel(pri.item(), prir.item()) < 2e-5
    for n, p in model.named_parameters():
        assert rel_l2(p.grad.cpu(), preq[n].grad) < 2e-4, n
    net.MNISTAuxIPVAE(clip_z_logvar="spm4")                     # every choice of NormalDistribution.clip_logvar constructs (round 4) ...
    with pytest.raises(NotImplementedError):
        net.MNISTAuxIPVAE(clip_z_logvar="spm7")                 # ... names it does not know are refused (the reference would silently not clip)


def test_toy_aux_model_module_surface_and_iwae(golden_dir):
    """ToyAuxIPVAE (`--model auxmlp`) through the reference's call surface: the hidden1a context, the latent mean, the SQUARE sampler
    (nz = 16 rows per image = 4 z0's x 4 z's; a non-square nz is refused), loss / gradients of model(...) + latent.backward(seed) against
    the oracle (pinned to the reference's class at 1e-15 in float64), and logprob (k x k encoder samples fit the proposal, ivae/auxtoy.py:313)
    against the reference's value with injected draws."""
    mc, cc, nz, full, fx, pm, pc = load_case(golden_dir, "tiny_auxtoy_grad")
    model, _ = build(mc, cc)
    model.load_state_dict(pm)
    model = model.to("cuda")
    x = torch.tensor(fx["s0/x_cdae"])
    B = x.size(0)
    hid = model.encode.forward_hidden(x.cuda(), std=0)
    assert hid.shape == (B, 2 * mc.h_dim) and rel_l2(hid, O.cdae_context(mc, O.TrainCfg(ctx_type="hidden1a"), pm, x)) < 1e-5
    z0 = model.encode(x.cuda(), std=0)
    assert z0.shape == (B, 1, mc.z_dim) and rel_l2(z0.reshape(B, -1), fx["s0/z0"].reshape(B, -1)) < 1e-5
    e0, e = torch.tensor(fx["s0/noise/sampler"]), torch.tensor(fx["s0/noise/sampler_z"])
    assert e0.shape == (B * 4, mc.noise_dim) and e.shape == (B * 16, mc.z_dim)
    z = model.forward_hidden(x.cuda(), nz=nz, noise=(e0.cuda(), e.cuda()))
    assert z.shape == (B, nz, mc.z_dim) and rel_l2(z.reshape(B * nz, -1), fx["s0/latent"].reshape(B * nz, -1)) < 1e-5
    with pytest.raises(ValueError, match="not a square"):
        model.forward_hidden(x.cuda(), nz=12)
    zz = model.forward_hidden(x.cuda(), nz=9)          # own draws
    assert zz.shape == (B, 9, mc.z_dim) and torch.isfinite(zz).all()
    xv = torch.tensor(fx["s0/x_vae"])
    nv = (torch.tensor(fx["s0/noise/vae"]), torch.tensor(fx["s0/noise/vae_z"]))
    seed = torch.tensor(fx["s0/score"]).reshape(B, 1, mc.z_dim) * 0.01
    _, _, latent, loss, rec, pri = model(xv.cuda(), beta=1.0, eta=0., lmbd=0., nz=1, noise=(nv[0].cuda(), nv[1].cuda()))
    loss.backward(retain_graph=True)
    latent.backward(seed.cuda())
    preq = {k: v.clone().requires_grad_(True) for k, v in pm.items()}
    zr, lr_, recr, prir, _ = O.vae_forward(mc, preq, xv, nv, 1.0, 1)
    (lr_ + (zr * seed).sum()).backward()
    assert rel(loss.item(), lr_.item()) < 1e-4 and rel(rec.item(), recr.item()) < 2e-5 and rel(pri.item(), prir.item()) < 2e-5
    for n, p in model.named_parameters():
        assert rel_l2(p.grad.cpu(), preq[n].grad) < 2e-4, n
    # nz = 4 rows per image through the VAE forward / backward (2 z0's x 2 z's: the sums over the q z's of a stage row)
    g4 = torch.Generator().manual_seed(4)
    n4 = (torch.randn(B * 2, mc.noise_dim, generator=g4), torch.randn(B * 4, mc.z_dim, generator=g4))
    model.zero_grad()
    _, _, lat4, loss4, _, _ = model(xv.cuda(), beta=0.7, eta=0., lmbd=0., nz=4, noise=(n4[0].cuda(), n4[1].cuda()))
    loss4.backward()
    preq = {k: v.clone().requires_grad_(True) for k, v in pm.items()}
    _, l4, _, _, _ = O.vae_forward(mc, preq, xv, n4, 0.7, 4)
    l4.backward()
    assert rel(loss4.item(), l4.item()) < 1e-4
    for n, p in model.named_parameters():
        assert rel_l2(p.grad.cpu(), preq[n].grad) < 2e-4, n
    # IWAE
    fi = dict(np.load(os.path.join(golden_dir, "iwae_tiny_auxtoy.npz")))
    model.load_state_dict({n: torch.tensor(fi["pm/" + n]).float() for n, _ in O.model_param_spec(mc)})
    k = int(fi["meta_k"])
    got = model.logprob(torch.tensor(fi["x"]).float().cuda(), sample_size=k,
                        enc_noise=(torch.tensor(fi["enc_noise"]).float().cuda(), torch.tensor(fi["enc_noise_z"]).float().cuda()),
                        prop_noise=torch.tensor(fi["prop_noise"]).float().cuda())
    assert abs(float(got) - float(fi["logprob"])) < 1e-4 * abs(float(fi["logprob"]))
    _, mean, zg = model.generate(5)
    assert mean.shape == (5, 2) and zg.shape == (5, 2)


def test_iwae_logprob_golden_aux(golden_dir):
    """MNISTAuxIPVAE.logprob (ivae/auxmnist.py:300-356: the same IWAE-k bound, proposal covariance + 1e-5 I) against the reference's
    value with injected draws."""
    fx = dict(np.load(os.path.join(golden_dir, "iwae_tiny_auxmnist.npz")))
    mc = O.ModelCfg("auxmnist", 24, 10, 48, 8, 2, "softplus")
    model, _ = build(mc, O.CdaeCfg("grad", 8, 96, 32, 2))
    model.load_state_dict({n: torch.tensor(fx["pm/" + n]).float() for n, _ in O.model_param_spec(mc)})
    model = model.to("cuda")
    k = int(fx["meta_k"])
    got = model.logprob(torch.tensor(fx["x"]).float().cuda(), sample_size=k,
                        enc_noise=(torch.tensor(fx["enc_noise"]).float().cuda(), torch.tensor(fx["enc_noise_z"]).float().cuda()),
                        prop_noise=torch.tensor(fx["prop_noise"]).float().cuda())
    ref = float(fx["logprob"])
    assert abs(float(got) - ref) < 1e-4 * abs(ref)          # fixture is float64; the device path is fp32
    _, mean, z = model.generate(5)
    assert mean.shape == (5, 24) and z.shape == (5, 8)


def test_auxconv_vae_phase_grads_vs_oracle(golden_dir):
    """MNISTConvAuxIPVAE (the shipped "hierarchical conv" recipe's model): every conv / fc / reparameterisation-head / transposed-conv
    gradient of the VAE phase, the hidden1a context and the sampler against the oracle (itself pinned to the reference's
    MNISTConvAuxIPVAE at 2e-6) on the fixture's inputs."""
    mc, cc, nz, full, fx, pm, pc = load_case(golden_dir, "auxconv_b4_nz8")
    model, cdae = build(mc, cc)
    assert [k for k in model.state_dict()] == [n for n, _ in O.model_param_spec(mc)]
    model.load_state_dict(pm); cdae.load_state_dict(pc)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    B = int(fx["meta_B"])
    tc = O.TrainCfg(nz_cdae=nz, ctx_type="hidden1a")
    eng = net.ArdaeEngine(model, cdae, train_config(mc, nz), batch_size=B)
    noise = noise_of(fx, 0, "cuda")
    xv = torch.tensor(fx["s0/x_vae"])
    hid = model.encode.forward_hidden(xv.cuda(), std=0)
    assert hid.shape == (B, 1600) and rel_l2(hid, O.cdae_context(mc, tc, pm, xv)) < 1e-5
    eng.vae_phase(xv.cuda(), noise=noise, apply_update=False)
    cpu_noise = {k[len("s0/noise/"):]: torch.tensor(v) for k, v in fx.items() if k.startswith("s0/noise/")}
    mloss, rec, pri, g, gm = O.vae_update_grads(mc, cc, tc, pm, pc, xv, cpu_noise)
    s = eng.stats()
    assert rel(s["model_loss"], mloss) < 1e-4 and rel(s["recon"], rec) < 2e-5 and rel(s["prior"], pri) < 2e-5
    off = 0
    for n, shp in O.model_param_spec(mc):
        k = int(np.prod(shp))
        assert rel_l2(eng.grads_m[off:off + k].cpu(), gm[n].reshape(-1)) < 2e-3, n
        off += k
    z = torch.randn(6, mc.z_dim)
    (logit,) = model.decode_params(z.cuda())
    assert rel_l2(logit, O.decode(mc, pm, z)[0]) < 1e-5


def test_iwae_logprob_golden_auxconv(golden_dir):
    """MNISTConvAuxIPVAE.logprob (ivae/auxconv.py:289-350) against the reference's value with injected draws."""
    fx = dict(np.load(os.path.join(golden_dir, "iwae_auxconv.npz")))
    mc = O.ModelCfg("auxconv", 784, 100, 800, 32, 1, "softplus")
    pm = O.init_params(O.model_param_spec(mc), int(fx["meta_pseed"]), O.model_init_special(mc))
    model, _ = build(mc, O.CdaeCfg("grad", 32, 1600, 32, 2))
    model.load_state_dict(pm)
    model = model.to("cuda")
    k = int(fx["meta_k"])
    got = model.logprob(torch.tensor(fx["x"]).float().cuda(), sample_size=k,
                        enc_noise=(torch.tensor(fx["enc_noise"]).float().cuda(), torch.tensor(fx["enc_noise_z"]).float().cuda()),
                        prop_noise=torch.tensor(fx["prop_noise"]).float().cuda())
    ref = float(fx["logprob"])
    assert abs(float(got) - ref) < 1e-4 * abs(ref)          # fixture is float64; the device path is fp32


@pytest.mark.parametrize("name", ["resconv_b4_nz8", "auxresconv_b4_nz8", "auxresconv_clip_b4_nz8"] + list(RESCONV_HEAD_CASES))
def test_resconv_vae_phase_grads_vs_oracle(golden_dir, name):
    """The weight-normalised residual-conv families: EVERY gradient tensor of the VAE phase - direction, scale and bias
    of all 45 weight-normalised operators (conv trunk, ResLinear / ResMLP sampler head or the two Gaussian heads with 'spm4' clipping,
    ResLinear + upsampling ResConv decoder) - against the oracle (pinned to the reference's own classes at 4e-14 / 2e-13 in float64),
    plus the cDAE context (lt0 / hidden1a) and the decoder-only entry point."""
    mc, cc, nz, full, fx, pm, pc = load_case(golden_dir, name)
    model, cdae = build(mc, cc)
    assert [k for k in model.state_dict()] == [n for n, _ in O.model_param_spec(mc)]
    model.load_state_dict(pm); cdae.load_state_dict(pc)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    B = int(fx["meta_B"])
    tcfg = train_config(mc, nz)
    tc = O.TrainCfg(nz_cdae=nz, ctx_type=tcfg.cdae_ctx_type, **RES_RECIPE)
    eng = net.ArdaeEngine(model, cdae, tcfg, batch_size=B)
    noise = noise_of(fx, 0, "cuda")
    xv = torch.tensor(fx["s0/x_vae"])
    if mc.kind == "auxresconv" and mc.clipped:      # the std = 0 calls are random draws: inject the fixture's unscaled eps0
        rc, rz = torch.tensor(fx["s0/noise/vctx_raw"]), torch.tensor(fx["s0/noise/vz0_raw"])
        hid = model._hidden(xv.cuda(), raw0=rc.cuda())
        assert hid.shape == (B, 450) and rel_l2(hid, O.cdae_context(mc, tc, pm, xv, rc)) < 1e-5
        z0 = model.forward_hidden(xv.cuda(), std=0, nz=1, noise=rz.cuda())
        assert rel_l2(z0.reshape(B, -1), O.encode(mc, pm, xv, O.zero_noise(mc, B, xv, rz), 1).reshape(B, -1)) < 1e-5
        d1, d2 = model.encode(xv.cuda(), std=0), model.encode(xv.cuda(), std=0)      # fresh draws: two calls differ (z0 = mu0 + eps0)
        assert not torch.equal(d1, d2) and torch.isfinite(d1).all()
    else:
        if mc.kind == "auxresconv":
            hid = model.encode.forward_hidden(xv.cuda(), std=0)
            assert hid.shape == (B, 450) and rel_l2(hid, O.cdae_context(mc, tc, pm, xv)) < 1e-5
        z0 = model.encode(xv.cuda(), std=0)
        assert rel_l2(z0.reshape(B, -1), O.encode(mc, pm, xv, O.zero_noise(mc, B, xv), 1).reshape(B, -1)) < 1e-5
    eng.vae_phase(xv.cuda(), noise=noise, apply_update=False)
    cpu_noise = {k[len("s0/noise/"):]: torch.tensor(v) for k, v in fx.items() if k.startswith("s0/noise/")}
    mloss, rec, pri, g, gm = O.vae_update_grads(mc, cc, tc, pm, pc, xv, cpu_noise)
    s = eng.stats()
    assert rel(s["model_loss"], mloss) < 1e-4 and rel(s["recon"], rec) < 2e-5 and rel(s["prior"], pri) < 2e-5
    off = 0
    for n, shp in O.model_param_spec(mc):
        k = int(np.prod(shp))
        assert rel_l2(eng.grads_m[off:off + k].cpu(), gm[n].reshape(-1)) < 2e-3, n
        off += k
    z = torch.randn(6, mc.z_dim)
    (logit,) = model.decode_params(z.cuda())
    assert rel_l2(logit, O.decode(mc, pm, z)[0]) < 1e-5
    # more Monte-Carlo samples than the fixture has: the sampler on B * 40 rows against the oracle
    gsm = torch.Generator().manual_seed(3)
    wide = torch.randn(B * 40, model._noise_width, generator=gsm)
    zs = model.forward_hidden(xv.cuda(), nz=40, noise=wide.cuda())
    nref = (wide[:, :mc.noise_dim], wide[:, mc.noise_dim:]) if mc.kind == "auxresconv" else wide
    assert rel_l2(zs.reshape(B * 40, -1), O.encode(mc, pm, xv, nref, 40).reshape(B * 40, -1)) < 1e-5


@pytest.mark.parametrize("name,kind,h", [("iwae_resconv", "resconv", 512), ("iwae_auxresconv", "auxresconv", 450), ("iwae_resconv_mlp", "resconv", 512),
                                         ("iwae_auxresconv_clip", "auxresconv", 450)])
def test_iwae_logprob_golden_resconv(golden_dir, name, kind, h):
    """logprob (ivae/resconv.py:325-380, ivae/auxresconv.py:275-345) of the residual-conv models against the reference's value with injected draws."""
    fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    mc = O.ModelCfg(kind, 784, 100, h, 32, 1, "elu", enc_type="mlp" if name.endswith("_mlp") else "res-wn-mlp", clipped=name.endswith("_clip"))
    pm = O.init_params(O.model_param_spec(mc), int(fx["meta_pseed"]), O.model_init_special(mc))
    model, _ = build(mc, O.CdaeCfg("res", 32, 450 if kind == "auxresconv" else 32, 32, 2))
    model.load_state_dict(pm)
    model = model.to("cuda")
    k = int(fx["meta_k"])
    enc = torch.tensor(fx["enc_noise"]).float().cuda()
    if "enc_noise_z" in fx:
        enc = (enc, torch.tensor(fx["enc_noise_z"]).float().cuda())
    got = model.logprob(torch.tensor(fx["x"]).float().cuda(), sample_size=k, enc_noise=enc, prop_noise=torch.tensor(fx["prop_noise"]).float().cuda())
    ref = float(fx["logprob"])
    assert abs(float(got) - ref) < 1e-4 * abs(ref)          # fixture is float64; the device path is fp32


@pytest.mark.parametrize("B,nz,nonlin", [(128, 256, "softplus"), (64, 625, "relu"), (256, 128, "softplus"),
                                         (64, 256, "softplus"), (32, 256, "relu")])      # few rows: hidden columns split over 2 / 4 waves
def test_nrow_sampler_fused_tail_vs_oracle(B, nz, nonlin):
    """forward_hidden(x, nz) on >= 32768 rows: the mnist-concat sampler's two N-row layers (noise -> h -> z) run as ONE launch that keeps
    the hidden rows on chip (linear_shortk.hip::sampler_tail_kernel); against the float64 oracle at config #2's widths, per-image row
    bias groups of nz rows (625: not a multiple of the 32-row wave tile)."""
    mc = O.ModelCfg("mnist", 784, 100, 256, 32, 2, nonlin)
    torch.manual_seed(3)
    pm = O.init_params(O.model_param_spec(mc), 7, O.model_init_special(mc))
    model, _ = build(mc, O.CdaeCfg("grad", 32, 32, 64, 2))
    model.load_state_dict(pm)
    model = model.to("cuda")
    x = (torch.rand(B, 784) < 0.2).float()
    noise = torch.randn(B * nz, mc.noise_dim)
    z = model.forward_hidden(x.cuda(), nz=nz, noise=noise.cuda())
    pm64 = {k: v.double() for k, v in pm.items()}
    ref = O.encode(mc, pm64, x.double(), noise.double(), nz)
    assert z.shape == (B, nz, mc.z_dim)
    assert rel_l2(z.reshape(B * nz, -1), ref.reshape(B * nz, -1).float()) < 2e-5
    assert float((z.reshape(B * nz, -1).cpu() - ref.reshape(B * nz, -1).float()).abs().max()) < 5e-4


@pytest.mark.parametrize("nonlin", ["softplus", "relu"])
def test_sampler_bits_do_not_depend_on_images_per_launch(nonlin):
    """"results independent of R".  A shard of the image batch must give each of its images the bits the whole batch gives
    it: the per-image trunk (784 -> h ..., linear_small.hip: 16 x 16 blocks for few rows, 32 x 32 otherwise - one k order since round 4)
    and the fused N-row sampler tail (its latent-space sum over the hidden columns is split over 1 / 2 / 4 waves depending on the row
    count - four column groups in one order since round 4).  512 images x 256 samples against its shards of 256 / 128 / 64 / 32 images
    (the 2 / 4 / 8 / 16-rank shards of config #2), for z = forward_hidden(x, nz) and z0 = encode(x, std=0)."""
    mc = O.ModelCfg("mnist", 784, 100, 256, 32, 2, nonlin)
    pm = O.init_params(O.model_param_spec(mc), 7, O.model_init_special(mc))
    model, _ = build(mc, O.CdaeCfg("grad", 32, 32, 64, 2))
    model.load_state_dict(pm)
    model = model.to("cuda")
    g = torch.Generator().manual_seed(11)
    B, nz = 512, 256
    x = (torch.rand(B, 784, generator=g) < 0.2).float().cuda()
    noise = torch.randn(B * nz, mc.noise_dim, generator=g).cuda()
    z = model.forward_hidden(x, nz=nz, noise=noise).clone()
    z0 = model.encode(x, std=0).clone()
    assert not torch.isnan(z).any()
    for b, off in ((256, 256), (128, 128), (64, 448), (32, 96)):
        zs = model.forward_hidden(x[off:off + b].contiguous(), nz=nz, noise=noise[off * nz:(off + b) * nz].contiguous())
        assert torch.equal(zs, z[off:off + b]), (b, float((zs - z[off:off + b]).abs().max()))
        z0s = model.encode(x[off:off + b].contiguous(), std=0)
        assert torch.equal(z0s.reshape(b, -1), z0.reshape(B, -1)[off:off + b]), b
    # h = 256 runs on the weight-stationary tail kernel (one wave per column group, any row count); the GENERAL tail kernel - 1 / 2 / 4
    # waves per 32-row block depending on the row count - must give the same bits: a child process with the weight-stationary kernel
    # switched off (ARDAE_TAIL_WS=0, a debug knob) on 512 / 64 / 32 images (1 / 2 / 4 waves per row block)
    if nonlin == "softplus":
        import subprocess, sys, tempfile
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with tempfile.TemporaryDirectory() as td:
            torch.save({"pm": pm, "x": x.cpu(), "noise": noise.cpu()}, os.path.join(td, "in.pt"))
            src = _TAIL_GENERAL_CHILD.format(root=root, tests=os.path.join(root, "tests"))
            env = dict({k: v for k, v in os.environ.items() if not k.startswith("ARDAE_")}, ARDAE_DEBUG_KNOBS="1", ARDAE_TAIL_WS="0")
            r = subprocess.run([sys.executable, "-c", src, td], env=env, capture_output=True, text=True, timeout=240)
            assert r.returncode == 0, r.stderr[-3000:]
            got = torch.load(os.path.join(td, "out.pt"), weights_only=True)
        assert got["kernels"] and all("sampler_tail_kernel" in k for k in got["kernels"]), got["kernels"]
        assert torch.equal(got["z512"], z.cpu()) and torch.equal(got["z64"], z[448:512].cpu()) and torch.equal(got["z32"], z[96:128].cpu())


_TAIL_GENERAL_CHILD = """
import sys, os, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
from ardae_amd import _lib as L
from oracle import ardae_oracle as O
import test_engine_gpu as T
d = torch.load(os.path.join(sys.argv[1], "in.pt"), weights_only=True)
mc = O.ModelCfg("mnist", 784, 100, 256, 32, 2, "softplus")
model, _ = T.build(mc, O.CdaeCfg("grad", 32, 32, 64, 2))
model.load_state_dict(d["pm"]); model = model.to("cuda")
x, noise, nz = d["x"].cuda(), d["noise"].cuda(), 256
L.lib().ardae_profile_enable(1)
out = dict(z512=model.forward_hidden(x, nz=nz, noise=noise).cpu(),
           z64=model.forward_hidden(x[448:512].contiguous(), nz=nz, noise=noise[448 * nz:512 * nz].contiguous()).cpu(),
           z32=model.forward_hidden(x[96:128].contiguous(), nz=nz, noise=noise[96 * nz:128 * nz].contiguous()).cpu())
out["kernels"] = [e["name"] for e in L.profile_report() if "sampler_tail" in e["name"]]
torch.save(out, os.path.join(sys.argv[1], "out.pt"))
"""


@pytest.mark.parametrize("m_opt,d_opt", [("amsgrad", "adam"), ("rmsprop", "sgd"), ("sgd", "amsgrad")])
def test_checkpoint_roundtrip_other_optimizers(tmp_path, m_opt, d_opt):
    """The four --m-optimizer / --d-optimizer choices keep torch.optim's state_dict() layouts (exp_avg / exp_avg_sq / max_exp_avg_sq,
    square_avg / momentum_buffer, nothing for SGD) and resume bit-identically, Adam's step counts included - under graph replay,
    with two cDAE updates per step (the cDAE's own step count advances twice per iteration)."""
    mc, cc, B = O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 8, 64, 3), 4
    tcfg = net.TrainConfig(nz_cdae=16, m_optimizer=m_opt, d_optimizer=d_opt, d_beta1=0.6, num_cdae_updates=2)

    def fresh(seeded):
        model, cdae = build(mc, cc)
        if seeded:
            model.load_state_dict(O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc)))
            cdae.load_state_dict(O.init_params(O.cdae_param_spec(cc), 1))
        model, cdae = model.to("cuda"), cdae.to("cuda")
        return model, cdae, net.ArdaeEngine(model, cdae, tcfg, batch_size=B)

    g = torch.Generator().manual_seed(5)
    xs = [torch.bernoulli(torch.full((B, 24), 0.3), generator=g).cuda() for _ in range(10)]
    net.manual_seed(77)
    model, cdae, eng = fresh(True)
    for t in range(4):                                    # eager, eager, capture + replay, replay
        eng.step(xs[2 * t], xs[2 * t + 1])
    mck, cck = eng.model_checkpoint(), eng.cdae_checkpoint()
    torch.save(mck, tmp_path / "m.pth.tar"); torch.save(cck, tmp_path / "c.pth.tar")
    mck2 = torch.load(tmp_path / "m.pth.tar", weights_only=True); cck2 = torch.load(tmp_path / "c.pth.tar", weights_only=True)
    names = {"sgd": set(), "adam": {"step", "exp_avg", "exp_avg_sq"}, "amsgrad": {"step", "exp_avg", "exp_avg_sq", "max_exp_avg_sq"},
             "rmsprop": {"step", "square_avg", "momentum_buffer"}}
    for ck, kind, steps in ((mck2, m_opt, 4), (cck2, d_opt, 8)):
        st = ck["optimizer"]["state"]
        assert (st == {}) if kind == "sgd" else (set(st[0]) == names[kind] and int(st[0]["step"]) == steps)
    # the same dicts load into the drop-in optimisers (= the reference's utils.Adam / torch.optim layouts)
    mk = {"sgd": lambda p: torch.optim.SGD(p, lr=1e-4), "adam": lambda p: net.Adam(p, lr=1e-4, betas=(0.5, 0.999)),
          "amsgrad": lambda p: net.Adam(p, lr=1e-4, betas=(0.5, 0.999), amsgrad=True), "rmsprop": lambda p: net.RMSprop(p, lr=1e-4, momentum=0.5)}
    model_c, cdae_c = build(mc, cc)
    mk[m_opt](list(model_c.to("cuda").parameters())).load_state_dict(mck2["optimizer"])
    mk[d_opt](list(cdae_c.to("cuda").parameters())).load_state_dict(cck2["optimizer"])
    eng.step(xs[8], xs[9])
    want_m, want_c = model.flat_params().clone(), cdae.flat_params().clone()
    model_b, cdae_b, eng_b = fresh(False)
    eng_b.load_checkpoints(mck2, cck2)
    assert eng_b.step_count == 4 and eng_b.opt_c.steps == 8
    eng_b.step(xs[8], xs[9])
    assert torch.equal(model_b.flat_params(), want_m) and torch.equal(cdae_b.flat_params(), want_c)
    # a checkpoint of another optimiser is refused, not silently reinterpreted (SGD's empty state included: param_groups tell)
    other = net.ArdaeEngine(*build_cuda(mc, cc), net.TrainConfig(nz_cdae=16, m_optimizer="rmsprop" if m_opt != "rmsprop" else "adam", d_optimizer=d_opt), batch_size=B)
    with pytest.raises(ValueError):
        other.load_checkpoints(mck2, cck2)


def build_cuda(mc, cc):
    model, cdae = build(mc, cc)
    return model.to("cuda"), cdae.to("cuda")


@pytest.mark.parametrize("B,nz,nstd,z", [(4, 8, 3, 8), (6, 256, 2, 32), (3, 625, 4, 32), (5, 16, 5, 2)])
def test_latent_perturb_nstd_kernel(B, nz, nstd, z):
    """--train-nstd-cdae > 1 (ivae_ardae.py:759-767): statistics over the nz samples, every sample row used nstd times with its own
    sigma and eps - row (b, i, j) <- latent[b, i]."""
    import ctypes
    from ardae_amd import _lib as L
    g = torch.Generator().manual_seed(B + nz + nstd + z)
    z0 = torch.randn(B, z, generator=g)
    latent = z0[:, None, :] + 0.05 * torch.randn(B, nz, z, generator=g)
    xi, eps = torch.randn(B, nz * nstd, 1, generator=g), torch.randn(B * nz * nstd, z, generator=g)
    u, std = O.latent_stats(latent, z0.view(B, 1, z), 1e4, 0.1)
    u_exp = u.unsqueeze(2).expand(B, nz, nstd, z).reshape(B * nz * nstd, z)
    sigma_ref = (std * xi).reshape(-1)
    xbar_ref = u_exp + sigma_ref[:, None] * eps
    d = lambda t: t.contiguous().cuda()
    N = B * nz * nstd
    xbar, sigma, std_b = torch.empty(N, z, device="cuda"), torch.empty(N, device="cuda"), torch.empty(B, device="cuda")
    lat_d, z0_d, xi_d, eps_d = d(latent), d(z0), d(xi.reshape(-1)), d(eps)          # keep the device copies alive across the call
    L.check(L.lib().ardae_latent_perturb_nstd(L.ptr(lat_d), L.ptr(z0_d), L.ptr(xi_d), L.ptr(eps_d), B, nz, nstd, z, 1e4, 0.1,
                                              L.ptr(xbar), L.ptr(sigma), L.ptr(std_b), L.stream_ptr()), "ardae_latent_perturb_nstd")
    torch.cuda.synchronize()
    assert rel_l2(std_b, std.reshape(-1)) < 1e-5 and rel_l2(sigma, sigma_ref) < 1e-5 and rel_l2(xbar, xbar_ref) < 1e-5


@pytest.mark.parametrize("B,nz", [(128, 256), (64, 1250)])
def test_auxconv_nrow_sampler_fused_two_head_tail_vs_oracle(B, nz):
    """MNISTConvAuxIPVAE.forward_hidden on >= 32768 rows: z0 -> 800 -> (mean | logvar) runs as one launch with the hidden rows on chip
    (sampler_tail_kernel with two heads; 25 column blocks - an odd count - and row-bias groups that straddle the 32-row blocks)."""
    mc = O.ModelCfg("auxconv", 784, 100, 800, 32, 1, "softplus")
    pm = O.init_params(O.model_param_spec(mc), 11, O.model_init_special(mc))
    model, _ = build(mc, O.CdaeCfg("grad", 32, 1600, 64, 2))
    model.load_state_dict(pm)
    model = model.to("cuda")
    g = torch.Generator().manual_seed(B + nz)
    x = (torch.rand(B, 784, generator=g) < 0.2).float()
    e0, e = torch.randn(B * nz, mc.noise_dim, generator=g), torch.randn(B * nz, mc.z_dim, generator=g)
    z = model.forward_hidden(x.cuda(), nz=nz, noise=(e0.cuda(), e.cuda()))
    pm64 = {k: v.double() for k, v in pm.items()}
    ref = O.encode(mc, pm64, x.double(), (e0.double(), e.double()), nz).reshape(B * nz, -1).float()
    assert z.shape == (B, nz, mc.z_dim)
    assert rel_l2(z.reshape(B * nz, -1), ref) < 2e-5
    assert float((z.reshape(B * nz, -1).cpu() - ref).abs().max()) < 1e-3


def _oracle_opt_state(ckpt, names, keys):
    """torch.optim-style per-parameter state of an engine checkpoint -> the oracle's {name: {...}} form."""
    out = {}
    for i, n in enumerate(names):
        if i in ckpt["optimizer"]["state"]:
            s = ckpt["optimizer"]["state"][i]
            out[n] = dict({"step": int(s["step"])}, **{k: s[k].detach().cpu().clone() for k in keys})
    return out


@pytest.mark.parametrize("kind", ["mnist", "toy"])
def test_engine_trajectory_production_shapes_vs_live_oracle(kind):
    """Twenty consecutive train steps at PRODUCTION kernel shapes (32 images x 256 samples = 8192 rows: software-pipelined N-row
    kernels, 256 x 256 / 256 x 32 weight gradients), every step against the live oracle.

      mnist: BASELINE config #2 - MNISTIPVAE 784 / 100 / h 256 / z 32 + mlp-grad h 256 L 3 (models/ivae/mnist.py, graddae/mlp.py)
      toy:   BASELINE config #1 - ToyIPVAE z 2, relu, ContextConcatMLP sampler (two-source 266-wide layers, models/layers.py:681-724),
             cDAE first layers with K = 2

    The oracle is TEACHER-FORCED: before every step it takes the engine's parameters and optimiser state (RMSprop square_avg /
    momentum_buffer, Adam exp_avg / exp_avg_sq / step - through the reference-format checkpoint dicts), runs the same step with the same
    images and the same injected noise, and the two are compared: losses at 1e-4 relative (north star; measured 5e-7), recon / prior 2e-5,
    and the parameter UPDATE of that step.  Only the very first update is sign-like (RMSprop / Adam normalise a gradient at the fp32
    noise floor to a full-size step: relative L2 2.5-4e-2 at step 0, the median element still agrees to 1e-6); once the second-moment
    estimates carry history the update is continuous in the gradient: from step 1 on the relative L2 of the update is typically 1e-4
    (criterion at the end of the test: median below 5e-3, at most one spike above 1e-2, none above 5e-2;
    measured at steps 1 / 5 / 10 / 15 / 19: model 4e-4 ... 8e-6, cDAE 2e-3 (mnist) / 2-5e-4 (toy) - the cDAE's is its gradient
    error, cf. test_cdae_gpu.py; about 3 % of the cDAE's elements, those whose gradient is fp32 noise, still differ by more than 1e-2
    of their own tiny update, which is why the element-wise criterion of assert_update_close is used for step 0 only).  A free-running
    comparison would only measure how chaotic the training dynamics are (the toy problem's loss swings 22 -> 347 -> 42 in its first
    steps): after 20 steps the free-running oracle's cDAE loss differs by 0.2 % (mnist) / 5 % (toy) while every single step agrees to 1e-7."""
    if kind == "mnist":
        mc, cc = O.ModelCfg("mnist", 784, 100, 256, 32, 2, "softplus"), O.CdaeCfg("grad", 32, 32, 256, 3)
    else:
        mc, cc = O.ModelCfg("toy", 2, 10, 256, 2, 2, "relu"), O.CdaeCfg("grad", 2, 2, 256, 3)
    B, NZ, STEPS = 32, 256, 20
    tc = O.TrainCfg(nz_cdae=NZ)
    pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc))
    pc = O.init_params(O.cdae_param_spec(cc), 1)
    gen = torch.Generator().manual_seed(11)

    def batch():
        if kind == "mnist":
            return torch.bernoulli(torch.full((B, 784), 0.2), generator=gen)
        return torch.randn(B, 2, generator=gen) * 0.3 + torch.randint(-2, 3, (B, 2), generator=gen).float() * 2      # the 25-Gaussians grid

    model, cdae = build(mc, cc)
    model.load_state_dict(pm); cdae.load_state_dict(pc)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=NZ), batch_size=B)
    mnames, cnames = [n for n, _ in O.model_param_spec(mc)], [n for n, _ in O.cdae_param_spec(cc)]
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    worst = {"loss": 0.0, "late_m": 0.0, "late_c": 0.0}
    late = []
    for t in range(STEPS):
        x1, x2, noise = batch(), batch(), O.draw_step_noise(mc, tc, B, gen)
        mck, cck = eng.model_checkpoint(), eng.cdae_checkpoint()
        rm = {n: mck["state_dict"][n].detach().cpu().clone() for n in mnames}
        rc = {n: cck["state_dict"][n].detach().cpu().clone() for n in cnames}
        st_m = _oracle_opt_state(mck, mnames, ("exp_avg", "exp_avg_sq"))
        st_c = _oracle_opt_state(cck, cnames, ("square_avg", "momentum_buffer"))
        before_m = torch.cat([rm[n].reshape(-1) for n in mnames]); before_c = torch.cat([rc[n].reshape(-1) for n in cnames])
        eng.step(x1.cuda(), x2.cuda(), noise={k: v.cuda().contiguous() for k, v in noise.items()})
        got = eng.stats()
        ref = O.train_step(mc, cc, tc, rm, rc, st_m, st_c, x1, x2, noise)
        for k in ("cdae_loss", "model_loss"):
            assert rel(got[k], ref[k]) < 1e-4, (t, k, got[k], float(ref[k]))
            worst["loss"] = max(worst["loss"], rel(got[k], ref[k]))
        for k in ("recon", "prior"):
            assert rel(got[k], ref[k]) < 2e-5, (t, k)
        after_m, after_c = model.flat_params().cpu(), cdae.flat_params().cpu()
        ref_m = torch.cat([rm[n].reshape(-1) for n in mnames]); ref_c = torch.cat([rc[n].reshape(-1) for n in cnames])
        if t >= 1:       # the optimisers' second moments carry history: the update is continuous in the gradient
            um, uc = rel_l2(after_m - before_m, ref_m - before_m), rel_l2((after_c - before_c)[:-1], (ref_c - before_c)[:-1])
            worst["late_m"], worst["late_c"] = max(worst["late_m"], um), max(worst["late_c"], uc)
            late.append((um, uc))
            assert um < 5e-2 and uc < 5e-2, (t, um, uc)
        else:            # the very first step: sign-like updates (median element + loose L2, see assert_update_close)
            assert_update_close(after_c[:-1], before_c[:-1], ref_c[:-1], f"cdae update, step {t}")
            assert_update_close(after_m, before_m, ref_m, f"model update, step {t}")
    # The per-step update error is heavy-tailed: a step where RMSprop / Adam normalise a few gradient elements that sit at the fp32 noise floor
    # shows up as a spike of the relative L2 (round 4, toy, 19 late steps, two builds of the kernels that differ only in the order of the
    # fp32 sums: typical 4e-5 ... 4e-4 for both; spikes 7.7e-3 (model, step 1) / 1.5e-2 (cDAE, step 5) with one, 3.4e-3 (cDAE, step 6)
    # with the other - which step spikes follows the rounding, not the build).  Criterion: the TYPICAL step (median) below 5e-3 (the
    # mnist cDAE's level is 1.5 - 2e-3, its gradient error: test_cdae_gpu.py; everything else sits at 1e-4), at most one step of a network
    # above 1e-2, none above 5e-2.
    import statistics
    for k, name in ((0, "model"), (1, "cdae")):
        errs = [e[k] for e in late]
        assert statistics.median(errs) < 5e-3, (name, sorted(errs)[-3:])
        assert sum(e > 1e-2 for e in errs) <= 1, (name, sorted(errs)[-3:])
    print(f"{kind}: worst loss error {worst['loss']:.1e}, worst late-step update error model {worst['late_m']:.1e} cdae {worst['late_c']:.1e}")
