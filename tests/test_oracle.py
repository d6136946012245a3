"""Oracle (CPU restatement) against the golden vectors generated from the reference (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import ardae_oracle as O

CASES = {
    "tiny_mnist_grad": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 8, 64, 3), 8, torch.float32),
    "tiny_mnist_grad_f64": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 8, 64, 3), 8, torch.float64),
    "tiny_mnist_res": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("res", 8, 8, 64, 3), 8, torch.float32),
    "tiny_toy_grad": (O.ModelCfg("toy", 2, 10, 64, 2, 2, "relu"), O.CdaeCfg("grad", 2, 2, 64, 3), 8, torch.float32),
    # hierarchical (aux) sampler + hidden1a context of the shipped "hierarchical mlp" recipe: the oracle of SURVEY 8 f-3's first
    # family is pinned against the reference's MNISTAuxIPVAE here; its HIP path is the next row to build
    "tiny_auxmnist_grad": (O.ModelCfg("auxmnist", 24, 10, 48, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 96, 64, 3), 8, torch.float32),
    "tiny_auxmnist_grad_f64": (O.ModelCfg("auxmnist", 24, 10, 48, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 96, 64, 3), 8, torch.float64),
    # --model auxmlp: ToyAuxIPVAE (q z0's x q z's per image: nz_cdae 16 = 4 x 4), Gaussian decoder, tanh, hidden1a context
    "tiny_auxtoy_grad": (O.ModelCfg("auxtoy", 2, 2, 32, 2, 2, "tanh"), O.CdaeCfg("grad", 2, 64, 64, 3), 16, torch.float32),
    "tiny_auxtoy_grad_f64": (O.ModelCfg("auxtoy", 2, 2, 32, 2, 2, "tanh"), O.CdaeCfg("grad", 2, 64, 64, 3), 16, torch.float64),
    "tiny_auxmnist_clip": (O.ModelCfg("auxmnist", 24, 10, 48, 8, 2, "softplus", clip_z0="spm4", clip_z="2tanh"), O.CdaeCfg("grad", 8, 96, 64, 3), 8, torch.float32),
    "tiny_auxtoy_clip": (O.ModelCfg("auxtoy", 2, 2, 32, 2, 2, "tanh", clip_z0="hard", clip_z="softplus"), O.CdaeCfg("grad", 2, 64, 64, 3), 16, torch.float32),
    # the other activations of get_nonlinear_func (utils/models.py:14-32): tanh = the class default of the reference's models and cDAEs,
    # relu = the default of --cdae-nonlin (mlp-grad: second-order terms vanish), elu, leaky_relu
    "tiny_toy_tanh": (O.ModelCfg("toy", 2, 10, 64, 2, 2, "tanh"), O.CdaeCfg("grad", 2, 2, 64, 3, "tanh"), 8, torch.float32),
    "tiny_mnist_elu": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "elu"), O.CdaeCfg("grad", 8, 8, 64, 3, "elu"), 8, torch.float32),
    "tiny_mnist_leaky": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "leaky_relu"), O.CdaeCfg("res", 8, 8, 64, 3, "leaky_relu"), 8, torch.float32),
    "tiny_toy_relu_relu": (O.ModelCfg("toy", 2, 10, 64, 2, 2, "relu"), O.CdaeCfg("grad", 2, 2, 64, 3, "relu"), 8, torch.float32),
    "tiny_mnist_tanh_res": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "tanh"), O.CdaeCfg("res", 8, 8, 64, 3, "tanh"), 8, torch.float32),
    # swish (utils/models.py:8-10): the seventh and last name of get_nonlinear_func (round 4)
    "tiny_mnist_swish": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "swish"), O.CdaeCfg("grad", 8, 8, 64, 3, "swish"), 8, torch.float32),
    "tiny_toy_swish_res": (O.ModelCfg("toy", 2, 10, 64, 2, 2, "swish"), O.CdaeCfg("res", 2, 2, 64, 3, "swish"), 8, torch.float32),
    # --cdae-ctx-type data: the image itself as context (centred for the MNIST family)
    "tiny_mnist_ctxdata": (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 24, 64, 3), 8, torch.float32),
    "tiny_toy_ctxdata": (O.ModelCfg("toy", 2, 10, 64, 2, 2, "relu"), O.CdaeCfg("res", 2, 2, 64, 3), 8, torch.float32),
}
CASES["tiny_mnist_nstd3"] = (O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 8, 64, 3), 8, torch.float32)   # --train-nstd-cdae 3
CTX_DATA = {"tiny_mnist_ctxdata": True, "tiny_toy_ctxdata": False}
# --m-optimizer / --d-optimizer pairs beyond the recipes' (adam, rmsprop): the oracle's optimisers against the reference's over three steps
OPT_PAIRS = {"tiny_mnist_opt_adam_adam": ("adam", "adam"), "tiny_mnist_opt_amsgrad_sgd": ("amsgrad", "sgd"),
             "tiny_mnist_opt_rmsprop_amsgrad": ("rmsprop", "amsgrad"), "tiny_mnist_opt_sgd_rmsprop": ("sgd", "rmsprop")}


@pytest.mark.parametrize("name", list(OPT_PAIRS))
def test_oracle_optimizer_pairs_match_reference_fixture(golden_dir, name):
    mc, cc = O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 8, 64, 3)
    tc = O.TrainCfg(nz_cdae=8, m_optimizer=OPT_PAIRS[name][0], d_optimizer=OPT_PAIRS[name][1], d_beta1=0.7, m_lr=2e-4, d_lr=3e-4)
    fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    pm = {n: torch.tensor(fx["pm/" + n]) for n, _ in O.model_param_spec(mc)}
    pc = {n: torch.tensor(fx["pc/" + n]) for n, _ in O.cdae_param_spec(cc)}
    st_m, st_c = {}, {}
    for t in range(int(fx["meta_steps"])):
        pre = f"s{t}/"
        # the reference's own gradients: this test is about the update rules and their state across steps
        gc = {n: (None if pre + "cdae_grads/" + n + "/none" in fx else torch.tensor(fx[pre + "cdae_grads/" + n])) for n, _ in O.cdae_param_spec(cc)}
        gm = {n: torch.tensor(fx[pre + "model_grads/" + n]) for n, _ in O.model_param_spec(mc)}
        with torch.no_grad():
            O.optimizer_step(tc.d_optimizer, pc, gc, st_c, tc.d_lr, tc.d_beta1, tc.d_momentum)
            O.optimizer_step(tc.m_optimizer, pm, gm, st_m, tc.m_lr, tc.m_beta1, tc.d_momentum)
        for n, _ in O.cdae_param_spec(cc):
            ref = torch.tensor(fx[pre + "cdae_params_after/" + n])
            assert torch.allclose(pc[n], ref, rtol=0, atol=2e-7 + 1e-3 * tc.d_lr), (n, t)
            pc[n] = ref.clone()
        for n, _ in O.model_param_spec(mc):
            ref = torch.tensor(fx[pre + "model_params_after/" + n])
            assert torch.allclose(pm[n], ref, rtol=0, atol=2e-7 + 1e-3 * tc.m_lr), (n, t)
            pm[n] = ref.clone()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_step_matches_reference_fixture(golden_dir, name):
    mc, cc, nz, dt = CASES[name]
    fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    tc = O.TrainCfg(nz_cdae=nz, ctx_type="hidden1a" if mc.kind in ("auxmnist", "auxtoy") else "lt0")
    if name in CTX_DATA:
        tc = O.TrainCfg(nz_cdae=nz, ctx_type="data", ctx_center=CTX_DATA[name])
    if name == "tiny_mnist_nstd3":
        tc = O.TrainCfg(nz_cdae=nz, nstd=3)
    tol = 2e-4 if dt == torch.float32 else 1e-9    # another CPU/BLAS than the one that wrote the fixture: fp32 noise x 1e4 (std_scale)
    pm = {n: torch.tensor(fx["pm/" + n]) for n, _ in O.model_param_spec(mc)}
    pc = {n: torch.tensor(fx["pc/" + n]) for n, _ in O.cdae_param_spec(cc)}
    for t in range(int(fx["meta_steps"])):
        pre = f"s{t}/"
        noise = {k[len(pre + "noise/"):]: torch.tensor(v) for k, v in fx.items() if k.startswith(pre + "noise/")}
        xc, xv = torch.tensor(fx[pre + "x_cdae"]), torch.tensor(fx[pre + "x_vae"])
        closs, gc, std = O.cdae_update_grads(mc, cc, tc, pm, pc, xc, noise)
        assert abs(float(closs) - float(fx[pre + "cdae_loss"])) / float(fx[pre + "cdae_loss"]) < tol
        assert rel_l2(std, torch.tensor(fx[pre + "std"])) < tol
        for n, _ in O.cdae_param_spec(cc):
            if pre + "cdae_grads/" + n + "/none" in fx:
                assert gc[n] is None
            else:
                assert rel_l2(gc[n], torch.tensor(fx[pre + "cdae_grads/" + n])) < 20 * tol, n
        pc = {n: torch.tensor(fx[pre + "cdae_params_after/" + n]) for n, _ in O.cdae_param_spec(cc)}
        mloss, rec, pri, g, gm = O.vae_update_grads(mc, cc, tc, pm, pc, xv, noise)
        assert abs(float(mloss) - float(fx[pre + "model_loss"])) / float(fx[pre + "model_loss"]) < tol
        assert abs(float(rec) - float(fx[pre + "recon"])) / float(fx[pre + "recon"]) < tol
        assert rel_l2(g, torch.tensor(fx[pre + "score"])) < 20 * tol
        for n, _ in O.model_param_spec(mc):
            assert rel_l2(gm[n], torch.tensor(fx[pre + "model_grads/" + n])) < 20 * tol, n
        pm = {n: torch.tensor(fx[pre + "model_params_after/" + n]) for n, _ in O.model_param_spec(mc)}


@pytest.mark.parametrize("name,mc,cc,ctx", [
    ("conv_b4_nz8", O.ModelCfg("conv", 784, 100, 800, 32, 1, "softplus"), O.CdaeCfg("grad", 32, 32, 64, 2), "lt0"),
    # hierarchical conv model (--model auxconv): oracle pinned against the reference's MNISTConvAuxIPVAE; its HIP path is not built yet
    ("auxconv_b4_nz8", O.ModelCfg("auxconv", 784, 100, 800, 32, 1, "softplus"), O.CdaeCfg("grad", 32, 1600, 64, 2), "hidden1a"),
    # weight-normalised residual-conv families of the shipped "implicit resconv" / "hierarchical resconv" recipes
    ("resconv_b4_nz8", O.ModelCfg("resconv", 784, 100, 512, 32, 1, "elu"), O.CdaeCfg("res", 32, 32, 64, 2), "lt0"),
    ("auxresconv_b4_nz8", O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu"), O.CdaeCfg("res", 32, 450, 64, 2), "hidden1a"),
    # --model resconv-res / auxresconv: the same families with do_center=False
    ("resconv_nocenter_b4_nz8", O.ModelCfg("resconv", 784, 100, 512, 32, 1, "elu", do_center=False), O.CdaeCfg("res", 32, 32, 64, 2), "lt0"),
    ("auxresconv_nocenter_b4_nz8", O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu", do_center=False), O.CdaeCfg("res", 32, 450, 64, 2), "hidden1a"),
    # --model auxresconv-clip / auxresconvct-clip (MNISTResConvAuxIPVAEClipped): the fixture's noise holds the unscaled eps0 of the std = 0 calls
    ("auxresconv_clip_b4_nz8", O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu", clipped=True), O.CdaeCfg("res", 32, 450, 64, 2), "hidden1a"),
    ("auxresconv_clip_nocenter_b4_nz8", O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu", do_center=False, clipped=True), O.CdaeCfg("res", 32, 450, 64, 2), "hidden1a"),
    # the other sampler heads of ResConvIPVAE (--model resconv / resconvct, -res2, -res3, -res4; one and two hidden layers)
] + [(nm, O.ModelCfg("resconv", 784, 100, 512, 32, nl, "elu", do_center=ctr, enc_type=et), O.CdaeCfg("res", 32, 32, 64, 2), "lt0") for nm, et, nl, ctr in (
    ("resconv_mlp_b4_nz8", "mlp", 1, True), ("resconv_mlp2_nocenter_b4_nz8", "mlp", 2, False), ("resconv_res2_b4_nz8", "res-mlp", 1, False),
    ("resconv_res2x2_b4_nz8", "res-mlp", 2, True), ("resconv_res3_b4_nz8", "res-wn-mlp-lin", 1, True), ("resconv_res3x2_b4_nz8", "res-wn-mlp-lin", 2, False),
    ("resconv_res4_b4_nz8", "res-mlp-lin", 1, False), ("resconv_resx2_b4_nz8", "res-wn-mlp", 2, True))])
def test_oracle_step_matches_reference_summaries(golden_dir, name, mc, cc, ctx):
    """Fixtures that hold summaries only (parameters regenerated from the seed): losses, latent statistics and the norm / sum /
    first elements of every gradient of step 0."""
    fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    tc = O.TrainCfg(nz_cdae=8, ctx_type=ctx)
    if "resconv" in mc.kind:      # run_vae_dbmnist.sh: --std-scale 100, Adam (0.9, 0.999) lr 1e-3, RMSprop momentum 0.9
        tc = O.TrainCfg(nz_cdae=8, std_scale=100., m_lr=1e-3, m_beta1=0.9, d_momentum=0.9, ctx_type=ctx)
    ps = int(fx["meta_pseed"])
    pm = O.init_params(O.model_param_spec(mc), ps, O.model_init_special(mc))
    pc = O.init_params(O.cdae_param_spec(cc), ps + 1)
    pre = "s0/"
    noise = {k[len(pre + "noise/"):]: torch.tensor(v) for k, v in fx.items() if k.startswith(pre + "noise/")}
    xc, xv = torch.tensor(fx[pre + "x_cdae"]), torch.tensor(fx[pre + "x_vae"])
    closs, gc, std = O.cdae_update_grads(mc, cc, tc, pm, pc, xc, noise)
    assert abs(float(closs) - float(fx[pre + "cdae_loss"])) / float(fx[pre + "cdae_loss"]) < 2e-4
    assert rel_l2(std, torch.tensor(fx[pre + "std"])) < 2e-4
    for n, _ in O.cdae_param_spec(cc):
        if pre + "cdae_grads/" + n + "/none" in fx:
            assert gc[n] is None
        else:
            assert abs(float(gc[n].double().norm()) - float(fx[pre + "cdae_grads/" + n + "/norm"])) <= 4e-3 * float(fx[pre + "cdae_grads/" + n + "/norm"]), n
    st_c = {}
    with torch.no_grad():
        O.rmsprop_step(pc, gc, st_c, tc.d_lr, tc.d_momentum)
    mloss, rec, pri, g, gm = O.vae_update_grads(mc, cc, tc, pm, pc, xv, noise)
    assert abs(float(mloss) - float(fx[pre + "model_loss"])) / float(fx[pre + "model_loss"]) < 2e-4
    assert abs(float(rec) - float(fx[pre + "recon"])) / float(fx[pre + "recon"]) < 2e-4
    for n, _ in O.model_param_spec(mc):
        ref = float(fx[pre + "model_grads/" + n + "/norm"])
        assert abs(float(gm[n].double().norm()) - ref) <= 4e-3 * ref, n


def test_oracle_optimizers_match_reference_fixture(golden_dir):
    mc, cc, nz, _ = CASES["tiny_mnist_grad"]
    fx = dict(np.load(os.path.join(golden_dir, "tiny_mnist_grad.npz")))
    pm = {n: torch.tensor(fx["pm/" + n]) for n, _ in O.model_param_spec(mc)}
    pc = {n: torch.tensor(fx["pc/" + n]) for n, _ in O.cdae_param_spec(cc)}
    st_m, st_c = {}, {}
    for t in range(int(fx["meta_steps"])):
        pre = f"s{t}/"
        gc = {n: (None if pre + "cdae_grads/" + n + "/none" in fx else torch.tensor(fx[pre + "cdae_grads/" + n])) for n in pc}
        gm = {n: torch.tensor(fx[pre + "model_grads/" + n]) for n in pm}
        O.rmsprop_step(pc, gc, st_c, 1e-4, 0.5)
        O.adam_ref_step(pm, gm, st_m, 1e-4, 0.5)
        for n in pc:
            assert torch.allclose(pc[n], torch.tensor(fx[pre + "cdae_params_after/" + n]), rtol=3e-7, atol=1e-9), n
        for n in pm:
            assert torch.allclose(pm[n], torch.tensor(fx[pre + "model_params_after/" + n]), rtol=3e-7, atol=1e-9), n
        pc = {n: torch.tensor(fx[pre + "cdae_params_after/" + n]) for n in pc}
        pm = {n: torch.tensor(fx[pre + "model_params_after/" + n]) for n in pm}
    assert "neglogprob.fc.bias" not in st_c            # no gradient -> no optimiser state


def test_oracle_iwae_matches_reference_fixture(golden_dir):
    fx = dict(np.load(os.path.join(golden_dir, "iwae_tiny.npz")))
    mc = O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus")
    pm = {n: torch.tensor(fx["pm/" + n]) for n, _ in O.model_param_spec(mc)}
    got = O.iwae_logprob(mc, pm, torch.tensor(fx["x"]), int(fx["meta_k"]), torch.tensor(fx["enc_noise"]), torch.tensor(fx["prop_noise"]))
    assert abs(float(got) - float(fx["logprob"])) / abs(float(fx["logprob"])) < 1e-9


def test_oracle_iwae_matches_reference_fixture_aux(golden_dir):
    fx = dict(np.load(os.path.join(golden_dir, "iwae_tiny_auxmnist.npz")))
    mc = O.ModelCfg("auxmnist", 24, 10, 48, 8, 2, "softplus")
    pm = {n: torch.tensor(fx["pm/" + n]) for n, _ in O.model_param_spec(mc)}
    got = O.iwae_logprob(mc, pm, torch.tensor(fx["x"]), int(fx["meta_k"]), (torch.tensor(fx["enc_noise"]), torch.tensor(fx["enc_noise_z"])),
                         torch.tensor(fx["prop_noise"]))
    assert abs(float(got) - float(fx["logprob"])) / abs(float(fx["logprob"])) < 1e-9


def test_oracle_iwae_matches_reference_fixture_auxtoy(golden_dir):
    fx = dict(np.load(os.path.join(golden_dir, "iwae_tiny_auxtoy.npz")))
    mc = O.ModelCfg("auxtoy", 2, 2, 32, 2, 2, "tanh")
    pm = {n: torch.tensor(fx["pm/" + n]) for n, _ in O.model_param_spec(mc)}
    got = O.iwae_logprob(mc, pm, torch.tensor(fx["x"]), int(fx["meta_k"]), (torch.tensor(fx["enc_noise"]), torch.tensor(fx["enc_noise_z"])),
                         torch.tensor(fx["prop_noise"]))
    assert abs(float(got) - float(fx["logprob"])) / abs(float(fx["logprob"])) < 1e-9


def test_oracle_iwae_matches_reference_fixture_auxconv(golden_dir):
    fx = dict(np.load(os.path.join(golden_dir, "iwae_auxconv.npz")))
    mc = O.ModelCfg("auxconv", 784, 100, 800, 32, 1, "softplus")
    pm = O.init_params(O.model_param_spec(mc), int(fx["meta_pseed"]), O.model_init_special(mc), torch.float64)
    got = O.iwae_logprob(mc, pm, torch.tensor(fx["x"]), int(fx["meta_k"]), (torch.tensor(fx["enc_noise"]), torch.tensor(fx["enc_noise_z"])),
                         torch.tensor(fx["prop_noise"]))
    assert abs(float(got) - float(fx["logprob"])) / abs(float(fx["logprob"])) < 1e-9


@pytest.mark.parametrize("name,mc", [("iwae_resconv", O.ModelCfg("resconv", 784, 100, 512, 32, 1, "elu")),
                                     ("iwae_resconv_mlp", O.ModelCfg("resconv", 784, 100, 512, 32, 1, "elu", enc_type="mlp")),
                                     ("iwae_auxresconv", O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu")),
                                     ("iwae_auxresconv_clip", O.ModelCfg("auxresconv", 784, 100, 450, 32, 1, "elu", clipped=True))])
def test_oracle_iwae_matches_reference_fixture_resconv(golden_dir, name, mc):
    fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    pm = O.init_params(O.model_param_spec(mc), int(fx["meta_pseed"]), O.model_init_special(mc), torch.float64)
    enc = (torch.tensor(fx["enc_noise"]), torch.tensor(fx["enc_noise_z"])) if "enc_noise_z" in fx else torch.tensor(fx["enc_noise"])
    got = O.iwae_logprob(mc, pm, torch.tensor(fx["x"]), int(fx["meta_k"]), enc, torch.tensor(fx["prop_noise"]))
    assert abs(float(got) - float(fx["logprob"])) / abs(float(fx["logprob"])) < 1e-9


def test_closed_form_known_answer_one_layer_score():
    """Analytic score of a 1-layer softplus energy (known-answer test): with L=1 the cDAE is
    E = w . sp(W1a sp(A1 x + b1) + ...), whose input-gradient is checked against autograd in float64."""
    cc = O.CdaeCfg("grad", 3, 2, 5, 1)
    p = O.init_params(O.cdae_param_spec(cc), 11, dtype=torch.float64)
    x = torch.randn(4, 3, dtype=torch.float64, requires_grad=True)
    ctx = torch.randn(4, 2, dtype=torch.float64)
    sig = torch.rand(4, 1, dtype=torch.float64)
    g = O.cdae_score(cc, p, x, ctx, sig, create_graph=False)
    A1, b1 = p["inp_encode.fc.weight"], p["inp_encode.fc.bias"]
    C1, c1 = p["ctx_encode.fc.weight"], p["ctx_encode.fc.bias"]
    W1, d1, w = p["neglogprob.layers.0.weight"], p["neglogprob.layers.0.bias"], p["neglogprob.fc.weight"]
    a = torch.nn.functional.softplus(x.detach() @ A1.T + b1)
    c = torch.nn.functional.softplus(ctx @ C1.T + c1)
    q = torch.cat([a, c, sig], 1) @ W1.T + d1
    e = -w * torch.sigmoid(q)                                           # e_L = -w (.) s(q_L)
    r = (e @ W1[:, :5]) * torch.sigmoid(x.detach() @ A1.T + b1)         # r_L = (W1a^T e) (.) s(p_L)
    assert torch.allclose(g, r @ A1, rtol=1e-12, atol=1e-14)            # g = A1^T r
