"""Drop-in surface details: reference-written checkpoints, the decoder sample the
reference's forward() / generate() return, the batched Cholesky of the IWAE proposal, optimisers that write through p.data,
batch validation, the scalar log channel and the static-binarised batch source."""
import json
import os

import numpy as np
import pytest
import torch

import ardae_amd as net
from ardae_amd import _lib as L
from oracle import ardae_oracle as O
from test_engine_gpu import assert_update_close, build, rel, rel_l2

pytestmark = pytest.mark.gpu

TINY_M = O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus")
TINY_C = O.CdaeCfg("grad", 8, 8, 64, 3)


def _reference_checkpoint_dicts(fx):
    """Rebuild, from the fixture's plain arrays, the dicts the reference loop saves (ivae_ardae.py:1117-1139, utils/msc.py:67-72)."""
    lay = json.loads(str(fx["layout_json"]))
    out = {}
    for tag, opt in (("model", "m_opt"), ("cdae", "c_opt")):
        sd = {k: torch.tensor(fx[f"{tag}_sd/{k}"]) for k in lay[tag + "_state_dict"]}
        state = {}
        for i, keys in lay[opt + "_state_keys"].items():
            tensor_fields = set(lay.get(opt + "_tensor_fields", {}).get(i, []))
            state[int(i)] = {k: (torch.tensor(fx[f"{opt}/{i}/{k}"]) if k in tensor_fields else int(fx[f"{opt}/{i}/{k}"])) for k in keys
                             if f"{opt}/{i}/{k}" in fx}
        out[tag] = {"epoch": 0, "batch_idx": 1, "train_num_iters_per_epoch": 100, tag: "x", "state_dict": sd, "best_val_loss": float("inf"),
                    "optimizer": {"state": state, "param_groups": lay[opt + "_param_groups"]}, "scheduler": None}
    return out["model"], out["cdae"], lay


def test_engine_loads_reference_written_checkpoint(golden_dir):
    """f-2: the state the REFERENCE's objects write after two steps (model / cDAE state_dict, utils.Adam and torch.optim.RMSprop
    state_dict()) goes into the fused engine, which must then land where the reference's step 3 lands; the engine's own
    checkpoint after that step has the reference's key layout and the reference's optimiser state."""
    fx = dict(np.load(os.path.join(golden_dir, "ckpt_tiny_mnist_grad.npz")))
    mck, cck, lay = _reference_checkpoint_dicts(fx)
    B, k = int(fx["meta_B"]), int(fx["meta_k"])
    model, cdae = build(TINY_M, TINY_C)                       # fresh random parameters: everything must come from the checkpoint
    model, cdae = model.to("cuda"), cdae.to("cuda")
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=8), batch_size=B)
    eng.load_checkpoints(mck, cck)
    assert eng.step_count == k
    before_c, before_m = cdae.flat_params().clone().cpu(), model.flat_params().clone().cpu()
    assert torch.equal(before_m, torch.cat([mck["state_dict"][n].reshape(-1) for n in lay["model_state_dict"]]))
    noise = {kk: torch.tensor(fx["noise/" + kk]).cuda().contiguous() for kk in ("sampler", "sigma", "eps", "vae")}
    eng.step(torch.tensor(fx["x_cdae"]).cuda(), torch.tensor(fx["x_vae"]).cuda(), noise=noise)
    s = eng.stats()
    for kk in ("cdae_loss", "model_loss"):
        assert rel(s[kk], fx[kk]) < 1e-4, kk
    ref_c = torch.cat([torch.tensor(fx["cdae_params_after/" + n]).reshape(-1) for n, _ in O.cdae_param_spec(TINY_C)])
    ref_m = torch.cat([torch.tensor(fx["model_params_after/" + n]).reshape(-1) for n, _ in O.model_param_spec(TINY_M)])
    assert_update_close(cdae.flat_params().cpu(), before_c, ref_c, "cdae update after resume")
    assert_update_close(model.flat_params().cpu(), before_m, ref_m, "model update after resume")
    # the engine's own files: the reference's keys, shapes and (after the same step) the reference's optimiser state
    mo, co = eng.model_checkpoint(), eng.cdae_checkpoint()
    assert list(mo["state_dict"]) == lay["model_state_dict"] and list(co["state_dict"]) == lay["cdae_state_dict"]
    assert {str(i) for i in mo["optimizer"]["state"]} == set(lay["m_opt_state_keys"]) and {str(i) for i in co["optimizer"]["state"]} == set(lay["c_opt_state_keys"])
    for i, st in mo["optimizer"]["state"].items():
        assert sorted(st) == lay["m_opt_state_keys"][str(i)] and int(st["step"]) == k + 1
        assert st["exp_avg"].shape == mck["optimizer"]["state"][i]["exp_avg"].shape
        assert rel_l2(st["exp_avg"], fx[f"m_opt_after/{i}/exp_avg"]) < 2e-3, i
        assert rel_l2(st["exp_avg_sq"], fx[f"m_opt_after/{i}/exp_avg_sq"]) < 4e-3, i
    for i, st in co["optimizer"]["state"].items():
        assert sorted(st) == lay["c_opt_state_keys"][str(i)] and int(st["step"]) == k + 1
        assert rel_l2(st["square_avg"], fx[f"c_opt_after/{i}/square_avg"]) < 4e-3, i
        assert rel_l2(st["momentum_buffer"], fx[f"c_opt_after/{i}/momentum_buffer"]) < 5e-2, i     # sign-like first steps (see assert_update_close)
    for key in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
        assert list(np.atleast_1d(mo["optimizer"]["param_groups"][0][key])) == list(np.atleast_1d(lay["m_opt_param_groups"][0][key])), key
    for key in ("lr", "momentum", "alpha", "eps", "centered", "weight_decay"):
        assert co["optimizer"]["param_groups"][0][key] == lay["c_opt_param_groups"][0][key], key
    # and the reference-written optimiser dicts load into the drop-in optimisers as they are
    m_opt = net.Adam(model.parameters(), lr=1e-4, betas=(0.5, 0.999)); c_opt = net.RMSprop(cdae.parameters(), lr=1e-4, momentum=0.5)
    m_opt.load_state_dict(mck["optimizer"]); c_opt.load_state_dict(cck["optimizer"])
    assert int(m_opt.state_dict()["state"][0]["step"]) == k


def test_forward_and_generate_return_decoder_sample():
    """a9: forward() / generate() return (x_sample, decoder mean, ...) like the reference (ivae/mnist.py:300,316): the relaxed
    Bernoulli sample sigmoid(logit + log(u/(1-u) + 1e-20)) (reparam.py:111-120) and sigmoid(logit); the toy model's Gaussian
    decoder returns mu + exp(logvar/2) eps and mu (reparam.py:42-51, toy.py:725-737)."""
    model, _ = build(TINY_M, TINY_C)
    model = model.to("cuda")
    B, nz = 5, 3
    x = torch.bernoulli(torch.full((B, 24), 0.3)).cuda()
    u = torch.rand(B * nz, 24).cuda()
    xs, xm, z, loss, rec, pri = model(x, nz=nz, dec_noise=u)
    logit = model.decode_params(z.detach().reshape(B * nz, -1))[0]
    want = torch.sigmoid(logit + torch.log(u / (1. - u) + 1e-20))
    assert xs.shape == (B * nz, 24) and xm.shape == (B * nz, 24)
    assert float((xs - want).abs().max()) < 2e-6 and float((xm - torch.sigmoid(logit)).abs().max()) < 2e-6
    assert loss.requires_grad and not xs.requires_grad
    xs2, xm2, z2 = model.generate(7)
    assert xs2.shape == (7, 24) and z2.shape == (7, 8) and 0.0 <= float(xs2.min()) and float(xs2.max()) <= 1.0
    assert float((xm2 - torch.sigmoid(model.decode_params(z2)[0])).abs().max()) < 2e-6
    assert abs(float(z2.mean())) < 1.5 and not torch.equal(model.generate(7)[0], xs2)
    model.return_samples = False
    assert model(x, nz=nz)[0] is None
    toy_m = O.ModelCfg("toy", 2, 10, 64, 2, 2, "relu")
    toy, _ = build(toy_m, O.CdaeCfg("grad", 2, 2, 64, 3))
    toy = toy.to("cuda")
    xt = torch.randn(6, 2).cuda()
    e = torch.randn(6, 2).cuda()
    xs, xm, z, *_ = toy(xt, dec_noise=e)
    mu, lv = toy.decode_params(z.detach().reshape(6, -1))
    assert float((xm - mu).abs().max()) == 0.0 and float((xs - (mu + torch.exp(0.5 * lv) * e)).abs().max()) < 1e-5 * (1 + float(mu.abs().max()))


@pytest.mark.parametrize("n,batch", [(32, 9), (8, 3), (64, 2), (1, 4)])
def test_cholesky_batched_kernel(n, batch):
    """The IWAE proposal's factorisation (MultivariateNormal(mu, cov), ivae/mnist.py:397-406) as one launch, against float64 LAPACK."""
    g = torch.Generator().manual_seed(n)
    a = torch.randn(batch, n, 2 * n + 3, generator=g)
    cov = (a @ a.transpose(1, 2) / (2 * n + 2)).contiguous()
    want = torch.linalg.cholesky(cov.double())
    got = torch.empty(batch, n, n, device="cuda")
    covd = cov.cuda()
    L.check(L.lib().ardae_cholesky_batched(L.ptr(covd), batch, n, L.ptr(got), L.stream_ptr()))
    assert rel_l2(got, want) < 2e-6
    assert float(got.cpu().triu(1).abs().max()) == 0.0
    bad = -torch.eye(n).expand(batch, n, n).contiguous().cuda()
    L.check(L.lib().ardae_cholesky_batched(L.ptr(bad), batch, n, L.ptr(got), L.stream_ptr()))
    assert not torch.isfinite(got).all()                          # not positive definite: NaNs, which logprob() turns into ValueError
    with pytest.raises(ValueError):
        L.check(L.lib().ardae_cholesky_batched(L.ptr(covd), batch, 65, L.ptr(got), L.stream_ptr()))


def test_parameter_updates_through_p_data_are_seen():
    """the reference's vendored Adam writes p.data.addcdiv_(...) (utils/optim.py:106), which bumps no version counter;
    the module surface must still compute with the updated weights on the next call."""
    model, cdae = build(TINY_M, TINY_C)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    x = torch.bernoulli(torch.full((4, 24), 0.3)).cuda()
    z_a = model.encode(x, std=0).clone()
    ctx, u = z_a.detach(), torch.randn(4, 6, 8).cuda()
    g_a = cdae.glogprob(u, ctx, std=torch.zeros(4, 6, 1).cuda()).clone()
    with torch.no_grad():
        for p in list(model.parameters()) + list(cdae.parameters()):
            p.data.add_(0.05 * torch.randn_like(p))
    z_b = model.encode(x, std=0)
    g_b = cdae.glogprob(u, ctx, std=torch.zeros(4, 6, 1).cuda())
    assert float((z_b - z_a).abs().max()) > 1e-3 and float((g_b - g_a).abs().max()) > 1e-6
    # and they agree with a model that was given the same parameters through load_state_dict
    model2, cdae2 = build(TINY_M, TINY_C)
    model2.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()}); cdae2.load_state_dict({k: v.cpu() for k, v in cdae.state_dict().items()})
    model2, cdae2 = model2.to("cuda"), cdae2.to("cuda")
    assert torch.equal(model2.encode(x, std=0), z_b) and torch.equal(cdae2.glogprob(u, ctx, std=torch.zeros(4, 6, 1).cuda()), g_b)


def test_engine_refuses_ragged_or_strided_batches():
    """a ragged last batch (loaders without drop_last), a strided view or a host tensor must raise, not read out of bounds."""
    model, cdae = build(TINY_M, TINY_C)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=8), batch_size=4)
    good = torch.bernoulli(torch.full((4, 24), 0.3)).cuda()
    eng.step(good, good)
    eng.step(good.view(4, 1, 4, 6), good)                           # image-shaped batches are fine
    for bad in (good[:3], torch.cat([good, good])[:, :24][::2], good.cpu(), good.double(), torch.zeros(4, 25).cuda(),
                torch.zeros(4, 48).cuda()[:, ::2]):
        with pytest.raises((ValueError, TypeError)):
            eng.step(bad, good)
        with pytest.raises((ValueError, TypeError)):
            eng.step(good, bad)
        with pytest.raises((ValueError, TypeError)):
            eng.cdae_phase(bad)


def test_scalar_log_channel(tmp_path):
    """f-4: the scalars the reference logs every --log-interval (ivae_ardae.py:850-906) leave the step through a device ring
    buffer (one kernel at the end of the step, graph-replayable, no host synchronisation) and are read in bulk."""
    model, cdae = build(TINY_M, TINY_C)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=8, d_lr=2e-4), batch_size=4)
    log = net.ScalarLog(eng, capacity=8, path=str(tmp_path), train_mode="train")
    g = torch.Generator().manual_seed(1)
    want = []
    for t in range(5):
        xc, xv = (torch.bernoulli(torch.full((4, 24), 0.3), generator=g).cuda() for _ in range(2))
        eng.step(xc, xv)
        want.append(eng.stats())
    recs = log.drain()
    assert [r["iter"] for r in recs] == [1, 2, 3, 4, 5]
    for r, w in zip(recs, want):
        assert r["train/model/loss/step"] == pytest.approx(w["model_loss"], rel=1e-6)
        assert r["train/model/recon/step"] == pytest.approx(w["recon"], rel=1e-6)
        assert r["train/model/prior/step"] == pytest.approx(w["prior"], rel=1e-6)
        assert r["train/cdae/loss/step"] == pytest.approx(w["cdae_loss"], rel=1e-6)
        assert r["train/cdae/std/eff/mean/step"] == pytest.approx(w["std_mean"], rel=1e-5)
        assert r["train/cdae/std/eff/max/step"] == pytest.approx(w["std_max"], rel=1e-6)
        assert r["train/cdae/std/eff/min/step"] == pytest.approx(w["std_min"], rel=1e-6)
        assert r["train/cdae/std/true/mean/step"] == pytest.approx(w["std_mean"] / 1e4, rel=1e-5)
        assert r["train/model/beta/step"] == 1.0 and r["train/cdae/lr/step"] == pytest.approx(2e-4)
    assert log.drain() == []                                       # nothing new
    for t in range(11):                                            # more steps than the ring holds between two drains: the newest survive
        eng.step(xc, xv)
    recs = log.drain()
    assert [r["iter"] for r in recs] == list(range(9, 17)) and log.dropped == 3
    lines = open(tmp_path / "log.txt").read().strip().splitlines()
    assert len(lines) == 13 and lines[0].startswith("| iter 1 |") and "loss (cdae)" in lines[0]
    sc = [json.loads(l) for l in open(tmp_path / "scalars.jsonl")]
    assert len(sc) == 13 and sc[0]["iter"] == 1 and len([k for k in sc[0] if k.startswith("train/")]) == 12


def test_static_binarized_source():
    """f-4 / config #3: statically binarised data = a FIXED table of pre-drawn rows (datasets/sbmnist.py:34-60), served in shuffled
    epochs without host copies; every epoch visits every row once and the rows never change."""
    net.manual_seed(11)
    src = net.data.StaticBinarizedSource.synthetic(num_rows=1000, input_dim=784, device="cuda", seed=3)
    table = src.table.clone()
    assert set(torch.unique(table).tolist()) <= {0.0, 1.0} and 0.05 < float(table.mean()) < 0.4
    seen = []
    for _ in range(1000 // 50):
        xb = src.next_batch(50)
        assert xb.shape == (50, 784) and xb.is_contiguous()
        seen.append(src.last_indices.clone())
        assert torch.equal(xb, table[src.last_indices])
    assert torch.equal(torch.sort(torch.cat(seen)).values.cpu(), torch.arange(1000))
    first_epoch = torch.cat(seen)
    second = torch.cat([(src.next_batch(50), src.last_indices.clone())[1] for _ in range(20)])
    assert not torch.equal(first_epoch, second) and torch.equal(src.table, table)
    assert src.epoch == 1
    # from a Larochelle .amat text file (the format datasets/sbmnist.py:46-49 parses)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        rows = (np.random.RandomState(0).rand(7, 784) < 0.2).astype(np.float32)
        np.savetxt(os.path.join(d, "binarized_mnist_train.amat"), rows, fmt="%d")
        s2 = net.data.StaticBinarizedSource.from_amat(os.path.join(d, "binarized_mnist_train.amat"), device="cuda")
        assert torch.equal(s2.table.cpu(), torch.tensor(rows))


def test_engine_input_buffers_zero_copy():
    """`input_buffers()` hands out the engine's static batch buffers; a step fed through them equals a step fed with separate tensors
    (eager steps and replayed ones)."""
    mc, cc = O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus"), O.CdaeCfg("grad", 8, 8, 64, 3)
    B = 8

    def run(use_buffers):
        torch.manual_seed(0)
        model = net.MNISTIPVAE(input_dim=24, noise_dim=10, h_dim=64, num_hidden_layers=2, nonlinearity="softplus", enc_type="concat", z_dim=8)
        cdae = net.MLPGradCARDAE(input_dim=8, context_dim=8, std=1., h_dim=64, num_hidden_layers=3, nonlinearity="softplus",
                                 noise_type="gaussian", enc_ctx=True, enc_input=True)
        model.load_state_dict(O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc)))
        cdae.load_state_dict(O.init_params(O.cdae_param_spec(cc), 1))
        model, cdae = model.cuda(), cdae.cuda()
        net.manual_seed(5)
        eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=16), batch_size=B)
        g = torch.Generator().manual_seed(2)
        if use_buffers:
            (xc,), xv = eng.input_buffers()
            assert xc.shape == (B, 24) and xv.shape == (B, 24)
        for _ in range(5):
            a, b = torch.bernoulli(torch.full((B, 24), 0.3), generator=g).cuda(), torch.bernoulli(torch.full((B, 24), 0.3), generator=g).cuda()
            if use_buffers:
                xc.copy_(a); xv.copy_(b)
                eng.step(xc, xv)
            else:
                eng.step(a, b)
        torch.cuda.synchronize()
        return model.flat_params().clone(), cdae.flat_params().clone()

    m0, c0 = run(False)
    m1, c1 = run(True)
    assert torch.equal(m0, m1) and torch.equal(c0, c1)
