"""BASELINE configs #4 and #5 at WORKLOAD shape (the widths the configs name, rows >= 8192 so that every N-row launch runs on
the production kernels, the 3L+1 / 2L+2 weight-gradient problems are grouped the way the full shard groups them and the
workspace arithmetic is the full-size one).

    #4  ConvIPVAE (28x28x1, z 32, noise 100) + mlp-grad cDAE h 512, L 4, nz_cdae 512     (1024 x 512 rows over 4 GPUs)
    #5  MNISTIPVAE(input_dim=3072) z 32 h 256 + mlp-res cDAE h 1024, L 6, nz_cdae 1024   (2048 x 1024 rows over 8 GPUs)

Per config: (i) the cDAE update (`ardae_cdae_loss_grads`) against the oracle in fp32 and float64, (ii) one whole engine
step (cDAE update + VAE update) against the oracle run live on the CPU, (iii) at the config's FULL per-GPU shard
(256 images x 512 / 1024 samples), which no CPU oracle finishes in test time, the additivity property: the update of the
shard equals the average of the updates of its four quarter shards (rows belong to one image; losses are means).
Reference: models/graddae/mlp.py:400-444, models/resdae/mlp.py:344-381, ivae_ardae.py:707-846.
Tolerances as in test_cdae_gpu.py / test_engine_gpu.py (losses 1e-4 relative, north star).
"""
import pytest
import torch

import ardae_amd as net
from oracle import ardae_oracle as O
from test_cdae_gpu import CdaeHarness, flat, rel_l2, split_flat, test_cdae_loss_grads_vs_oracle as cdae_vs_oracle
from test_engine_gpu import assert_update_close, build, rel

pytestmark = pytest.mark.gpu

CFG4_CDAE = ("grad", 32, 512, 4)     # kind, z, h, L
CFG5_CDAE = ("res", 32, 1024, 6)


@pytest.mark.parametrize("cfg,B,S", [(CFG4_CDAE, 16, 512), (CFG5_CDAE, 8, 1024)], ids=["cfg4", "cfg5"])
def test_cdae_update_workload_shape_vs_oracle(cfg, B, S):
    """8192 rows at the config's own nz_cdae and widths: loss, score and every gradient tensor against the fp32 and float64 oracle."""
    kind, z, h, L = cfg
    cdae_vs_oracle(kind, B, S, z, h, L)


def _step_vs_oracle(mc, cc, B, nz, p_pix):
    tc = O.TrainCfg(nz_cdae=nz)
    pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc))
    pc = O.init_params(O.cdae_param_spec(cc), 1)
    gen = torch.Generator().manual_seed(23)
    x1 = torch.bernoulli(torch.full((B, mc.input_dim), p_pix), generator=gen)
    x2 = torch.bernoulli(torch.full((B, mc.input_dim), p_pix), generator=gen)
    noise = O.draw_step_noise(mc, tc, B, gen)
    model, cdae = build(mc, cc)
    model.load_state_dict(pm); cdae.load_state_dict(pc)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    before_c, before_m = cdae.flat_params().clone().cpu(), model.flat_params().clone().cpu()
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=nz), batch_size=B)
    eng.step(x1.cuda(), x2.cuda(), noise={k: v.cuda().contiguous() for k, v in noise.items()})
    got = eng.stats()
    rm, rc = {k: v.clone() for k, v in pm.items()}, {k: v.clone() for k, v in pc.items()}
    ref = O.train_step(mc, cc, tc, rm, rc, {}, {}, x1, x2, noise)
    for k in ("cdae_loss", "model_loss"):
        assert rel(got[k], ref[k]) < 1e-4, (k, got[k], float(ref[k]))
    for k in ("recon", "prior"):
        assert rel(got[k], ref[k]) < 2e-5, (k, got[k], float(ref[k]))
    ref_c = torch.cat([rc[n].reshape(-1) for n, _ in O.cdae_param_spec(cc)])
    ref_m = torch.cat([rm[n].reshape(-1) for n, _ in O.model_param_spec(mc)])
    assert_update_close(cdae.flat_params().cpu(), before_c, ref_c, "cdae update")
    assert_update_close(model.flat_params().cpu(), before_m, ref_m, "model update")


def test_engine_step_cfg4_vs_oracle():
    """Config #4: ConvIPVAE + mlp-grad h 512 L 4, 16 images x 512 samples = 8192 rows, one whole step against the live oracle."""
    _step_vs_oracle(O.ModelCfg("conv", 784, 100, 800, 32, 1, "softplus"), O.CdaeCfg("grad", 32, 32, 512, 4), 16, 512, 0.2)


def test_engine_step_cfg5_vs_oracle():
    """Config #5: MNISTIPVAE(input_dim=3072) + mlp-res h 1024 L 6, 8 images x 1024 samples = 8192 rows, one whole step."""
    _step_vs_oracle(O.ModelCfg("mnist", 3072, 100, 256, 32, 2, "softplus"), O.CdaeCfg("res", 32, 32, 1024, 6), 8, 1024, 0.5)


@pytest.mark.parametrize("cfg,B,S", [(CFG4_CDAE, 256, 512), (CFG5_CDAE, 256, 1024)], ids=["cfg4", "cfg5"])
def test_cdae_full_shard_additivity(cfg, B, S):
    """The config's full per-GPU shard (256 images; 131072 / 262144 rows; 8.6 / 51.6 GB of workspace): the update of the shard
    must equal the average of the updates of its four 64-image quarters, and its score rows must equal the quarters' (row-local)."""
    kind, z, h, Ln = cfg
    cc = O.CdaeCfg(kind, z, z, h, Ln)
    pc = O.init_params(O.cdae_param_spec(cc), 5)
    g = torch.Generator().manual_seed(78)
    N = B * S
    xbar = torch.randn(N, z, generator=g) * 2
    sigma = torch.randn(N, generator=g) * 0.3
    eps = torch.randn(N, z, generator=g)
    ctx = torch.randn(B, z, generator=g)
    hn = CdaeHarness(cc, flat(pc, O.cdae_param_spec(cc)))
    loss, grads, score = hn.loss_grads(xbar, sigma, eps, ctx, B, S)
    n_used = grads.numel() - (1 if kind == "grad" else 0)          # neglogprob.fc.bias: untouched (NaN sentinel)
    assert torch.isfinite(grads[:n_used]).all()
    parts, Bs = 4, B // 4
    acc_l, acc_g = 0.0, torch.zeros(n_used, dtype=torch.float64)
    for i in range(parts):
        r = slice(i * Bs * S, (i + 1) * Bs * S)
        l_i, g_i, sc_i = hn.loss_grads(xbar[r], sigma[r], eps[r], ctx[i * Bs:(i + 1) * Bs], Bs, S)
        acc_l += float(l_i) / parts
        acc_g += g_i[:n_used].double() / parts
        # row-local N-row kernels give the same bits; the per-image context layers run on 16 x 16 blocks for 64 images and on 32 x 32
        # blocks for 256 (linear_small.hip): another order of the fp32 sums over k (measured 1.2e-6)
        assert rel_l2(score[r], sc_i) < 5e-6
    assert abs(float(loss) - acc_l) <= 1e-5 * abs(acc_l)
    spec = O.cdae_param_spec(cc)
    full, shard = split_flat(grads, spec), split_flat(torch.cat([acc_g.float(), torch.zeros(grads.numel() - n_used)]), spec)
    for n, _ in spec[:-1] if kind == "grad" else spec:
        assert rel_l2(full[n], shard[n]) < 2e-4, n                  # fp32 sums over the rows in two different orders
