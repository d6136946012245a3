"""Probe: engine trained with the ORACLE's noise (injected) on the oracle's batches at config #2 widths - how far does its IWAE-64 drift from
the oracle's own value for the same seed?  (oracle numbers: oracle/gen_quality_golden.py logs)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ardae_amd as net
from oracle import ardae_oracle as O
from oracle.gen_quality_golden import MC, CC, B, NZ, K, batches, eval_set

lr = float(os.environ.get("LR", "3e-4")); steps = int(os.environ.get("STEPS", "800")); every = 200
torch.set_num_threads(16)
pm0 = O.init_params(O.model_param_spec(MC), 0, O.model_init_special(MC)); pc0 = O.init_params(O.cdae_param_spec(CC), 1)
x_eval, enc_noise, prop_noise = eval_set()
bs = batches(steps)
tc = O.TrainCfg(nz_cdae=NZ, m_lr=lr, d_lr=lr)
for seed in [int(s) for s in os.environ.get("SEEDS", "2024 1").split()]:
    for mode in ("shared", "own"):
        model = net.MNISTIPVAE(input_dim=784, noise_dim=100, h_dim=256, num_hidden_layers=2, nonlinearity="softplus", enc_type="concat", z_dim=32)
        cdae = net.MLPGradCARDAE(input_dim=32, context_dim=32, std=1., h_dim=256, num_hidden_layers=3, nonlinearity="softplus", noise_type="gaussian", enc_ctx=True, enc_input=True)
        model.load_state_dict(pm0); cdae.load_state_dict(pc0)
        model, cdae = model.to("cuda"), cdae.to("cuda")
        net.manual_seed(seed)
        eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=NZ, m_lr=lr, d_lr=lr), batch_size=B)
        gn = torch.Generator().manual_seed(seed)
        t0 = time.time(); out = []
        for t, (x1, x2) in enumerate(bs, 1):
            if mode == "shared":
                noise = O.draw_step_noise(MC, tc, B, gn)
                eng.step(x1.cuda(), x2.cuda(), noise={k: v.cuda().contiguous() for k, v in noise.items()})
            else:
                eng.step(x1.cuda(), x2.cuda())
            if t % every == 0:
                pm = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
                out.append(round(float(O.iwae_logprob(MC, pm, x_eval, K, enc_noise, prop_noise)), 3))
        print(f"lr {lr} seed {seed} {mode:6s} noise: IWAE-64 at {list(range(every, steps + 1, every))}: {out}  ({time.time() - t0:.0f} s)", flush=True)
