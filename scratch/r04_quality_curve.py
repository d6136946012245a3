"""Probe (round 4): the engine's IWAE-64 curve at config #2 widths (32 images x 256 rows) over a LONGER training, several noise seeds,
raw weights and an exponential moving average of them - where does the curve flatten, and how large is the seed-to-seed spread there?
(What a quality gate at these widths can resolve; oracle/gen_quality_golden.py is the CPU side.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ardae_amd as net
from oracle import ardae_oracle as O
from oracle.gen_quality_golden import MC, CC, B, NZ, K, batches, eval_set

lr = float(os.environ.get("LR", "3e-4")); steps = int(os.environ.get("STEPS", "2400")); every = int(os.environ.get("EVERY", "100"))
decay = float(os.environ.get("EMA", "0.99"))
torch.set_num_threads(16)
pm0 = O.init_params(O.model_param_spec(MC), 0, O.model_init_special(MC)); pc0 = O.init_params(O.cdae_param_spec(CC), 1)
x_eval, enc_noise, prop_noise = eval_set()
bs = batches(steps)
rows, rows_ema = [], []
for seed in [int(s) for s in os.environ.get("SEEDS", "31337 11 12 13 14 15").split()]:
    model = net.MNISTIPVAE(input_dim=784, noise_dim=100, h_dim=256, num_hidden_layers=2, nonlinearity="softplus", enc_type="concat", z_dim=32)
    cdae = net.MLPGradCARDAE(input_dim=32, context_dim=32, std=1., h_dim=256, num_hidden_layers=3, nonlinearity="softplus", noise_type="gaussian", enc_ctx=True, enc_input=True)
    model.load_state_dict(pm0); cdae.load_state_dict(pc0)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    net.manual_seed(seed)
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=NZ, m_lr=lr, d_lr=lr), batch_size=B)
    ema = model.flat_params().clone()
    keys = list(model.state_dict().keys())
    t0 = time.time(); out, out_e = [], []
    for t, (x1, x2) in enumerate(bs, 1):
        eng.step(x1.cuda(), x2.cuda())
        ema.lerp_(model.flat_params(), 1.0 - decay)
        if t % every == 0:
            pm = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            out.append(round(float(O.iwae_logprob(MC, pm, x_eval, K, enc_noise, prop_noise)), 2))
            # the EMA weights through the same state_dict layout (parameters are views of the flat buffer, in named_parameters order)
            pe, off = {}, 0
            for k in keys:
                n = pm[k].numel(); pe[k] = ema[off:off + n].view_as(pm[k]).cpu().clone(); off += n
            out_e.append(round(float(O.iwae_logprob(MC, pe, x_eval, K, enc_noise, prop_noise)), 2))
    rows.append(out); rows_ema.append(out_e)
    print(f"seed {seed} raw {out}\n          ema {out_e}  ({time.time() - t0:.0f} s)", flush=True)
r, e = np.array(rows), np.array(rows_ema)
print("marks      ", list(range(every, steps + 1, every)))
print("raw mean   ", np.round(r.mean(0), 2).tolist()); print("raw sd     ", np.round(r.std(0, ddof=1), 2).tolist())
print("ema mean   ", np.round(e.mean(0), 2).tolist()); print("ema sd     ", np.round(e.std(0, ddof=1), 2).tolist())
