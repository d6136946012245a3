import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import ardae_oracle as O
import test_cdae_gpu as T
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in ["tiny_mnist_grad", "cfg2"]:
    if name == "cfg2":
        fx = T.load(gd, "cfg2_b8_nz16")
        mc = O.ModelCfg("mnist", 784, 100, 256, 32, 2, "softplus"); cc = O.CdaeCfg("grad", 32, 32, 256, 3); nz = 16
        pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc)); pc = O.init_params(O.cdae_param_spec(cc), 1)
    else:
        mc, cc, nz = T.CASES[name]; fx = T.load(gd, name)
        pm = {n: torch.tensor(fx["pm/" + n]) for n, _ in O.model_param_spec(mc)}
        pc = {n: torch.tensor(fx["pc/" + n]) for n, _ in O.cdae_param_spec(cc)}
    tc = O.TrainCfg(nz_cdae=nz)
    x = torch.tensor(fx["s0/x_cdae"]); noise = {k: torch.tensor(fx["s0/noise/" + k]) for k in ("sampler", "sigma", "eps", "vae")}
    B = x.size(0)
    z0, _, xbar, sigma = T.prep_inputs(mc, tc, pm, x, noise)
    print(name, "xbar absmax", float(xbar.abs().max()), "sigma absmax", float(sigma.abs().max()))
    hn = T.CdaeHarness(cc, T.flat(pc, O.cdae_param_spec(cc)))
    loss, grads, score = hn.loss_grads(xbar, sigma, noise["eps"], z0, B, nz)
    g = T.split_flat(grads, O.cdae_param_spec(cc))
    _, g32, _ = O.cdae_update_grads(mc, cc, tc, pm, pc, x, noise)
    l64, g64, sc64 = T.oracle64_grads(cc, pc, xbar, sigma, noise["eps"], z0, nz)
    print(" loss hip", float(loss), "f64", float(l64), " score hip-vs-64", T.rel_l2(score, sc64))
    for n, _ in O.cdae_param_spec(cc):
        if g64[n] is None: continue
        print(f" {n:30s} hip-vs-64 {T.rel_l2(g[n], g64[n]):.2e}  ref32-vs-64 {T.rel_l2(g32[n], g64[n]):.2e}  hip-vs-ref32 {T.rel_l2(g[n], g32[n]):.2e}")
