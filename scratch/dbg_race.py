"""Which buffer diverges first between two identical engines (same seed, same batches)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import ardae_amd as net
from oracle import ardae_oracle as O
from test_engine_gpu import build

mc = O.ModelCfg("mnist", 24, 10, 64, 8, 2, "softplus")
cc = O.CdaeCfg("grad", 8, 8, 64, 3)
B = 4
pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc))
pc = O.init_params(O.cdae_param_spec(cc), 1)
gen = torch.Generator().manual_seed(21)
xs = [torch.bernoulli(torch.full((B, 24), 0.3), generator=gen).cuda() for _ in range(3)]
NUPD = int(os.environ.get("NUPD", "2"))
names = ["z0", "latent", "noise_s", "xi", "eps", "xbar", "sigma", "std_b", "noise_v", "zv", "z0v", "u", "g", "loss_c", "losses_m", "grads_c", "grads_m"]

def run(graph):
    net.manual_seed(5)
    model, cdae = build(mc, cc)
    model.load_state_dict(pm); cdae.load_state_dict(pc)
    model, cdae = model.to("cuda"), cdae.to("cuda")
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=8, num_cdae_updates=NUPD), batch_size=B, graph=graph)
    snaps = []
    for t in range(int(os.environ.get("STEPS", "6"))):
        eng.step([xs[(t + i) % 3] for i in range(NUPD)], xs[(t + 2) % 3])
        torch.cuda.synchronize()
        s = {n: getattr(eng, n).clone() for n in names}
        s["pm"] = model.flat_params().clone(); s["pc"] = cdae.flat_params().clone()
        snaps.append(s)
    return snaps

for graph in (False, True):
    for rep in range(3):
        a, b = run(graph), run(graph)
        bad = [(t, n) for t in range(len(a)) for n in a[t] if not torch.equal(a[t][n], b[t][n])]
        print(f"graph={graph} rep {rep}: first differences: {bad[:8]}", flush=True)
