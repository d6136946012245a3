"""Probe for the production-shape trajectory test: free-running engine vs oracle over 20 steps (same images, same injected noise), and
teacher-forced single-step update errors at late steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import ardae_amd as net
from oracle import ardae_oracle as O
from test_engine_gpu import build

kind = sys.argv[1] if len(sys.argv) > 1 else "mnist"
if kind == "mnist":
    mc = O.ModelCfg("mnist", 784, 100, 256, 32, 2, "softplus"); cc = O.CdaeCfg("grad", 32, 32, 256, 3); B = 32; p = 0.2
else:
    mc = O.ModelCfg("toy", 2, 10, 256, 2, 2, "relu"); cc = O.CdaeCfg("grad", 2, 2, 256, 3); B = 32
NZ, STEPS = 256, int(os.environ.get("STEPS", "20"))
tc = O.TrainCfg(nz_cdae=NZ)
torch.set_num_threads(16)
pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc))
pc = O.init_params(O.cdae_param_spec(cc), 1)
gen = torch.Generator().manual_seed(11)
def batch():
    if kind == "mnist":
        return torch.bernoulli(torch.full((B, 784), 0.2), generator=gen)
    return torch.randn(B, 2, generator=gen) * 0.3 + torch.randint(-2, 3, (B, 2), generator=gen).float() * 2
data = [(batch(), batch(), O.draw_step_noise(mc, tc, B, gen)) for _ in range(STEPS)]
model, cdae = build(mc, cc)
model.load_state_dict(pm); cdae.load_state_dict(pc)
model, cdae = model.to("cuda"), cdae.to("cuda")
eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=NZ), batch_size=B)
rm, rc = {k: v.clone() for k, v in pm.items()}, {k: v.clone() for k, v in pc.items()}
st_m, st_c = {}, {}
mnames = [n for n, _ in O.model_param_spec(mc)]; cnames = [n for n, _ in O.cdae_param_spec(cc)]
def rel_l2(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
def opt_state(ck, names, keys):
    out = {}
    for i, n in enumerate(names):
        if i in ck["optimizer"]["state"]:
            s = ck["optimizer"]["state"][i]
            out[n] = {"step": int(s["step"]), **{k: s[k].detach().cpu().clone() for k in keys}}
    return out
for t, (x1, x2, noise) in enumerate(data):
    # teacher-forced copy of the engine's state for a single oracle step
    mck, cck = eng.model_checkpoint(), eng.cdae_checkpoint()
    tm = {n: mck["state_dict"][n].detach().cpu().clone() for n in mnames}; tcd = {n: cck["state_dict"][n].detach().cpu().clone() for n in cnames}
    tsm = opt_state(mck, mnames, ("exp_avg", "exp_avg_sq")); tsc = opt_state(cck, cnames, ("square_avg", "momentum_buffer"))
    bm = torch.cat([tm[n].reshape(-1) for n in mnames]); bc = torch.cat([tcd[n].reshape(-1) for n in cnames])
    eng.step(x1.cuda(), x2.cuda(), noise={k: v.cuda().contiguous() for k, v in noise.items()})
    got = eng.stats()
    ref = O.train_step(mc, cc, tc, rm, rc, st_m, st_c, x1, x2, noise)                       # free-running oracle
    line = f"step {t:2d}: cdae_loss {got['cdae_loss']:.6f} vs {float(ref['cdae_loss']):.6f} (rel {abs(got['cdae_loss']-float(ref['cdae_loss']))/abs(float(ref['cdae_loss'])):.2e})  model_loss rel {abs(got['model_loss']-float(ref['model_loss']))/abs(float(ref['model_loss'])):.2e}"
    if t in (0, 1, 5, 10, 15, STEPS - 1):
        tf = O.train_step(mc, cc, tc, tm, tcd, tsm, tsc, x1, x2, noise)                     # one oracle step from the engine's state
        am = model.flat_params().cpu(); ac = cdae.flat_params().cpu()
        fm = torch.cat([tm[n].reshape(-1) for n in mnames]); fc = torch.cat([tcd[n].reshape(-1) for n in cnames])
        um, uc = rel_l2(am - bm, fm - bm), rel_l2((ac - bc)[:-1], (fc - bc)[:-1])
        errc = ((ac - bc) - (fc - bc)).abs()[:-1] / ((fc - bc).abs()[:-1] + 1e-12)
        line += f" | forced: loss rel {abs(got['cdae_loss']-float(tf['cdae_loss']))/abs(float(tf['cdae_loss'])):.2e} / {abs(got['model_loss']-float(tf['model_loss']))/abs(float(tf['model_loss'])):.2e}  update relL2 model {um:.2e} cdae {uc:.2e} (cdae median {float(errc.median()):.1e}, frac>1e-2 {float((errc>1e-2).double().mean()):.3f})"
    print(line, flush=True)
fm = torch.cat([rm[n].reshape(-1) for n in mnames]); fc = torch.cat([rc[n].reshape(-1) for n in cnames])
p0m = torch.cat([pm[n].reshape(-1) for n in mnames]); p0c = torch.cat([pc[n].reshape(-1) for n in cnames])
print("free-run after", STEPS, "steps: total movement relL2 model", rel_l2(model.flat_params().cpu() - p0m, fm - p0m), "cdae", rel_l2((cdae.flat_params().cpu() - p0c)[:-1], (fc - p0c)[:-1]))
