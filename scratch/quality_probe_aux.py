"""Seed spread of the aux-model IWAE gate (scratch; see tests/test_training_quality_gpu.py)."""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from oracle import ardae_oracle as O
import test_training_quality_gpu as T
mc, cc, B, NZ, STEPS, K, LR = T.MC_AUX, T.CC_AUX, T.B, T.NZ, T.STEPS, T.K, T.LR
pm0 = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc)); pc0 = O.init_params(O.cdae_param_spec(cc), 1)
gen = torch.Generator().manual_seed(123)
batches = [(T._data(gen, B), T._data(gen, B)) for _ in range(STEPS)]
x_eval = T._data(torch.Generator().manual_seed(999), 256)
ge = torch.Generator().manual_seed(7)
enc = (torch.randn(256, K, mc.noise_dim, generator=ge), torch.randn(256, K, mc.z_dim, generator=ge)); prop = torch.randn(256, K, mc.z_dim, generator=ge)
torch.set_num_threads(4)
for seed in (2024, 1, 2):
    tc = O.TrainCfg(nz_cdae=NZ, m_lr=LR, d_lr=LR, ctx_type="hidden1a")
    pm = {k: v.clone() for k, v in pm0.items()}; pc = {k: v.clone() for k, v in pc0.items()}
    gn = torch.Generator().manual_seed(seed); st_m, st_c = {}, {}
    for x1, x2 in batches:
        O.train_step(mc, cc, tc, pm, pc, st_m, st_c, x1, x2, O.draw_step_noise(mc, tc, B, gn))
    print("oracle", seed, T._iwae(pm, x_eval, enc, prop, mc), flush=True)
for seed in (31337, 1, 2):
    model, eng = T.build_engine(mc, cc, pm0, pc0, seed)
    for x1, x2 in batches:
        eng.step(x1.cuda(), x2.cuda())
    torch.cuda.synchronize()
    print("hip", seed, T._iwae({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, x_eval, enc, prop, mc), flush=True)
