"""Spread of the IWAE-64 gate over noise seeds / engine modes (scratch; see tests/test_training_quality_gpu.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ardae_amd as net
from oracle import ardae_oracle as O
import test_training_quality_gpu as T

MC, CC, B, NZ, STEPS, K, LR = T.MC, T.CC, T.B, T.NZ, T.STEPS, T.K, T.LR
pm0 = O.init_params(O.model_param_spec(MC), 0, O.model_init_special(MC))
pc0 = O.init_params(O.cdae_param_spec(CC), 1)
gen = torch.Generator().manual_seed(123)
batches = [(T._data(gen, B), T._data(gen, B)) for _ in range(STEPS)]
x_eval = T._data(torch.Generator().manual_seed(999), 256)
ge = torch.Generator().manual_seed(7)
enc_noise = torch.randn(256, K, MC.noise_dim, generator=ge)
prop_noise = torch.randn(256, K, MC.z_dim, generator=ge)
torch.set_num_threads(4)

def oracle(seed):
    tc = O.TrainCfg(nz_cdae=NZ, m_lr=LR, d_lr=LR)
    pm = {k: v.clone() for k, v in pm0.items()}; pc = {k: v.clone() for k, v in pc0.items()}
    st_m, st_c = {}, {}
    gn = torch.Generator().manual_seed(seed)
    for x1, x2 in batches:
        r = O.train_step(MC, CC, tc, pm, pc, st_m, st_c, x1, x2, O.draw_step_noise(MC, tc, B, gn))
    return T._iwae(pm, x_eval, enc_noise, prop_noise), {k: float(v) for k, v in r.items()}

def hip(seed, graph=True, inject=False):
    dev = torch.device("cuda", 0)
    model = net.MNISTIPVAE(input_dim=MC.input_dim, noise_dim=MC.noise_dim, h_dim=MC.h_dim, num_hidden_layers=MC.n_layers,
                           nonlinearity=MC.nonlin, enc_type="concat", z_dim=MC.z_dim)
    cdae = net.MLPGradCARDAE(input_dim=CC.input_dim, context_dim=CC.context_dim, std=1., h_dim=CC.h_dim, num_hidden_layers=CC.n_layers,
                             nonlinearity=CC.nonlin, noise_type="gaussian", enc_ctx=True, enc_input=True)
    model.load_state_dict(pm0); cdae.load_state_dict(pc0)
    model, cdae = model.to(dev), cdae.to(dev)
    net.manual_seed(seed)
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=NZ, m_lr=LR, d_lr=LR), batch_size=B, graph=graph)
    gn = torch.Generator().manual_seed(seed)
    tc = O.TrainCfg(nz_cdae=NZ, m_lr=LR, d_lr=LR)
    for x1, x2 in batches:
        nz = None
        if inject:   # the oracle's own torch draws, injected: isolates the RNG from the arithmetic
            nz = {k: v.to(dev).contiguous() for k, v in O.draw_step_noise(MC, tc, B, gn).items()}
        eng.step(x1.to(dev), x2.to(dev), noise=nz)
    torch.cuda.synchronize()
    pm = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    return T._iwae(pm, x_eval, enc_noise, prop_noise), eng.stats()

for s in (2024, 1, 2):
    print("oracle", s, oracle(s)[0], flush=True)
for s in (31337, 1, 2):
    print("hip graph", s, hip(s, True)[0], flush=True)
