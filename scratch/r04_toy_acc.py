"""Probe (round 4): accuracy of the toy configuration's pieces against the float64 oracle - run once per library build (ARDAE_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import ardae_amd as net
from oracle import ardae_oracle as O
import test_engine_gpu as T
import test_cdae_gpu as C
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
mc, cc = O.ModelCfg("toy", 2, 10, 256, 2, 2, "relu"), O.CdaeCfg("grad", 2, 2, 256, 3)
B, NZ = 32, 256
pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc)); pc = O.init_params(O.cdae_param_spec(cc), 1)
g = torch.Generator().manual_seed(11)
x = torch.randn(B, 2, generator=g) * 0.3 + torch.randint(-2, 3, (B, 2), generator=g).float() * 2
model, cdae = T.build(mc, cc); model.load_state_dict(pm); cdae.load_state_dict(pc); model, cdae = model.to("cuda"), cdae.to("cuda")
noise = torch.randn(B * NZ, mc.noise_dim, generator=g)
z = model.forward_hidden(x.cuda(), nz=NZ, noise=noise.cuda())
z0 = model.encode(x.cuda(), std=0)
pm64 = {k: v.double() for k, v in pm.items()}
zr = O.encode(mc, pm64, x.double(), noise.double(), NZ); z0r = O.encode(mc, pm64, x.double(), torch.zeros(B, mc.noise_dim).double(), 1)
print("sampler z  rel", rel(z.reshape(-1, 2), zr.reshape(-1, 2)), " z0 rel", rel(z0.reshape(-1, 2), z0r.reshape(-1, 2)))
# cDAE loss + grads at this shape against float64
tc = O.TrainCfg(nz_cdae=NZ)
u, std = O.latent_stats(zr.float(), z0r.float(), tc.std_scale, tc.delta)
sig = (std * torch.randn(B, NZ, 1, generator=g)).reshape(-1); eps = torch.randn(B * NZ, 2, generator=g)
xbar = (u.reshape(-1, 2) + sig[:, None] * eps).contiguous()
H = C.CdaeHarness(cc, C.flat(pc, O.cdae_param_spec(cc)))
loss, grads, score = H.loss_grads(xbar, sig, eps, z0r.float().reshape(B, 2), B, NZ)
l64, g64, s64 = C.oracle64_grads(cc, pc, xbar, sig, eps, z0r.float().reshape(B, 2), NZ)
gs = C.split_flat(grads, O.cdae_param_spec(cc))
print("cdae loss rel", abs(float(loss) - float(l64)) / abs(float(l64)), " score rel", rel(score, s64))
for n, _ in O.cdae_param_spec(cc):
    if g64[n] is not None: print("  %-26s %.2e" % (n, rel(gs[n], g64[n])))
