"""rms / max error of the per-image kernels against float64 (run with and without ARDAE_SMALL16_MAX_BLOCKS=0 and compare)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_linear_gpu import run_linear, pack, d1
from ardae_amd import _lib as L
for (M, K, Nout) in [(32, 784, 256), (32, 256, 784), (32, 256, 256), (32, 32, 256), (32, 256, 32), (64, 256, 256)]:
    g = torch.Generator().manual_seed(M + K + Nout)
    X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / K ** 0.5
    S = torch.nn.functional.softplus(torch.randn(M, Nout, generator=g) * 3)
    Q = torch.randn(M, Nout, generator=g); b = torch.randn(Nout, generator=g)
    v = X.double() @ W.double().T
    wpk = pack(W.cuda())
    out = []
    Y = torch.full((M, Nout), float("nan"), device="cuda")
    run_linear(L.EPI_ACT, M, Nout, [(X.cuda(), wpk)], act=0, bias=b.cuda(), Y=Y)
    e = (Y.double().cpu() - (v + b.double())); out.append(("lin", float(e.pow(2).mean().sqrt()), float(e.abs().max())))
    run_linear(L.EPI_ACT, M, Nout, [(X.cuda(), wpk)], act=2, bias=b.cuda(), Y=Y)
    e = (Y.double().cpu() - torch.nn.functional.softplus(v + b.double())); out.append(("sp", float(e.pow(2).mean().sqrt()), float(e.abs().max())))
    run_linear(L.EPI_DACT, M, Nout, [(X.cuda(), wpk)], act=2, S=S.cuda(), Q=Q.cuda(), Y=Y)
    e = (Y.double().cpu() - (v * d1("softplus", S.double()) + Q.double())); out.append(("dactq", float(e.pow(2).mean().sqrt()), float(e.abs().max())))
    print(M, K, Nout, "  ".join(f"{n}: rms {r:.2e} max {m:.2e}" for n, r, m in out))
