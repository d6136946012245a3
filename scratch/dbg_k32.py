import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import ardae_amd
from ardae_amd import _lib as L
from test_linear_gpu import pack, run_linear, relerr, d1
M, K, Nout = int(sys.argv[1]), 32, 256
epi = sys.argv[2]
g = torch.Generator().manual_seed(1)
X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / K ** 0.5
S = torch.nn.functional.softplus(torch.randn(M, Nout, generator=g) * 3); Q = torch.randn(M, Nout, generator=g)
v = X.double() @ W.double().T; s1 = d1("softplus", S.double())
wpk = pack(W.cuda())
Y = torch.full((M, Nout), float("nan"), device="cuda")
if epi == "dact":
    run_linear(L.EPI_DACT, M, Nout, [(X.cuda(), wpk)], act=2, S=S.cuda(), Y=Y); ref = v * s1
else:
    run_linear(L.EPI_DACT, M, Nout, [(X.cuda(), wpk)], act=2, S=S.cuda(), Q=Q.cuda(), Y=Y); ref = v * s1 + Q.double()
torch.cuda.synchronize()
print(M, epi, "relerr", relerr(Y, ref))
