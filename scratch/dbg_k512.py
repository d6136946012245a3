import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import ardae_amd
from ardae_amd import _lib as L
from test_linear_gpu import pack, run_linear, relerr, d1
for (M, K, Nout) in [(8192, 512, 512), (8192, 256, 256)]:
    g = torch.Generator().manual_seed(1)
    X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / K ** 0.5
    S = torch.nn.functional.softplus(torch.randn(M, Nout, generator=g) * 3); Q = torch.randn(M, Nout, generator=g)
    v = X.double() @ W.double().T; s1 = d1("softplus", S.double())
    wpk = pack(W.cuda())
    for inplace in (False, True):
        Qd = Q.cuda(); Y = Qd if inplace else torch.full((M, Nout), float("nan"), device="cuda")
        run_linear(L.EPI_DACT, M, Nout, [(X.cuda(), wpk)], act=2, S=S.cuda(), Q=Qd, Y=Y)
        err = (Y.cpu().double() - (v * s1 + Q.double())).abs()
        bad = (err > 1e-3).nonzero()
        print(M, K, Nout, "inplace" if inplace else "separate", "relerr", relerr(Y, v * s1 + Q.double()), "bad elements", bad.shape[0],
              "rows", sorted(set((bad[:, 0] % 64).tolist()))[:20], "cols", sorted(set((bad[:, 1] % 128).tolist()))[:40], "row tiles", sorted(set((bad[:, 0] // 64).tolist()))[:10])
