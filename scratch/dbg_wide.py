import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ardae_amd
from ardae_amd import _lib as L
from test_linear_gpu import pack, run_linear, d1
M, K, N = 8192, 256, 256
for rep in range(3):
    g = torch.Generator().manual_seed(M + K + N + rep)
    X = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5
    S = torch.nn.functional.softplus(torch.randn(M, N, generator=g) * 3)
    R = torch.randn(M, N, generator=g)
    v = X.double() @ W.double().T; s1 = d1("softplus", S.double())
    ref1, ref2 = v * s1, v * R.double() * (1 - s1)
    wpk = pack(W.cuda())
    Y = torch.full((M, N), float("nan"), device="cuda"); Y2 = torch.full((M, N), float("nan"), device="cuda")
    Rd = R.cuda(); Sd = S.cuda()
    run_linear(L.EPI_CHAIN, M, N, [(X.cuda(), wpk)], act=2, S=Sd, R=Rd, Y=Y, Y2=Y2)
    for nm, got, ref in (("Y", Y, ref1), ("Y2", Y2, ref2)):
        err = (got.cpu().double() - ref).abs()
        bad = err > 1e-3 * ref.abs().max()
        print("rep", rep, nm, ": bad fraction %.4f" % bad.float().mean().item(), "nan", torch.isnan(got).sum().item())
        if bad.any():
            rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
            print("  bad rows: count", len(rows), "first", rows[:8].tolist(), "mod64 set", sorted(set((rows % 64).tolist())))
            print("  bad cols: count", len(cols), "set mod 64", sorted(set((cols % 64).tolist()))[:64], "col blocks", sorted(set((cols // 64).tolist())))
            r, c = bad.nonzero()[0].tolist()
            want_R = (ref[r, c] / (v[r, c] * (1 - s1[r, c]))).item() if nm == "Y2" else 0
            print("  sample", r, c, "got", got[r, c].item(), "ref", ref[r, c].item(), " implied R", got[r, c].item() / (v[r, c] * (1 - s1[r, c])).item(), "true R", want_R)
            # is the implied R equal to R at some other position?
            if nm == "Y2":
                impl = got[r, c].item() / (v[r, c] * (1 - s1[r, c])).item()
                d = (R - impl).abs()
                k = d.argmin().item()
                print("  nearest R element to implied:", divmod(k, N), "diff", d.flatten()[k].item())
