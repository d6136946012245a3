import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import ardae_amd
from ardae_amd import _lib as L
from test_linear_gpu import pack
M, K, N = 131072, 256, 256
epi = int(os.environ.get("EPI", "2"))
X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 16
S = torch.nn.functional.softplus(torch.randn(M, N, device="cuda")); Q = torch.randn(M, N, device="cuda"); R = torch.randn(M, N, device="cuda")
Y = torch.empty(M, N, device="cuda"); Y2 = torch.empty(M, N, device="cuda"); b = torch.randn(N, device="cuda")
wp = pack(W)
G = int(os.environ.get("ARDAE_WIDE_GRID", "768"))
st = torch.zeros(G * 4 * 4 * 4, dtype=torch.int64, device="cuda")
a = L.LinearArgs(); a.M, a.Nout, a.nsrc = M, N, 1
a.src[0].x = X.data_ptr(); a.src[0].ld = K; a.src[0].K = K; a.src[0].wp = wp.data_ptr()
a.act = 2; a.Y = Y.data_ptr(); a.ldY = N
if epi == 0: a.bias = b.data_ptr()
if epi in (1, 2): a.S = S.data_ptr(); a.ldS = N
if epi == 1: a.Q = Q.data_ptr(); a.ldQ = N
if epi == 2: a.R = R.data_ptr(); a.ldR = N; a.Y2 = Y2.data_ptr(); a.ldY2 = N
a.tile_loss = st.data_ptr()
for _ in range(3): L.check(L.lib().ardae_linear(ctypes.byref(a), epi, L.stream_ptr()))
torch.cuda.synchronize()
t = st.cpu().numpy().reshape(G, 4, 4, 4).astype(np.float64)   # block, iter, wave, stamp
valid = t[..., 0] > 0
base = t[..., 0][valid].min()
d = np.where(valid[..., None], t - base, np.nan)
print("span:", np.nanmax(d[..., 3]))
for nm, a_, b_ in (("prologue", 0, 1), ("kloop", 1, 2), ("epilogue", 2, 3), ("tile", 0, 3)):
    x = (d[..., b_] - d[..., a_])
    x = x[~np.isnan(x)]
    print(f"{nm:9s} mean {x.mean():9.0f}  p10 {np.percentile(x,10):9.0f}  p50 {np.percentile(x,50):9.0f} p90 {np.percentile(x,90):9.0f}")
# timeline of one CU's three workgroups (b, b+256, b+512), wave 0
for b0 in (0, 100):
    for ph in range((G + 255) // 256):
        blk = b0 + 256 * ph
        if blk >= G: continue
        row = []
        for it in range(4):
            if valid[blk, it, 0]:
                row.append("[%6.0f %6.0f %6.0f %6.0f]" % tuple(d[blk, it, 0]))
        print("blk %3d:" % blk, " ".join(row))
