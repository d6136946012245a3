import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import ardae_amd
from ardae_amd import _lib as L
from test_linear_gpu import pack
M, K, N = 131072, 256, 256
epi = int(os.environ.get("EPI", "1"))
X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 16
S = torch.nn.functional.softplus(torch.randn(M, N, device="cuda")); Q = torch.randn(M, N, device="cuda")
Y = torch.empty(M, N, device="cuda")
wp = pack(W)
st = torch.zeros(2048 * 4 * 4, dtype=torch.int64, device="cuda")
a = L.LinearArgs(); a.M, a.Nout, a.nsrc = M, N, 1
a.src[0].x = X.data_ptr(); a.src[0].ld = K; a.src[0].K = K; a.src[0].wp = wp.data_ptr()
a.act = 2; a.Y = Y.data_ptr(); a.ldY = N
a.S = S.data_ptr(); a.ldS = N; a.Q = Q.data_ptr(); a.ldQ = N
a.tile_loss = st.data_ptr()
for _ in range(3): L.check(L.lib().ardae_linear(ctypes.byref(a), epi, L.stream_ptr()))
torch.cuda.synchronize()
t = st.cpu().numpy().reshape(2048, 4, 4).astype(np.float64)
base = t[:, :, 0].min()
d = t - base
print("kernel span (cycles of s_memtime @100MHz?):", d[:, :, 3].max())
for nm, a_, b_ in (("stage0", 0, 1), ("kloop", 1, 2), ("epilogue", 2, 3), ("total", 0, 3)):
    x = t[:, :, b_] - t[:, :, a_]
    print(f"{nm:9s} mean {x.mean():9.0f}  p10 {np.percentile(x,10):9.0f}  p50 {np.percentile(x,50):9.0f} p90 {np.percentile(x,90):9.0f}")
start = d[:, 0, 0]
print("block start times: p0 %d p25 %d p50 %d p75 %d p100 %d" % tuple(np.percentile(start, [0, 25, 50, 75, 100])))
