import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import ardae_amd
from ardae_amd import _lib as L
from test_linear_gpu import pack
M, K, N = 131072, 256, 256
epi = int(os.environ.get("EPI", "0"))
X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 16
S = torch.nn.functional.softplus(torch.randn(M, N, device="cuda")); Q = torch.randn(M, N, device="cuda")
Y = torch.empty(M, N, device="cuda"); b = torch.randn(N, device="cuda")
wp = pack(W)
st = torch.zeros(256 * 12 * 4, dtype=torch.int64, device="cuda")
a = L.LinearArgs(); a.M, a.Nout, a.nsrc = M, N, 1
a.src[0].x = X.data_ptr(); a.src[0].ld = K; a.src[0].K = K; a.src[0].wp = wp.data_ptr()
a.act = 2; a.Y = Y.data_ptr(); a.ldY = N
if epi == 0: a.bias = b.data_ptr()
else: a.S = S.data_ptr(); a.ldS = N; a.Q = Q.data_ptr(); a.ldQ = N
a.tile_loss = st.data_ptr()
for _ in range(3): L.check(L.lib().ardae_linear(ctypes.byref(a), epi, L.stream_ptr()))
torch.cuda.synchronize()
t = st.cpu().numpy().reshape(256, 12, 4).astype(np.float64)
G = t[0, 0, 2]
r = lambda sl: (t[:, sl, :2].mean(axis=(0, 1)) / G).round(0)
print("EPI", epi, "steps/WG", G, "per-step [work, barrier wait]: consumer", r(slice(0, 4)), "loaderE", r(slice(4, 6)), "loaderO", r(slice(6, 8)), "epilogue", r(slice(8, 12)))
x = st.cpu().numpy().reshape(256, 12, 4)[:, 4:8, 3].astype(np.uint64)
print("loader per ACTIVE step: wait-for-loads", ((x >> np.uint64(32)).astype(np.float64).mean() / (G / 2)).round(0), " stash", ((x & np.uint64(0xffffffff)).astype(np.float64).mean() / (G / 2)).round(0))
