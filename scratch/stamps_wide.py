import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
import ardae_amd
from ardae_amd import _lib as L
from test_linear_gpu import pack
M, K, N = int(os.environ.get("MROWS", "131072")), 256, 256
epi = int(os.environ.get("EPI", "2"))
X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 16
S = torch.nn.functional.softplus(torch.randn(M, N, device="cuda")); Q = torch.randn(M, N, device="cuda"); R = torch.randn(M, N, device="cuda")
Y = torch.empty(M, N, device="cuda"); Y2 = torch.empty(M, N, device="cuda"); b = torch.randn(N, device="cuda")
wp = pack(W)
G = int(os.environ.get("ARDAE_WIDE_GRID", "256"))
st = torch.zeros(G * 4 * 8, dtype=torch.int64, device="cuda")
a = L.LinearArgs(); a.M, a.Nout, a.nsrc = M, N, 1
a.src[0].x = X.data_ptr(); a.src[0].ld = K; a.src[0].K = K; a.src[0].wp = wp.data_ptr()
a.act = 2; a.Y = Y.data_ptr(); a.ldY = N
if epi == 0: a.bias = b.data_ptr()
if epi in (1, 2): a.S = S.data_ptr(); a.ldS = N
if epi == 1: a.Q = Q.data_ptr(); a.ldQ = N
if epi == 2: a.R = R.data_ptr(); a.ldR = N; a.Y2 = Y2.data_ptr(); a.ldY2 = N
a.tile_loss = st.data_ptr()
for _ in range(int(os.environ.get("REPS", "3"))): L.check(L.lib().ardae_linear(ctypes.byref(a), epi, L.stream_ptr()))
torch.cuda.synchronize()
t = st.cpu().numpy().reshape(G, 4, 8).astype(np.float64)
ntile = (M // 64) / G
print("EPI %d: per tile: K loop %.0f  epilogue %.0f ; whole wave %.0f (ticks of s_memtime); ideal K loop 32768 cycles" %
      (epi, t[..., 0].mean() / ntile, t[..., 1].mean() / ntile, t[..., 2].mean()))
clk = (t[..., 2] / t[..., 3]).mean() * 100.0
print("   in-kernel clock %.0f MHz; stamped region %.1f us (median over waves)" % (clk, np.median(t[..., 3]) / 100.0))
k0 = t[..., 6].min()
print("   waves: kernel entry %.1f..%.1f us | loop begin %.1f..%.1f | loop end (after drain) %.1f..%.1f (median %.1f)   [us after the first wave entered]" % (
    (t[..., 6].min() - k0) / 100, (t[..., 6].max() - k0) / 100, (t[..., 4].min() - k0) / 100, (t[..., 4].max() - k0) / 100,
    (t[..., 5].min() - k0) / 100, (t[..., 5].max() - k0) / 100, (np.median(t[..., 5]) - k0) / 100))
