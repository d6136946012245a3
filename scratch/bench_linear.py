import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import ardae_amd
from ardae_amd import _lib as L
from test_linear_gpu import pack
M = int(os.environ.get("MROWS", "131072")); K = N = int(os.environ.get("KN", "256"))
epi = int(os.environ.get("EPI", "1"))
X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 16
S = torch.nn.functional.softplus(torch.randn(M, N, device="cuda")); Q = torch.randn(M, N, device="cuda"); R = torch.randn(M, N, device="cuda")
Y = torch.empty(M, N, device="cuda"); Y2 = torch.empty(M, N, device="cuda")
b = torch.randn(N, device="cuda")
wp = pack(W)
a = L.LinearArgs(); a.M, a.Nout, a.nsrc = M, N, 1
a.src[0].x = X.data_ptr(); a.src[0].ld = K; a.src[0].K = K; a.src[0].wp = wp.data_ptr()
a.act = 2; a.Y = Y.data_ptr(); a.ldY = N
if epi == 0: a.bias = b.data_ptr()
if epi in (1, 2): a.S = S.data_ptr(); a.ldS = N
if epi == 1 and os.environ.get("NOQ") is None: a.Q = Q.data_ptr(); a.ldQ = N
if epi == 2: a.R = R.data_ptr(); a.ldR = N; a.Y2 = Y2.data_ptr(); a.ldY2 = N
def run(): L.check(L.lib().ardae_linear(ctypes.byref(a), epi, L.stream_ptr()))
for _ in range(5): run()
torch.cuda.synchronize()
it = int(os.environ.get("ITERS", "30"))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(it): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / it
print(f"EPI={epi} M={M} K={K} N={N}: {ms*1e3:.1f} us  {2*M*K*N/ms/1e9:.1f} TFLOP/s")
if os.environ.get("GRAPH"):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(it): run()
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"graph replay: {e0.elapsed_time(e1) / (5 * it) * 1e3:.1f} us per launch")
