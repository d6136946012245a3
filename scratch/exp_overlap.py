"""Per-launch overhead of the N-row linear kernel (t = a + b M) and what two independent half-size launches on two streams recover."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import ardae_amd
from ardae_amd import _lib as L
from test_linear_gpu import pack
K = N = 256
MMAX = 262144
X = torch.randn(MMAX, K, device="cuda"); W = torch.randn(N, K, device="cuda") / 16
S = torch.nn.functional.softplus(torch.randn(MMAX, N, device="cuda")); Q = torch.randn(MMAX, N, device="cuda"); R = torch.randn(MMAX, N, device="cuda")
Y = torch.empty(MMAX, N, device="cuda"); Y2 = torch.empty(MMAX, N, device="cuda"); b = torch.randn(N, device="cuda")
wp = pack(W)
def args(epi, M, row0=0):
    a = L.LinearArgs(); a.M, a.Nout, a.nsrc = M, N, 1
    o = row0 * K * 4
    a.src[0].x = X.data_ptr() + o; a.src[0].ld = K; a.src[0].K = K; a.src[0].wp = wp.data_ptr()
    a.act = 2; a.Y = Y.data_ptr() + o; a.ldY = N
    if epi == 0: a.bias = b.data_ptr()
    if epi in (1, 2): a.S = S.data_ptr() + o; a.ldS = N
    if epi == 1: a.Q = Q.data_ptr() + o; a.ldQ = N
    if epi == 2: a.R = R.data_ptr() + o; a.ldR = N; a.Y2 = Y2.data_ptr() + o; a.ldY2 = N
    return a
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
lib = L.lib()
for epi in (0, 1, 2):
    res = []
    for M in (16384, 32768, 65536, 131072, 262144):
        a = args(epi, M)
        res.append((M, timeit(lambda: L.check(lib.ardae_linear(ctypes.byref(a), epi, L.stream_ptr())))))
    (m1, t1), (m2, t2) = res[-2], res[-1]
    slope = (t2 - t1) / (m2 - m1)
    print(f"EPI {epi}: " + "  ".join(f"M={m}: {t:.1f}us" for m, t in res) + f" | slope {slope*131072:.1f} us per 131072 rows, intercept {t1 - slope*m1:.1f} us; ideal 109.2")
    # two half launches on two streams vs one full launch; dependent-chain emulation: 6 launches per stream
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    M = 131072
    aF, aA, aB = args(epi, M), args(epi, M // 2, 0), args(epi, M // 2, M // 2)
    def full():
        for _ in range(6): L.check(lib.ardae_linear(ctypes.byref(aF), epi, L.stream_ptr()))
    def halves_one_stream():
        for _ in range(6):
            L.check(lib.ardae_linear(ctypes.byref(aA), epi, L.stream_ptr())); L.check(lib.ardae_linear(ctypes.byref(aB), epi, L.stream_ptr()))
    def halves_two_streams():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            for _ in range(6): L.check(lib.ardae_linear(ctypes.byref(aA), epi, L.stream_ptr()))
        with torch.cuda.stream(s2):
            for _ in range(6): L.check(lib.ardae_linear(ctypes.byref(aB), epi, L.stream_ptr()))
        cur.wait_stream(s1); cur.wait_stream(s2)
    print(f"   6 launches: full {timeit(full, 10):.1f} us | halves, one stream {timeit(halves_one_stream, 10):.1f} | halves, two streams {timeit(halves_two_streams, 10):.1f}")
