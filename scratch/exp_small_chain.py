"""Where does a level of linear_small_chain_kernel spend its time?  ARDAE_SC_DEBUG=2: no compute (hand-over only), 4: no hand-over, 6: neither."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import ardae_oracle as O
from test_cdae_gpu import CdaeHarness
from ardae_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cc = O.CdaeCfg("grad", 32, 32, 256, 3, "softplus")
pc = O.init_params(O.cdae_param_spec(cc), 3)
H = CdaeHarness(cc, torch.cat([pc[n].reshape(-1) for n, _ in O.cdae_param_spec(cc)]))
lib = L.lib()
x, ctx, sigma = torch.randn(B, 32).cuda(), torch.randn(B, 32).cuda(), torch.zeros(B).cuda()
ws = torch.empty(lib.ardae_cdae_workspace_floats(ctypes.byref(H.d), B, 1, 0), device="cuda")
out = torch.empty(B, 32, device="cuda")
def run():
    L.check(lib.ardae_cdae_score(ctypes.byref(H.d), L.ptr(H.params), L.ptr(H.packed), L.ptr(x), L.ptr(sigma), L.ptr(ctx), B, 1, L.ptr(ws), ws.numel(), L.ptr(out), L.stream_ptr()))
for _ in range(20): run()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20): run()
    for _ in range(5): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(20): g.replay()
    e1.record(s)
torch.cuda.synchronize()
print(f"B={B} dbg={os.environ.get('ARDAE_SC_DEBUG','0')} chain={os.environ.get('ARDAE_SMALL_CHAIN','1')}: {e0.elapsed_time(e1) * 1e3 / 400:.2f} us per score pass (fill + 14-level launch)")

if int(os.environ.get("ARDAE_SC_DEBUG", "0")) & 8:
    nrb = (B + 15) // 16
    n = (32 * nrb + 63) // 64 * 64
    st = ws[-n:][8:24].view(torch.int64).cpu().tolist()
    t0 = st[7]
    names = ["block start (args decoded)", "epilogue operands landed", "first fragments landed", "MFMAs done", "reduced", "stored+acked", "after block", "level loop top"]
    for k in (7, 0, 1, 2, 3, 4, 5, 6):
        print(f"  {st[k] - t0:8d} cycles  {names[k]}")
