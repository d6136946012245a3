import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ardae_amd
from ardae_amd import _lib as L
B, S, z, h, Ln = (int(os.environ.get(k, v)) for k, v in (("CB", 512), ("CS", 256), ("CZ", 32), ("CH", 256), ("CL", 3)))
kind = int(os.environ.get("KIND", "0"))
d = L.CdaeDesc(kind, z, z, h, Ln, 2)
lib = L.lib()
npar = lib.ardae_cdae_param_floats(ctypes.byref(d))
params = (torch.rand(npar, device="cuda") - 0.5) * 0.12
packed = torch.empty(lib.ardae_cdae_packed_floats(ctypes.byref(d)), device="cuda")
L.check(lib.ardae_cdae_pack(ctypes.byref(d), L.ptr(params), L.ptr(packed), L.stream_ptr()))
N = B * S
xbar = torch.randn(N, z, device="cuda") * 3; sigma = torch.randn(N, device="cuda") * 0.3
eps = torch.randn(N, z, device="cuda"); ctx = torch.randn(B, z, device="cuda")
wsn = lib.ardae_cdae_workspace_floats(ctypes.byref(d), B, S, 1)
print("workspace GB", wsn * 4 / 1e9)
ws = torch.empty(wsn, device="cuda"); loss = torch.zeros(1, device="cuda"); grads = torch.zeros(npar, device="cuda")
def run():
    L.check(lib.ardae_cdae_loss_grads(ctypes.byref(d), L.ptr(params), L.ptr(packed), L.ptr(xbar), L.ptr(sigma), L.ptr(eps), L.ptr(ctx),
                                      B, S, L.ptr(ws), wsn, L.ptr(loss), L.ptr(grads), None, L.stream_ptr()))
for _ in range(3): run()
torch.cuda.synchronize()
it = int(os.environ.get("ITERS", "10"))
t0 = time.time()
for _ in range(it): run()
torch.cuda.synchronize()
dt = (time.time() - t0) / it
Lm = Ln - 1
F_inp = z * h + Lm * h * h
F_neg = (2 * h + 1) * h + Lm * h * h + (h if kind == 0 else h * z)
S_ = (h + Lm * h * h + h * h + Lm * h * h + z * h) if kind == 0 else 0
flop = 2 * (N * 3 * (F_inp + F_neg + S_) + B * 3 * F_inp)
print(f"cdae update: {dt*1e3:.3f} ms  -> {flop/dt/1e12:.1f} TFLOP/s (algorithmic {flop/1e9:.1f} GFLOP), loss {float(loss):.4f}")
