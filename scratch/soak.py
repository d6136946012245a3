"""Soak run of the contract workload (BASELINE config #2) through the fused engine under graph replay, with the scalar log channel on:
    python scratch/soak.py [steps]          -> prints the logged scalars every 1000 steps and the wall time; checks finiteness."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ardae_amd as net

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = net.MNISTIPVAE(input_dim=784, noise_dim=100, h_dim=256, num_hidden_layers=2, nonlinearity="softplus", enc_type="concat", z_dim=32).to(dev)
cdae = net.MLPGradCARDAE(input_dim=32, context_dim=32, std=1., h_dim=256, num_hidden_layers=3, nonlinearity="softplus", noise_type="gaussian",
                         enc_ctx=True, enc_input=True).to(dev)
eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=256), batch_size=512)
net.manual_seed(1)
src = net.data.StaticBinarizedSource.synthetic(50000, 784, device=dev, seed=3)
d = tempfile.mkdtemp()
log = net.ScalarLog(eng, capacity=2048, path=d)           # attaches itself to the engine
t0 = time.perf_counter()
for i in range(1, steps + 1):
    xc, xv = src.next_batch(512), src.next_batch(512)
    eng.step(xc, xv)
    if i % 1000 == 0:
        s = eng.stats()
        assert all(v == v and abs(v) < 1e30 for v in s.values()), s
        print(f"step {i:6d}  {(time.perf_counter() - t0):7.1f} s  cdae_loss {s['cdae_loss']:.4f}  model_loss {s['model_loss']:.2f}  recon {s['recon']:.2f}  "
              f"prior {s['prior']:.2f}  std {s['std_mean']:.1f}", flush=True)
        if log is not None:
            log.drain()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{steps} steps in {dt:.1f} s = {steps / dt:.1f} steps/s incl. batch gathers and log drains; dropped log records: {getattr(log, 'dropped', None)}")
