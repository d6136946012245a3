"""Small-shard step (the 8-rank shard of config #2 on one GPU): graph replay vs eager, host cost of a replay, capture order."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ardae_amd as net

def build(B, **kw):
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    m = net.MNISTIPVAE(input_dim=784, noise_dim=100, h_dim=256, num_hidden_layers=2, nonlinearity="softplus", enc_type="concat", z_dim=32).to(dev)
    c = net.MLPGradCARDAE(input_dim=32, context_dim=32, std=1., h_dim=256, num_hidden_layers=3, nonlinearity="softplus",
                          noise_type="gaussian", enc_ctx=True, enc_input=True).to(dev)
    eng = net.ArdaeEngine(m, c, net.TrainConfig(nz_cdae=256), batch_size=B, **kw)
    net.manual_seed(42)
    (xc,), xv = eng.input_buffers(1)
    xc.copy_((torch.rand(B, 784, device=dev) < 0.13).float()); xv.copy_((torch.rand(B, 784, device=dev) < 0.13).float())
    return eng, xc, xv

def run(tag, B, steps=200, **kw):
    eng, xc, xv = build(B, **kw)
    for _ in range(5):
        eng.step(xc, xv)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(xc, xv)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{tag:34s} B={B:4d}: {dt / steps * 1e3:7.3f} ms/step ({steps / dt:7.1f} steps/s)   host enqueue {t_host / steps * 1e3:7.3f} ms/step   graph={eng._graph is not None}", flush=True)
    del eng
    torch.cuda.empty_cache()

if __name__ == "__main__":
    Bs = [int(b) for b in sys.argv[1:]] or [64]
    for B in Bs:
        run(os.environ.get("TAG", "default"), B)
