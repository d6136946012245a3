"""Whole-step time of the BASELINE.json configs' per-GPU shards on one MI355X (scratch; the contract bench is bench.py = config #2).

    python scratch/bench_configs.py [1 2 4 5]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ardae_amd as net

def cfg(i):
    if i == 1:   # 25 Gaussians, ToyIPVAE concat z=2, relu; cdae mlp-grad h=256 L=3; batch 512 x nz 256
        m = net.ToyIPVAE(input_dim=2, noise_dim=10, h_dim=256, num_hidden_layers=2, nonlinearity="relu", enc_type="concat", z_dim=2)
        c = net.MLPGradCARDAE(input_dim=2, context_dim=2, std=1., h_dim=256, num_hidden_layers=3, nonlinearity="softplus",
                              noise_type="gaussian", enc_ctx=True, enc_input=True)
        return m, c, 512, 256, lambda B, dev: net.data.gaussians25(25 * ((B + 24) // 25))[0][torch.randperm(25 * ((B + 24) // 25))[:B].to(dev)].contiguous(), dict(z=2, h=256, L=3, kind="grad")
    if i == 2:
        m = net.MNISTIPVAE(input_dim=784, noise_dim=100, h_dim=256, num_hidden_layers=2, nonlinearity="softplus", enc_type="concat", z_dim=32)
        c = net.MLPGradCARDAE(input_dim=32, context_dim=32, std=1., h_dim=256, num_hidden_layers=3, nonlinearity="softplus",
                              noise_type="gaussian", enc_ctx=True, enc_input=True)
        return m, c, 512, 256, lambda B, dev: (torch.rand(B, 784, device=dev) < 0.13).float(), dict(z=32, h=256, L=3, kind="grad")
    if i == 4:   # conv model, cdae h=512 L=4, batch 1024 x nz 512 over 4 GPUs -> 256 images per GPU
        m = net.ConvIPVAE(input_height=28, input_channels=1, z_dim=32, noise_dim=100, nonlinearity="softplus")
        c = net.MLPGradCARDAE(input_dim=32, context_dim=32, std=1., h_dim=512, num_hidden_layers=4, nonlinearity="softplus",
                              noise_type="gaussian", enc_ctx=True, enc_input=True)
        return m, c, 256, 512, lambda B, dev: (torch.rand(B, 1, 28, 28, device=dev) < 0.13).float(), dict(z=32, h=512, L=4, kind="grad")
    if i == 5:   # 3072-pixel flat images, resdae h=1024 L=6, batch 2048 x nz 1024 over 8 GPUs -> 256 images per GPU
        m = net.MNISTIPVAE(input_dim=3072, noise_dim=100, h_dim=256, num_hidden_layers=2, nonlinearity="softplus", enc_type="concat", z_dim=32)
        c = net.MLPResCARDAE(input_dim=32, context_dim=32, std=1., h_dim=1024, num_hidden_layers=6, nonlinearity="softplus",
                             noise_type="gaussian", enc_ctx=True, enc_input=True)
        return m, c, 256, 1024, lambda B, dev: (torch.rand(B, 3072, device=dev) < 0.5).float(), dict(z=32, h=1024, L=6, kind="res")
    if i == 6:   # the shipped dbMNIST recipe for the mlp model (run_vae_dbmnist.sh: mnist-concat, batch 128, nz_cdae 625, cdae L 5)
        m = net.MNISTIPVAE(input_dim=784, noise_dim=100, h_dim=256, num_hidden_layers=2, nonlinearity="softplus", enc_type="concat", z_dim=32)
        c = net.MLPGradCARDAE(input_dim=32, context_dim=32, std=1., h_dim=256, num_hidden_layers=5, nonlinearity="softplus",
                              noise_type="gaussian", enc_ctx=True, enc_input=True)
        return m, c, 128, 625, lambda B, dev: (torch.rand(B, 784, device=dev) < 0.13).float(), dict(z=32, h=256, L=5, kind="grad")
    if i == 7:   # the shipped "hierarchical mlp" recipe (run_vae_dbmnist.sh: --model auxmnist h 300, --cdae-ctx-type hidden1a, nz_cdae 625, L 5)
        m = net.MNISTAuxIPVAE(input_dim=784, noise_dim=100, h_dim=300, num_hidden_layers=2, nonlinearity="softplus", enc_type="simple", z_dim=32,
                              clip_z0_logvar="none", clip_z_logvar="none")
        c = net.MLPGradCARDAE(input_dim=32, context_dim=600, std=1., h_dim=256, num_hidden_layers=5, nonlinearity="softplus",
                              noise_type="gaussian", enc_ctx=True, enc_input=True)
        return m, c, 128, 625, lambda B, dev: (torch.rand(B, 784, device=dev) < 0.13).float(), dict(z=32, h=256, L=5, kind="grad", ctx="hidden1a")
    if i == 8:   # the shipped "hierarchical conv" recipe (run_vae_dbmnist.sh: --model auxconv, hidden1a context of 1600 columns, nz_cdae 625, L 5)
        m = net.MNISTConvAuxIPVAE(input_height=28, input_channels=1, z0_dim=100, z_dim=32, nonlinearity="softplus")
        c = net.MLPGradCARDAE(input_dim=32, context_dim=1600, std=1., h_dim=256, num_hidden_layers=5, nonlinearity="softplus",
                              noise_type="gaussian", enc_ctx=True, enc_input=True)
        return m, c, 128, 625, lambda B, dev: (torch.rand(B, 1, 28, 28, device=dev) < 0.13).float(), dict(z=32, h=256, L=5, kind="grad", ctx="hidden1a")
    if i in (9, 10):   # the shipped "implicit resconv" / "hierarchical resconv" recipes (run_vae_dbmnist.sh: --model resconvct-res / auxresconvct, ELU,
        # --cdae mlp-res h 512 L 5, nz_cdae 625, --std-scale 100, --num-cdae-updates 2, Adam (0.9, 0.999) lr 1e-3, RMSprop momentum 0.9)
        if i == 9:
            m = net.ResConvIPVAE(input_height=28, input_channels=1, z_dim=32, h_dim=512, num_hidden_layers=1, noise_dim=100, nonlinearity="elu",
                                 do_center=True, enc_type="res-wn-mlp")
        else:
            m = net.MNISTResConvAuxIPVAE(input_height=28, input_channels=1, z_dim=32, c_dim=450, z0_dim=100, nonlinearity="elu", do_center=True)
        c = net.MLPResCARDAE(input_dim=32, context_dim=32 if i == 9 else 450, std=1., h_dim=512, num_hidden_layers=5, nonlinearity="softplus",
                             noise_type="gaussian", enc_ctx=True, enc_input=True)
        return m, c, 128, 625, lambda B, dev: (torch.rand(B, 1, 28, 28, device=dev) < 0.13).float(), dict(
            z=32, h=512, L=5, kind="res", ctx="lt0" if i == 9 else "hidden1a", updates=2, tcfg=dict(std_scale=100., m_lr=1e-3, m_beta1=0.9, d_momentum=0.9))
    raise SystemExit(f"no config {i}")

def cdae_flops(B, nz, z, h, L, kind):
    N, Lm = B * nz, L - 1
    F_inp = z * h + Lm * h * h
    F_neg = (2 * h + 1) * h + Lm * h * h + (h if kind == "grad" else h * z)
    S = (h + Lm * h * h + h * h + Lm * h * h + z * h) if kind == "grad" else 0
    return 2 * (N * 3 * (F_inp + F_neg + S) + B * 3 * F_inp)

for i in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 5]:
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m, c, B, nz, data, shp = cfg(i)
    nz = int(os.environ.get('CFG_NZ', nz))
    m, c = m.to(dev), c.to(dev)
    updates = shp.pop("updates", 1)
    eng = net.ArdaeEngine(m, c, net.TrainConfig(nz_cdae=nz, cdae_ctx_type=shp.pop("ctx", "lt0"), num_cdae_updates=updates, **shp.pop("tcfg", {})), batch_size=B)
    x1, x2 = data(B, dev), data(B, dev)
    steps = int(os.environ.get("CFG_STEPS", "20" if i < 4 else "5"))
    for _ in range(3):
        eng.step(x1, x2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step(x1, x2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    fl = updates * cdae_flops(B, nz, **shp)
    st = eng.stats()
    print(f"config #{i}: {B} images x {nz} samples per GPU: {dt*1e3:8.2f} ms/step  {1/dt:7.1f} steps/s  cDAE-update FLOPs {fl/1e12:.3f} T -> >= {fl/dt/1e12:5.1f} TFLOP/s"
          f"  (cdae_loss {st['cdae_loss']:.4f}, model_loss {st['model_loss']:.2f})", flush=True)
    del eng, m, c
    torch.cuda.empty_cache()
