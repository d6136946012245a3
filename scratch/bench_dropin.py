"""Steps/s of the reference's own loop body (ivae_ardae.py:713-846) written against this package's drop-in modules and
optimisers, at BASELINE config #2 - what a user gets WITHOUT switching to ArdaeEngine."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ardae_amd as net

device = torch.device("cuda")
B, nz, nz_model, std_scale, delta, beta = 512, 256, 1, 1e4, 0.1, 1.0
model = net.MNISTIPVAE(input_dim=784, noise_dim=100, h_dim=256, num_hidden_layers=2, nonlinearity="softplus", enc_type="concat", z_dim=32).to(device)
cdae = net.MLPGradCARDAE(input_dim=32, context_dim=32, std=1., h_dim=256, num_hidden_layers=3, nonlinearity="softplus",
                         noise_type="gaussian", enc_ctx=True, enc_input=True).to(device)
model_optimizer = net.Adam(model.parameters(), lr=1e-4, betas=(0.5, 0.999))
cdae_optimizer = net.RMSprop(cdae.parameters(), lr=1e-4, momentum=0.5)
p = torch.full((784,), 0.13, device=device)

def step():
    x = net.data.dynamic_binarize(p.expand(B, -1).contiguous())
    model.train(); cdae.train()
    cdae_optimizer.zero_grad()
    context = model.encode(x, std=0).detach()
    latent_mean = model.encode(x, std=0).detach()
    latent = model.forward_hidden(x, nz=nz).detach()
    lsm = std_scale * (latent - latent_mean)
    std = delta * torch.mean(torch.std(lsm, dim=1, keepdim=True), dim=2, keepdim=True)
    stdmat = std * torch.randn(B, nz, 1, device=device)
    _, cdae_loss = cdae(lsm, context, std=stdmat, scale=std_scale)
    cdae_loss.backward()
    cdae_optimizer.step()
    xv = net.data.dynamic_binarize(p.expand(B, -1).contiguous())
    model.train(); cdae.eval()
    model_optimizer.zero_grad()
    _, _, latent, model_loss, rec, pri = model(xv, beta=beta, eta=0., lmbd=0., nz=nz_model)
    model_loss.backward(retain_graph=True)
    context = model.encode(xv, std=0).detach()
    latent_mean = model.encode(xv, std=0).detach()
    lsm = std_scale * (latent - latent_mean).detach()
    grad = cdae.glogprob(lsm, context, std=torch.zeros(B, nz_model, 1, device=device), scale=std_scale).detach()
    (std_scale * (latent - latent_mean)).backward(beta * grad / float(B * nz_model))
    model_optimizer.step()
    return cdae_loss, model_loss

for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n):
    cl, ml = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"drop-in module loop, config #2: {dt*1e3:.2f} ms/step = {1/dt:.1f} steps/s (cdae_loss {float(cl):.4f}, model_loss {float(ml):.2f})")
