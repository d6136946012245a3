"""Data-parallel equivalence on CPU (gloo, world_size 2): 1 rank x B == mean over 2 ranks x B/2.

The HIP kernels cannot run here, so the per-rank compute is the oracle; what is under test is the package's DP host
logic (pytorch-ardae-vae_amd/dist.py): shard boundaries, flat-buffer all-reduce + averaging, and the entropy-seed
normalisation rule under sharding (SURVEY 8 a-P: the seed uses the global batch size).
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import ardae_oracle as O

MC = O.ModelCfg("mnist", 24, 10, 32, 4, 2, "softplus")
CC = O.CdaeCfg("grad", 4, 4, 32, 2)
TC = O.TrainCfg(nz_cdae=6)
B = 8


def _inputs():
    g = torch.Generator().manual_seed(3)
    x1 = torch.bernoulli(torch.full((B, 24), 0.3), generator=g)
    x2 = torch.bernoulli(torch.full((B, 24), 0.3), generator=g)
    noise = O.draw_step_noise(MC, TC, B, g)
    pm = O.init_params(O.model_param_spec(MC), 0, O.model_init_special(MC))
    pc = O.init_params(O.cdae_param_spec(CC), 1)
    return x1, x2, noise, pm, pc


def _flat(grads, spec):
    return torch.cat([(grads[n] if grads[n] is not None else torch.zeros(shp)).reshape(-1) for n, shp in spec])


def _vae_grads_local(pm, pc, x, noise_v, local_B):
    """VAE phase with the seed normalised by the LOCAL batch (dist.entropy_seed_scale) -- what each rank computes."""
    from ardae_amd import dist
    pm_req = {k: v.detach().requires_grad_(True) for k, v in pm.items()}
    z, loss, _, _, _ = O.vae_forward(MC, pm_req, x, noise_v, TC.beta, TC.nz_model)
    with torch.no_grad():
        z0 = O.encode(MC, pm, x, torch.zeros(x.size(0), MC.noise_dim), 1)
    u = (TC.std_scale * (z - z0)).detach()
    g = O.cdae_glogprob(CC, pc, u, z0, torch.zeros(x.size(0), TC.nz_model, 1))
    seed = g * dist.entropy_seed_scale(TC.std_scale, TC.beta, local_B, TC.nz_model)
    total = loss + (z * seed).sum()
    names = list(pm_req)
    return dict(zip(names, torch.autograd.grad(total, [pm_req[n] for n in names])))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from ardae_amd import dist
    x1, x2, noise, pm, pc = _inputs()
    lo, hi = dist.shard_rows(B)
    assert (lo, hi) == (rank * B // world, (rank + 1) * B // world)
    nzc = TC.nz_cdae
    n_loc = {"sampler": noise["sampler"][lo * nzc:hi * nzc], "sigma": noise["sigma"][lo:hi], "eps": noise["eps"][lo * nzc:hi * nzc],
             "vae": noise["vae"][lo:hi]}
    _, gc, _ = O.cdae_update_grads(MC, CC, TC, pm, pc, x1[lo:hi], n_loc)
    fc = _flat(gc, O.cdae_param_spec(CC))
    dist.allreduce_mean_(fc)
    gm = _vae_grads_local(pm, pc, x2[lo:hi], n_loc["vae"], hi - lo)
    fm = _flat(gm, O.model_param_spec(MC))
    dist.allreduce_mean_(fm)
    if rank == 0:
        torch.save({"fc": fc, "fm": fm}, out)
    torch.distributed.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_sharding_equals_single_rank(tmp_path):
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    x1, x2, noise, pm, pc = _inputs()
    _, gc, _ = O.cdae_update_grads(MC, CC, TC, pm, pc, x1, noise)
    _, _, _, _, gm = O.vae_update_grads(MC, CC, TC, pm, pc, x2, noise)      # reference rule: seed / (global B * nz)
    ref_c, ref_m = _flat(gc, O.cdae_param_spec(CC)), _flat(gm, O.model_param_spec(MC))
    rel = lambda a, b: float((a - b).norm() / b.norm())
    assert rel(got["fc"], ref_c) < 5e-4            # fp32 reduction order only (same tolerance regime as the cDAE tests)
    assert rel(got["fm"], ref_m) < 5e-4


def test_shard_rows_rejects_uneven_batches():
    from ardae_amd import dist
    assert dist.shard_rows(512) == (0, 512)        # single process
    assert dist.world_size() == 1 and dist.rank() == 0
    assert dist.entropy_seed_scale(1e4, 1.0, 64, 1) == pytest.approx(1e4 / 64)
