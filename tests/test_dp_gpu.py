"""Data parallelism with the REAL engine: two ranks share the one GPU of the test box (gloo moves the flat gradient
buffers; RCCL refuses two ranks on one device), each runs the HIP path on its half of the image batch, and the averaged
gradients must equal the single-process engine's on the whole batch (the same ArdaeEngine code path - eager
launches, side stream, all-reduce before each optimiser step - that `bench.py --gpus N` runs under RCCL).

Also rehearses `bench.py` itself under `torch.distributed.run` with two ranks (BENCH_BACKEND=gloo, both on cuda:0).
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

from oracle import ardae_oracle as O

pytestmark = pytest.mark.gpu

MC = O.ModelCfg("mnist", 48, 12, 64, 8, 2, "softplus")
CC = O.CdaeCfg("grad", 8, 8, 64, 3)
B, NZ = 16, 32          # 512 Monte-Carlo rows in all, 256 per rank


def _build(dev):
    import ardae_amd as net
    model = net.MNISTIPVAE(input_dim=MC.input_dim, noise_dim=MC.noise_dim, h_dim=MC.h_dim, num_hidden_layers=MC.n_layers,
                           nonlinearity=MC.nonlin, enc_type="concat", z_dim=MC.z_dim)
    cdae = net.MLPGradCARDAE(input_dim=CC.input_dim, context_dim=CC.context_dim, std=1., h_dim=CC.h_dim, num_hidden_layers=CC.n_layers,
                             nonlinearity=CC.nonlin, noise_type="gaussian", enc_ctx=True, enc_input=True)
    model.load_state_dict(O.init_params(O.model_param_spec(MC), 0, O.model_init_special(MC)))
    cdae.load_state_dict(O.init_params(O.cdae_param_spec(CC), 1))
    return model.to(dev), cdae.to(dev)


def _inputs():
    g = torch.Generator().manual_seed(11)
    x1 = torch.bernoulli(torch.full((B, MC.input_dim), 0.3), generator=g)
    x2 = torch.bernoulli(torch.full((B, MC.input_dim), 0.3), generator=g)
    noise = O.draw_step_noise(MC, O.TrainCfg(nz_cdae=NZ), B, g)
    return x1, x2, noise


def _grads(x1, x2, noise, per_rank_batch, **cfg):
    """cDAE-phase and VAE-phase gradients (all-reduced inside the engine when a process group is up), no parameter update.
    noise=None: the engine's own Philox stream - every rank generates ITS ROWS of the global draws, so the result must not
    depend on the number of ranks either."""
    import ardae_amd as net
    dev = torch.device("cuda", 0)
    model, cdae = _build(dev)
    net.manual_seed(99)
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=NZ, **cfg), batch_size=per_rank_batch)
    nz = None if noise is None else {k: v.to(dev).contiguous() for k, v in noise.items()}
    eng.cdae_phase(x1.to(dev), nz, apply_update=False)
    eng.vae_phase(x2.to(dev), nz, apply_update=False)
    torch.cuda.synchronize()
    return eng.grads_c[:eng.n_c].cpu().clone(), eng.grads_m.cpu().clone(), eng.loss_c.cpu().clone()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from ardae_amd import dist
    x1, x2, noise = _inputs()
    lo, hi = dist.shard_rows(B)
    n_loc = {"sampler": noise["sampler"][lo * NZ:hi * NZ], "sigma": noise["sigma"][lo:hi], "eps": noise["eps"][lo * NZ:hi * NZ],
             "vae": noise["vae"][lo:hi]}
    gc, gm, loss = _grads(x1[lo:hi], x2[lo:hi], n_loc, hi - lo)
    lt = loss.clone()
    torch.distributed.all_reduce(lt)
    gc2, gm2, loss2 = _grads(x1[lo:hi], x2[lo:hi], None, hi - lo)        # own noise stream
    lt2 = loss2.clone()
    torch.distributed.all_reduce(lt2)
    if rank == 0:
        torch.save({"gc": gc, "gm": gm, "loss": lt / world, "gc_own": gc2, "gm_own": gm2, "loss_own": lt2 / world}, out)
    torch.distributed.destroy_process_group()


def _steps(per_rank_batch, lo, hi, graph, nsteps=5, force_dp=False, dp_comm="auto", **cfg):
    """nsteps full train steps (own Philox noise, fresh images per step) on images [lo, hi) of each global batch; returns the
    flat parameters and whether the engine ended up replaying captured graphs."""
    import ardae_amd as net
    dev = torch.device("cuda", 0)
    model, cdae = _build(dev)
    net.manual_seed(99)
    eng = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=NZ, **cfg), batch_size=per_rank_batch, graph=graph, force_dp=force_dp, dp_comm=dp_comm)
    g = torch.Generator().manual_seed(21)
    for _ in range(nsteps):
        x1 = torch.bernoulli(torch.full((B, MC.input_dim), 0.3), generator=g)
        x2 = torch.bernoulli(torch.full((B, MC.input_dim), 0.3), generator=g)
        eng.step(x1[lo:hi].contiguous().to(dev), x2[lo:hi].contiguous().to(dev))
    torch.cuda.synchronize()
    return model.flat_params().cpu().clone(), cdae.flat_params().cpu().clone(), eng.plan_summary()


def _worker_steps(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from ardae_amd import dist
    lo, hi = dist.shard_rows(B)
    pm_e, pc_e, g_e = _steps(hi - lo, lo, hi, graph=False)
    pm_g, pc_g, g_g = _steps(hi - lo, lo, hi, graph=True)
    if rank == 0:
        torch.save({"pm_eager": pm_e, "pc_eager": pc_e, "pm_graph": pm_g, "pc_graph": pc_g, "eager_has_graph": g_e is not None,
                    "segments": g_g}, out)
    torch.distributed.destroy_process_group()


def test_two_rank_segmented_graph_replay_equals_eager(tmp_path):
    """world > 1 replays the step as linear graphs cut at the two gradient all-reduces and at the cross-stream waits
    (engine._plan / _units): five steps of two ranks with replay must give exactly the parameters of five steps of two ranks with
    eager launches (same kernels, same collectives, same noise), and both must track the single-process run on the whole batch."""
    out = str(tmp_path / "dp_steps.pt")
    mp.spawn(_worker_steps, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    assert got["eager_has_graph"] is False
    # VAE forward half beside | cDAE gradients | all-reduce | cDAE update | VAE backward (joins the side stream) | all-reduce | model update + advance
    assert got["segments"] == ["graph:side", "graph:main", "allreduce", "graph:main", "graph:main", "allreduce", "graph:main"]
    assert torch.equal(got["pm_graph"], got["pm_eager"]) and torch.equal(got["pc_graph"], got["pc_eager"])
    pm1, pc1, g1 = _steps(B, 0, B, graph=True)
    assert g1 == ["graph:side", "graph:main", "graph:main"]
    # five sign-like RMSprop / Adam steps amplify the fp32 sum-order difference between "two halves averaged" and "one batch"
    # (single-step agreement is pinned at 5e-4 in test_two_ranks_on_one_gpu_equal_single_process): the trajectories must track
    model0, cdae0 = _build(torch.device("cuda", 0))
    for name, after, ref, p0 in (("model", got["pm_graph"], pm1, model0.flat_params().cpu()), ("cdae", got["pc_graph"], pc1, cdae0.flat_params().cpu())):
        upd, want = (after - p0).double(), (ref - p0).double()
        err = (upd - want).abs() / (want.abs() + 1e-12)
        # (round 4: the per-image layers add their products in ONE order whatever block shape the row count selects - linear_small.hip,
        # test_linear_per_image_blocks_bit_identical - so only the sums over rows differ between the half batches and the whole one)
        assert float(err.median()) < 1e-2, name
        assert float((upd - want).norm() / want.norm()) < 0.1, name


def _worker_nstd(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from ardae_amd import dist
    x1, x2, _ = _inputs()
    lo, hi = dist.shard_rows(B)
    gc, gm, loss = _grads(x1[lo:hi], x2[lo:hi], None, hi - lo, nstd_cdae=3)
    lt = loss.clone()
    torch.distributed.all_reduce(lt)
    if rank == 0:
        torch.save({"gc": gc, "gm": gm, "loss": lt / world}, out)
    torch.distributed.destroy_process_group()


def test_two_ranks_equal_single_process_with_nstd(tmp_path):
    """--train-nstd-cdae 3 under data parallelism: every rank draws ITS rows of the (B nz nstd)-row global sigma / eps draws, so two
    ranks on half batches reproduce the single-process gradients."""
    out = str(tmp_path / "dp_nstd.pt")
    mp.spawn(_worker_nstd, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    x1, x2, _ = _inputs()
    gc, gm, loss = _grads(x1, x2, None, B, nstd_cdae=3)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert abs(float(got["loss"]) - float(loss)) <= 1e-5 * abs(float(loss))
    assert rel(got["gc"], gc) < 5e-4 and rel(got["gm"], gm) < 5e-4


TWO_UPDATES = dict(num_cdae_updates=2, m_optimizer="amsgrad", d_optimizer="adam", d_beta1=0.6)


def _worker_steps2(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from ardae_amd import dist
    lo, hi = dist.shard_rows(B)
    pm_e, pc_e, _ = _steps(hi - lo, lo, hi, graph=False, **TWO_UPDATES)
    pm_g, pc_g, g_g = _steps(hi - lo, lo, hi, graph=True, **TWO_UPDATES)
    if rank == 0:
        torch.save({"pm_eager": pm_e, "pc_eager": pc_e, "pm_graph": pm_g, "pc_graph": pc_g,
                    "segments": g_g}, out)
    torch.distributed.destroy_process_group()


def test_two_rank_replay_with_two_cdae_updates_and_adam_pair(tmp_path):
    """--num-cdae-updates 2 (the shipped residual-conv recipes) on two ranks: three gradient all-reduces per step, so the step
    replays as linear graphs cut at each of them (and at the join with the side stream); with Adam on both networks each keeps its own device-resident step count (the cDAE's advances twice
    per step).  Replayed == eager bit for bit."""
    out = str(tmp_path / "dp_steps2.pt")
    mp.spawn(_worker_steps2, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    # the cDAE update of batch 0 and the gradients of batch 1 share a graph (no collective, no wait between them)
    assert got["segments"] == ["graph:side", "graph:main", "allreduce", "graph:main", "allreduce", "graph:main", "graph:main", "allreduce", "graph:main"]
    assert torch.equal(got["pm_graph"], got["pm_eager"]) and torch.equal(got["pc_graph"], got["pc_eager"])
    assert torch.isfinite(got["pm_graph"]).all() and torch.isfinite(got["pc_graph"]).all()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_gpu_equal_single_process(tmp_path):
    out = str(tmp_path / "dp_gpu.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    x1, x2, noise = _inputs()
    gc, gm, loss = _grads(x1, x2, noise, B)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert abs(float(got["loss"]) - float(loss)) <= 1e-5 * abs(float(loss))
    # same arithmetic on the same rows; only the order of the fp32 sums over rows differs (split in two, then averaged)
    assert rel(got["gc"], gc) < 5e-4
    assert rel(got["gm"], gm) < 5e-4
    # the engine's own draws: rank r generated rows [r B/2, (r+1) B/2) of the same global noise tensors
    gc, gm, loss = _grads(x1, x2, None, B)
    # (half batches run their per-image layers on 16 x 16 blocks, the full batch on 32 x 32: the SAME order of the sums over k since
    # round 4 - linear_small.hip - so a row's bits do not depend on the shard size)
    assert abs(float(got["loss_own"]) - float(loss)) <= 1e-5 * abs(float(loss))
    assert rel(got["gc_own"], gc) < 5e-4
    assert rel(got["gm_own"], gm) < 5e-4


def test_bench_two_ranks_rehearsal(tmp_path):
    """bench.py under torch.distributed.run with two ranks on the one GPU: the launch contract (env rendezvous, barrier,
    max-over-ranks timing, one JSON line from rank 0) end to end."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ARDAE_DEBUG_KNOBS="1", BENCH_BACKEND="gloo", BENCH_ONE_GPU="1", BENCH_GLOBAL_B="32")
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)          # bench.py must provide it itself - before anything touches the GPU
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=240, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["per_gpu_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["roofline"]["kernel"]
    for k in ("cdae_loss", "model_loss"):
        assert d["losses"][k] == d["losses"][k]          # not NaN
    # the HSA / RCCL environment defaults were in place BEFORE the first call that initialises the GPU runtime (bench.py records
    # torch.cuda.is_initialized() at the moment it sets them), and the line says who took part
    assert d["ranks"]["env"]["set_before_device_init"] is True and d["ranks"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert d["ranks"]["torch_distributed"] == {"backend": "gloo", "world_size": 2} and d["ranks"]["rccl"] is None
    assert d["config"]["workload"].startswith("EXPERIMENT (BENCH_GLOBAL_B=32)") and "batch 32," in d["metric"]
    # the multi-rank fallback (a refused capture of the collectives): every rank fails attempt 0, all agree, attempt 1 runs and the line says so
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "3",
                        "--no-cpu-baseline", "--prof-steps", "0"], env=dict(env, BENCH_INJECT_FAILURE="0"), capture_output=True, text=True, timeout=240, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d2 = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert "attempt 0" in d2["ranks"]["dp_fallback"] and d2["value"] > 0 and d["ranks"]["dp_fallback"] is None
    # a rank that fails must end the job with a non-zero status (here: a global batch that does not divide over the ranks)
    env_bad = dict(env, BENCH_GLOBAL_B="33")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline"], env=env_bad, capture_output=True, text=True, timeout=240, cwd=root)
    assert r.returncode != 0 and "[bench.py rank" in r.stderr


def test_bench_one_rank_rccl_end_to_end(tmp_path):
    """`bench.py` under torch.distributed.run on the `nccl` (= RCCL) backend, as far as one GPU allows: ONE rank with the data-parallel
    plan forced (BENCH_FORCE_DP=1) - environment defaults, process group, the C ABI's RCCL communicator, both gradient all-reduces
    captured in the step graphs, the collective-latency report - the code path the driver's N > 1 runs take, end to end."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ARDAE_DEBUG_KNOBS="1", BENCH_FORCE_DP="1", BENCH_GLOBAL_B="32")
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=240, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["hip_graph"] is True and d["value"] > 0
    rk = d["ranks"]
    assert rk["env"]["set_before_device_init"] is True and rk["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert rk["torch_distributed"] == {"backend": "nccl", "world_size": 1}
    assert rk["rccl"]["ranks"] == 1 and rk["rccl"]["rank"] == 0 and rk["rccl"]["library"].startswith("RCCL ")
    assert rk["plan"] == ["graph:side", "graph:main", "graph:main"]          # the all-reduces are inside the graphs
    assert set(rk["collectives"]) == {"cdae_grads", "model_grads"}


def _worker_rccl_one_rank(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from ardae_amd import dist, _lib as L
    # (a) the C ABI's own communicator (dist.DpComm): what `ArdaeEngine` builds by itself on an RCCL process group
    pm_e, pc_e, _ = _steps(B, 0, B, graph=False, force_dp=True)
    pm_g, pc_g, plan = _steps(B, 0, B, graph=True, force_dp=True)
    # (b) torch.distributed's collectives as eager items between the graphs (dp_comm=None: the rehearsal path, here on RCCL)
    pm_t, pc_t, plan_t = _steps(B, 0, B, graph=True, force_dp=True, dp_comm=None)
    pm_1, pc_1, plan1 = _steps(B, 0, B, graph=True)                 # the ordinary single-rank plan (no collectives)
    # (c) the entry points themselves: query, a stream-ordered call, a captured call
    comm = dist.DpComm()
    buf = torch.randn(543489, device="cuda")
    want = buf.clone()
    comm.allreduce_mean_(buf)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=side):
        buf.mul_(2.0)
        comm.allreduce_mean_(buf)
        buf.add_(1.0)
    g.replay(); g.replay()
    torch.cuda.synchronize()
    q = comm.query()
    comm.close()
    bad = L.lib().ardae_dp_allreduce_mean(None, L.ptr(buf), buf.numel(), None)
    torch.save({"pm_eager": pm_e, "pc_eager": pc_e, "pm_graph": pm_g, "pc_graph": pc_g, "pm_torch": pm_t, "pc_torch": pc_t, "pm_plain": pm_1,
                "pc_plain": pc_1, "plan": plan, "plan_torch": plan_t, "plain_plan": plan1, "backend": torch.distributed.get_backend(),
                "query": list(q), "rccl": comm.backend, "captured_ok": bool(torch.equal(buf, (want * 2 + 1) * 2 + 1)), "bad_rc": bad}, out)
    torch.distributed.destroy_process_group()


def test_rccl_allreduce_inside_the_c_abi_captured_in_the_step_graphs(tmp_path):
    """SURVEY 8(b) `dp_allreduce_flat`, as far as one GPU allows: a ONE-rank RCCL communicator owned by the C ABI (`ardae_dp_comm_create`;
    RCCL accepts nranks = 1).  With `force_dp=True` the step runs the data-parallel plan, whose two gradient all-reduces are calls of
    `ardae_dp_allreduce_mean` on the launch stream - CAPTURED into the step's linear graphs, so the replayed multi-rank step submits the
    same three graphs as the single-rank one and no collective cuts it.  A one-rank all-reduce is the identity: the result must equal
    the eager run of the same plan, the plan with torch.distributed's collectives between the graphs, and the ordinary single-rank
    plan, bit for bit."""
    out = str(tmp_path / "rccl1.pt")
    mp.spawn(_worker_rccl_one_rank, args=(1, _free_port(), out), nprocs=1, join=True)
    got = torch.load(out, weights_only=True)
    assert got["backend"] == "nccl" and got["rccl"].startswith("RCCL ")
    assert got["query"] == [1, 0, 0] and got["captured_ok"] and got["bad_rc"] < 0
    assert got["plan"] == ["graph:side", "graph:main", "graph:main"] == got["plain_plan"]            # no "allreduce" cut
    assert got["plan_torch"] == ["graph:side", "graph:main", "allreduce", "graph:main", "graph:main", "allreduce", "graph:main"]
    for a, b in (("pm_graph", "pm_eager"), ("pc_graph", "pc_eager"), ("pm_graph", "pm_plain"), ("pc_graph", "pc_plain"),
                 ("pm_graph", "pm_torch"), ("pc_graph", "pc_torch")):
        assert torch.equal(got[a], got[b]), (a, b)
    assert torch.isfinite(got["pm_graph"]).all()
