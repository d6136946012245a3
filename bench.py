#!/usr/bin/env python3
"""bench.py -- AR-DAE-VAE train-steps/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload): BASELINE config #2 -- MNISTIPVAE (784 -> z=32, h=256, 2 layers, softplus) with the
mlp-grad conditional AR-DAE (h=256, L=3), GLOBAL batch 512 images x nz_cdae 256 Monte-Carlo rows, synthetic
dynamically-binarised images generated on the device.  One step = the loop body of the reference's train()
(ivae_ardae.py:707-846): one cDAE update (RMSprop) + one VAE update (Adam), nothing skipped.  With N GPUs the image
batch is sharded (512/N per rank, strong scaling), gradients all-reduced over RCCL.

Prints ONE JSON line on rank 0 (contract in the task description) including
  roofline     -- the dominant kernel, timed live with HIP events on the launch stream in a second, instrumented pass
  cpu_baseline -- the oracle's restatement of the same step timed on this box's host cores (bounded sample)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense FP32 matrix peak (v_mfma_f32_32x32x2_f32)
BF16_MFMA_PEAK_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense BF16 matrix peak (v_mfma_f32_32x32x16_bf16: 2027 sustained, scratch/mfma/bf16x6.hip)
HBM_PEAK_GBS = 8000.0


def algorithmic_flops_per_step(B, nz, z=32, h=256, L=3, D=784, nd=100):
    """2*[N*3(F_inp + F_neg + S) + B*3 F_ctx] + sampler forward on N rows (+ the VAE update on B rows)."""
    N = B * nz
    F_inp = z * h + (L - 1) * h * h
    F_ctx = z * h + (L - 1) * h * h
    F_neg = (2 * h + 1) * h + (L - 1) * h * h + h
    S = h + (L - 1) * h * h + h * h + (L - 1) * h * h + z * h
    cdae = 2 * (N * 3 * (F_inp + F_neg + S) + B * 3 * F_ctx)
    sampler = 2 * N * (nd * h + h * z)
    F_vae = D * h + 3 * h * h + (h + nd) * h + h * z + z * h + 2 * h * h + h * D
    return cdae + sampler + 2 * B * 3 * F_vae


def cpu_baseline(steps=5, warm=2):
    """The oracle (a restatement of the reference's op sequence, pinned against it) on this box's host cores.
    Protocol of BASELINE.md section 3: 2 warm-up steps, then 5 timed steps, median."""
    from oracle import ardae_oracle as O
    # a one-GPU box grants 16 host cores (the driver's CPU share); more threads than that only oversubscribe
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    mc = O.ModelCfg("mnist", 784, 100, 256, 32, 2, "softplus")
    cc = O.CdaeCfg("grad", 32, 32, 256, 3)
    tc = O.TrainCfg(nz_cdae=256)
    pm = O.init_params(O.model_param_spec(mc), 0, O.model_init_special(mc))
    pc = O.init_params(O.cdae_param_spec(cc), 1)
    st_m, st_c = {}, {}
    gen = torch.Generator().manual_seed(0)
    p = (torch.rand(784, generator=gen) < 0.2).float() * 0.6 + 0.03
    times = []
    for t in range(warm + steps):
        xc = torch.bernoulli(p.expand(512, -1), generator=gen)
        xv = torch.bernoulli(p.expand(512, -1), generator=gen)
        noise = O.draw_step_noise(mc, tc, 512, gen)
        t0 = time.perf_counter()
        O.train_step(mc, cc, tc, pm, pc, st_m, st_c, xc, xv, noise)
        times.append(time.perf_counter() - t0)
    timed = sorted(times[warm:])
    dt = timed[len(timed) // 2]
    return {"value": 1.0 / dt, "unit": "train-steps/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"median of {steps} full steps (B=512, nz_cdae=256) after {warm} warm-up, oracle/ardae_oracle.py (PyTorch CPU autograd)",
            "s_per_step": dt, "s_per_step_all": [round(t, 3) for t in times[warm:]]}


def kernel_source_hash():
    """sha256 over the kernel sources the loaded library was built from (csrc/*.hip, *.h, include/*.h): profiles/pmc_summary.json carries
    the same hash of the tree its counters were taken from (tools/summarize_profile.py), so stale counters are never quoted."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for d, pats in ((os.path.join(ROOT, "pytorch-ardae-vae_amd", "csrc"), (".hip", ".h")), (os.path.join(ROOT, "include"), (".h",))):
        files += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(pats)]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def kernel_family(name):
    """Template instantiations of one kernel are ONE family: `linear_wide_kernel<8, 4, 2, 2, 0, 0>` -> `linear_wide_kernel`."""
    return name.split("<", 1)[0].strip()


def roofline_of(rep, prof_steps):
    """The `roofline` object of the bench line, from the live HIP-event log of the instrumented steps.

    The dominant kernel is the kernel FAMILY (template arguments stripped) with the largest share of the step; `achieved`
    is that family's algorithmic FLOPs / its summed launch durations.  The time-weighted figure over every MFMA kernel
    family and the single best instantiation are reported beside it, so the headline cannot be one well-tuned kernel that
    is a small share of the step."""
    fam = {}
    for e in rep:
        f = fam.setdefault(kernel_family(e["name"]), {"name": kernel_family(e["name"]), "calls": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
        for k in ("calls", "total_ms", "flops", "bytes"):
            f[k] += e[k]
    fams = sorted(fam.values(), key=lambda e: -e["total_ms"])
    top = fams[0]
    tf = lambda e: e["flops"] / (e["total_ms"] * 1e-3) / 1e12 if e["total_ms"] > 0 else 0.0
    ach = tf(top)
    tot_ms = sum(e["total_ms"] for e in fams)
    tot_fl = sum(e["flops"] for e in fams)
    traffic, traffic_src = None, "profiles/pmc_summary.json is missing"
    doc, meta, here = {}, {}, None
    pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc):      # HBM bytes per launch from the committed rocprofv3 --pmc passes (a STORED profile value, not measured in this run)
        with open(pmc) as f:
            doc = json.load(f)
        meta = doc.get("_meta", {})
        here = kernel_source_hash()
        if meta.get("kernel_source_sha256") != here:
            # the counters were taken from another build of the kernels: say so instead of quoting them
            traffic_src = (f"null: profiles/pmc_summary.json was recorded from kernel sources {str(meta.get('kernel_source_sha256'))[:12]} "
                           f"(git {meta.get('git', '?')}), the loaded library is built from {here[:12]}")
        else:
            # (multi-layer launches log their layer count behind the instantiation, "... x5": the counter rows are keyed by the instantiation)
            per = [(doc[n]["hbm_bytes_per_launch"], e["calls"]) for e in rep for n in [e["name"].split(" x")[0]]
                   if kernel_family(n) == top["name"] and n in doc and doc[n].get("hbm_bytes_per_launch")]
            if per:
                traffic = sum(b * c for b, c in per) / sum(c for _, c in per)
                traffic_src = (f"profiles/pmc_summary.json (stored rocprofv3 --pmc passes of git {meta.get('git', '?')}, same kernel sources as the loaded "
                               "library; launch-weighted mean over the family's instantiations)")
            else:
                traffic_src = "null: no counter rows for the dominant family in profiles/pmc_summary.json"
    # roofline.frac: the dominant family's launch duration from the committed rocprofv3 --kernel-trace pass when it was taken from THIS build
    # of the kernels (same source hash), the live HIP-event brackets otherwise - and the line says which
    frac_source = "live HIP-event brackets around the launches of this run's instrumented steps"
    ach_events = ach
    if os.path.exists(pmc) and meta.get("kernel_source_sha256") == here:
        per = [(doc[n]["trace_avg_us"], e["flops"] / e["calls"], e["calls"]) for e in rep for n in [e["name"].split(" x")[0]]
               if kernel_family(n) == top["name"] and n in doc and doc[n].get("trace_avg_us")]
        if per and sum(c for _, _, c in per) == top["calls"]:
            ach = sum(f * c for _, f, c in per) / sum(us * 1e-6 * c for us, _, c in per) / 1e12
            frac_source = (f"rocprofv3 --kernel-trace average launch durations of profiles/ (git {meta.get('git', '?')}, same kernel sources as the loaded "
                           f"library) x this run's per-launch algorithmic FLOPs; live event brackets of this run give {ach_events:.1f} TFLOP/s")
    single = max((e for e in rep if e["flops"] > 0 and e["total_ms"] > 0.02 * tot_ms), key=tf, default=None)
    return {"bound": "mfma", "kernel": top["name"], "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / FP32_MFMA_PEAK_TFLOPS, "frac_source": frac_source, "achieved_live_events": ach_events,
            "traffic": traffic, "traffic_source": traffic_src,
            # two matrix paths in the step: the N-row layers run v_mfma_f32_32x32x2_f32 (the FP32-MFMA roof above); the 256 x 256 weight
            # gradients form their fp32 products EXACTLY from three bf16 pieces per operand on the BF16 matrix cores (wgrad_x9_kernel:
            # nine v_mfma_f32_32x32x16_bf16 piece products per fp32 product, fp32 accumulation) - its fp32-equivalent rate is bounded by
            # BF16_MFMA_PEAK / 9 and by HBM, not by the FP32-MFMA peak
            "matrix_paths": {"fp32_mfma_peak_tflops": FP32_MFMA_PEAK_TFLOPS, "bf16x9_fp32_equivalent_peak_tflops": BF16_MFMA_PEAK_TFLOPS / 9.0,
                             "bf16x9_kernels": sorted({kernel_family(e["name"]) for e in rep if "x9" in e["name"]})},
            "share_of_step_kernel_time": top["total_ms"] / tot_ms,
            "avg_launch_us": 1e3 * top["total_ms"] / top["calls"], "launches_per_step": top["calls"] / prof_steps,
            "algorithmic_gflop_per_launch": top["flops"] / top["calls"] / 1e9,
            "algorithmic_mbytes_per_launch": top["bytes"] / top["calls"] / 1e6,
            # FLOPs the kernels actually execute per step (per-launch algorithmic FLOPs of every launch of the instrumented steps): less than
            # SURVEY 8(d)'s formula, which counts the W_1c c products on all N rows although the engine computes them once per image
            "executed_tflop_per_step": tot_fl / prof_steps / 1e12,
            "best_single_kernel": None if single is None else {"name": single["name"], "achieved": tf(single), "frac": tf(single) / FP32_MFMA_PEAK_TFLOPS,
                                                                "share_of_step_kernel_time": single["total_ms"] / tot_ms},
            # HIP-event brackets around SINGLE launches of instrumented eager steps: a bracket has a floor of ~20 us, so families of short
            # launches read high here and the list does NOT sum to ms_per_step (which times graph replays); kernel-trace durations of the
            # same kernels are in profiles/r03_kernel_stats.csv
            "families_note": "event-bracketed single launches (floor ~20 us per launch): short-launch families read high; not additive to ms_per_step",
            "families": [{"name": e["name"], "calls_per_step": e["calls"] / prof_steps, "event_ms_per_step": e["total_ms"] / prof_steps,
                          "event_avg_launch_us": 1e3 * e["total_ms"] / e["calls"], "tflops": tf(e) if 1e3 * e["total_ms"] / e["calls"] >= 40.0 else None}
                         for e in fams[:int(os.environ.get("BENCH_TOPK", "8"))]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-steps", type=int, default=5, help="instrumented steps for the live roofline numbers")
    args = ap.parse_args()

    # ---- process environment FIRST: HSA / RCCL read these when the runtime initialises, i.e. at the first call that touches the GPU
    # (torch.cuda.set_device below) - a default set after that point never takes effect in this process
    env_defaults = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}     # the host driver only supports dmabuf IPC (RCCL peer buffers over xGMI)
    gpu_was_up = torch.cuda.is_initialized()
    for k, v in env_defaults.items():
        os.environ.setdefault(k, v)
    env_report = {"set_before_device_init": not gpu_was_up, **{k: os.environ[k] for k in env_defaults},
                  **{k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_"))}}

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    knobs = os.environ.get("ARDAE_DEBUG_KNOBS") == "1"
    # Rehearsal knobs (tests/test_dp_gpu.py; honoured only with ARDAE_DEBUG_KNOBS=1, like every experiment switch): BENCH_ONE_GPU=1 puts
    # every rank on cuda:0 and BENCH_BACKEND=gloo moves the gradient buffers through the host, because RCCL refuses two ranks on one
    # device; BENCH_FORCE_DP=1 runs the data-parallel plan (process group, RCCL communicator, captured all-reduces) with ONE rank;
    # BENCH_GLOBAL_B: another global batch (the per-rank shards of the multi-GPU runs on one GPU).  The driver's runs use none of them.
    knob = lambda name, default: os.environ.get(name, default) if knobs else default
    backend = knob("BENCH_BACKEND", "nccl")
    force_dp = knob("BENCH_FORCE_DP", "0") == "1"
    dev_index = 0 if (world == 1 or knob("BENCH_ONE_GPU", "0") == "1") else local_rank
    torch.cuda.set_device(dev_index)
    distributed = world > 1 or (force_dp and "RANK" in os.environ)
    if distributed:
        if backend == "nccl":
            torch.distributed.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            torch.distributed.init_process_group(backend=backend)
        assert torch.distributed.get_world_size() == world and torch.distributed.get_rank() == rank
    dev = torch.device("cuda", dev_index)

    import ardae_amd as net
    from ardae_amd import _lib as L
    import ctypes

    GLOBAL_B, NZ = int(knob("BENCH_GLOBAL_B", "512")), 256
    assert GLOBAL_B % world == 0, "the global batch must divide over the ranks"
    B = GLOBAL_B // world
    torch.manual_seed(0)                                  # identical parameters on every rank
    model = net.MNISTIPVAE(input_dim=784, noise_dim=100, h_dim=256, num_hidden_layers=2, nonlinearity="softplus",
                           enc_type="concat", z_dim=32).to(dev)
    cdae = net.MLPGradCARDAE(input_dim=32, context_dim=32, std=1., h_dim=256, num_hidden_layers=3, nonlinearity="softplus",
                             noise_type="gaussian", enc_ctx=True, enc_input=True).to(dev)
    g = torch.Generator(device="cpu").manual_seed(1234)
    pimg = ((torch.rand(784, generator=g) < 0.2).float() * 0.6 + 0.03).to(dev)
    lib = L.lib()
    state = {"i": 0}
    init_m, init_c = model.flat_params().clone(), cdae.flat_params().clone()

    def make_engine(**kw):
        with torch.no_grad():                             # (a second attempt starts from the same parameters)
            model.flat_params().copy_(init_m); cdae.flat_params().copy_(init_c)
        e = net.ArdaeEngine(model, cdae, net.TrainConfig(nz_cdae=NZ), batch_size=B, force_dp=force_dp and distributed, **kw)
        net.manual_seed(42)                               # one noise stream: every rank draws its rows of the global draw (engine._normal)
        state["i"] = 0
        return e

    def barrier():
        if distributed:
            torch.distributed.barrier()

    # Several ranks: the step with its two RCCL all-reduces CAPTURED in the graphs first (`ardae_dp_allreduce_mean`; what one GPU can test
    # of it is tested).  Should capturing a collective be refused on the machine at hand, every rank falls back - together: the outcome is
    # agreed on with an all-reduce - to the form whose pieces all have run in the suite, graphs cut at the collectives with
    # torch.distributed between them, and the line says so (`dp_fallback`).
    dp_fallback = None
    for attempt, kw in enumerate(({}, {"dp_comm": None}, {"dp_comm": None, "graph": False})):
        eng = make_engine(**kw)
        (xc,), xv = eng.input_buffers(1)                  # the binarisation kernel writes the engine's static batch buffers: step() copies nothing

        def one_step():
            # dynamic binarisation on the device (datasets/mnist.py:36-40): two fresh batches per step
            i = state["i"]; state["i"] += 1
            L.check(lib.ardae_bernoulli(L.ptr(pimg), B, 784, L.ptr(xc), ctypes.c_uint64(1000 + rank), ctypes.c_uint64(2 * i), L.stream_ptr()))
            L.check(lib.ardae_bernoulli(L.ptr(pimg), B, 784, L.ptr(xv), ctypes.c_uint64(1000 + rank), ctypes.c_uint64(2 * i + 1), L.stream_ptr()))
            eng.step(xc, xv)

        err = None
        try:
            if knob("BENCH_INJECT_FAILURE", "") == str(attempt):      # rehearsal of the fallback (tests/test_dp_gpu.py)
                raise RuntimeError("injected failure of attempt %d" % attempt)
            for _ in range(args.warmup):                  # (the capture happens at the engine's third step)
                one_step()
            torch.cuda.synchronize()
        except Exception as exc:                          # noqa: BLE001 - anything a refused capture can raise
            if not distributed or attempt == 2:
                raise
            err = f"{type(exc).__name__}: {exc}"
        if not distributed:
            break
        flag = torch.tensor([0.0 if err is None else 1.0], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if float(flag) == 0.0:
            break
        dp_fallback = f"attempt {attempt} ({kw or 'captured RCCL all-reduces'}) failed on some rank" + (f": {err}" if err else "")
        sys.stderr.write(f"[bench.py rank {rank}] {dp_fallback}; falling back\n")
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    rank_report = None
    if distributed:
        # per-rank wall times of the timed region (the headline takes their MAX) and the latency of the step's two collectives,
        # so that a multi-GPU line explains itself: compute per rank vs. time spent in the gradient all-reduces
        mine = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        t = mine.clone()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
        from ardae_amd import dist as D
        reduce_ = eng.comm.allreduce_mean_ if eng.comm is not None else D.allreduce_mean_      # the call the step itself makes
        lat = {}
        for name, buf in (("cdae_grads", eng.grads_c[:eng.n_c]), ("model_grads", eng.grads_m)):
            scratch = buf.clone()
            for _ in range(3):
                reduce_(scratch)
            torch.cuda.synchronize(); torch.distributed.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                reduce_(scratch)
            e1.record(); torch.cuda.synchronize()
            lat[name] = {"bytes": scratch.numel() * 4, "us_per_allreduce_mean": 1e3 * e0.elapsed_time(e1) / 20}
        seen = eng.comm.query() if eng.comm is not None else None
        rank_report = {"ms_per_step_by_rank": [1e3 * float(x) / args.steps for x in every], "collectives": lat,
                       "plan": eng.plan_summary(),
                       # who took part: torch.distributed's view and, when the C ABI owns the communicator, RCCL's own
                       "torch_distributed": {"backend": str(torch.distributed.get_backend()), "world_size": torch.distributed.get_world_size()},
                       "rccl": None if seen is None else {"library": eng.comm.backend, "ranks": seen[0], "rank": seen[1], "device": seen[2],
                                                          "allreduce": "ardae_dp_allreduce_mean, captured in the step graphs"},
                       "dp_fallback": dp_fallback, "env": env_report}
    stats = eng.stats()

    # ---- live per-kernel timing (HIP events on the launch stream), separate from the throughput region
    roofline = None
    rep = None
    if args.prof_steps > 0:
        # every rank runs the instrumented steps (they contain the gradient all-reduces); rank 0 reports its own kernels
        graph_was = eng.use_graph
        eng.use_graph = False                 # the per-kernel HIP events need individual launches
        lib.ardae_profile_enable(1)
        for _ in range(args.prof_steps):
            one_step()
        torch.cuda.synchronize()
        rep = L.profile_report()
        lib.ardae_profile_enable(0)
        eng.use_graph = graph_was
    if rank == 0 and rep:
        roofline = roofline_of(rep, args.prof_steps)
    barrier()

    if rank == 0:
        steps_per_s = args.steps / dt
        flop = algorithmic_flops_per_step(GLOBAL_B, NZ)
        executed = roofline["executed_tflop_per_step"] * 1e12 * world if roofline else None      # rank 0's kernels x ranks (equal shards)
        out = {
            "metric": f"AR-DAE-VAE train-steps/sec (batch {GLOBAL_B}, nz_cdae {NZ}) on dbMNIST", "value": steps_per_s, "unit": "train-steps/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE config #2" if GLOBAL_B == 512 else f"EXPERIMENT (BENCH_GLOBAL_B={GLOBAL_B}): BASELINE config #2's networks") +
                                   f": MNISTIPVAE mlp z=32 h=256 + cDAE mlp-grad h=256 L=3, global batch {GLOBAL_B}, "
                                   f"nz_cdae {NZ} ({GLOBAL_B * NZ} rows), dynamic binarisation on device, 1 cDAE + 1 VAE update per step",
                       "global_batch": GLOBAL_B, "nz_cdae": NZ, "per_gpu_batch": B, "parallelism": f"dp{world}"},
            # two whole-step figures: FLOPs the kernels EXECUTE (the honest utilisation), and SURVEY 8(d)'s formula, which also counts
            # work the reference performs on all N rows but the engine hoists to once per image (W_1c c and its weight gradient)
            "executed_tflop_per_step": None if executed is None else executed / 1e12,
            "whole_step_tflops_executed": None if executed is None else executed * steps_per_s / 1e12,
            "whole_step_frac_executed": None if executed is None else executed * steps_per_s / 1e12 / (FP32_MFMA_PEAK_TFLOPS * world),
            "whole_step_frac_note": "fp32-equivalent FLOPs of the whole step against the FP32-MFMA peak; since round 4 the 256 x 256 weight gradients "
                                    "(33 % of those FLOPs) run on the BF16 matrix cores (bf16x9), which that peak does not bound",
            "algorithmic_tflop_per_step": flop / 1e12,
            "algorithmic_tflop_note": "SURVEY 8(d) formula, incl. per-image-hoisted work the reference performs on every row",
            "whole_step_tflops": flop * steps_per_s / 1e12,
            "whole_step_frac_of_fp32_mfma_peak": flop * steps_per_s / 1e12 / (FP32_MFMA_PEAK_TFLOPS * world),
            "losses": stats,
            "hip_graph": bool(eng._graph is not None),
            "ranks": rank_report,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (contract)
            out["cpu_baseline"] = cpu_baseline()
            out["speedup_vs_cpu_baseline"] = steps_per_s / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:      # every rank: a failure must end the job with a non-zero status, whichever rank it happens on
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        import traceback
        sys.stderr.write(f"[bench.py rank {os.environ.get('RANK', '0')}] failed:\n{traceback.format_exc()}")
        sys.stderr.flush()
        os._exit(1)                 # no interpreter teardown: a rank stuck in a collective's destructor would keep the job alive
