"""Fused train step: the loop body of the reference's `train()` (ivae_ardae.py:707-846) as a straight line of C-ABI calls.

No autograd graph, no per-step allocation, no host synchronisation: every buffer is created once, noise comes from the
engine's Philox stream (or is injected for parity tests), losses stay on the device until `.stats()` is called.
Data parallel: the image batch is sharded over ranks, parameters are replicated, and the two flat gradient
buffers are all-reduced (RCCL over xGMI via torch.distributed backend "nccl") before their optimiser steps.
"""
import ctypes
import os
from dataclasses import dataclass

import torch

from . import _lib as L
from . import dist
from . import rng


@dataclass
class TrainConfig:
    """Loop constants (argparse defaults / run_vae_dbmnist.sh:36-37, run_vae_25gaussians.sh:3-12)."""
    delta: float = 0.1            # --delta
    std_scale: float = 1e4        # --std-scale
    nz_cdae: int = 256            # --train-nz-cdae
    nstd_cdae: int = 1            # --train-nstd-cdae: noise levels (sigma, eps pairs) per sample row of the cDAE update
    nz_model: int = 1             # --train-nz-model
    num_cdae_updates: int = 1     # --num-cdae-updates
    beta: float = 1.0             # --beta-fin (no annealing in the shipped recipes)
    m_lr: float = 1e-4            # --m-lr, Adam betas (m_beta1, 0.999)
    m_beta1: float = 0.5
    d_lr: float = 1e-4            # --d-lr, RMSprop momentum d_momentum
    d_momentum: float = 0.5
    m_optimizer: str = "adam"     # --m-optimizer / --d-optimizer: sgd | adam | amsgrad | rmsprop (ivae_ardae.py:545-556,612-622); the shipped
    d_optimizer: str = "rmsprop"  # recipes pass adam / rmsprop.  The model's RMSprop takes d_momentum, as the reference's does (:553)
    d_beta1: float = 0.5          # --d-beta1 (cDAE Adam / amsgrad)
    cdae_ctx_type: str = "lt0"    # --cdae-ctx-type: "lt0" (context = encode(x, std=0)) | "hidden1a" (aux models: encoder hiddens) | "data"
    ctx_data_center: bool = True  # "data": the image as 2x - 1 when 'mnist' is in --dataset (ivae_ardae.py:731-734,810-813), x itself otherwise


def annealing_func(val_init, val_fin, val_annealing, step):
    """utils/msc.py:53-55."""
    if val_annealing is None:
        return float(val_fin)
    return float(val_init + (val_fin - val_init) / float(val_annealing) * float(min(val_annealing, step)))


class _FlatOpt:
    """One network's optimiser on its flat parameter / gradient buffers: torch.optim.SGD(lr), the reference's vendored Adam (amsgrad
    optional; utils/optim.py:49-108) or torch.optim.RMSprop(momentum) - the four choices of --m-optimizer / --d-optimizer.  `n`: floats
    that receive gradients (the cDAE's trailing neglogprob.fc.bias does not and keeps no state).  Adam's t and bias corrections live in a
    32-byte device block (`ardae_step_state_advance`) so that a captured step can be replayed."""
    KINDS = ("sgd", "adam", "amsgrad", "rmsprop")

    def __init__(self, kind, flat, n, lr, beta1, momentum, state=None):
        if kind not in self.KINDS:
            raise NotImplementedError(f"unknown optimizer: {kind}")                     # ivae_ardae.py:555-556,621-622
        self.kind, self.flat, self.n, self.lr, self.beta1, self.momentum = kind, flat, int(n), float(lr), float(beta1), float(momentum)
        z = lambda: torch.zeros_like(flat)
        self.a = None if kind == "sgd" else z()                                         # exp_avg | square_avg
        self.b = None if kind == "sgd" else z()                                         # exp_avg_sq | momentum_buffer
        self.c = z() if kind == "amsgrad" else None                                     # max_exp_avg_sq
        self.steps = 0
        self.state = state if state is not None else torch.zeros(4, dtype=torch.int64, device=flat.device)

    @property
    def adam(self):
        return self.kind in ("adam", "amsgrad")

    def advance(self, lib, rng_inc=0):
        L.check(lib.ardae_step_state_advance(ctypes.c_void_p(self.state.data_ptr()), ctypes.c_uint64(rng_inc), self.lr, self.beta1, 0.999,
                                             L.stream_ptr()), "ardae_step_state_advance")

    def apply(self, lib, grads, in_step):
        st, p, g = L.stream_ptr(), L.ptr(self.flat), L.ptr(grads)
        if self.kind == "sgd":
            L.check(lib.ardae_sgd_step(p, g, self.n, self.lr, st), "ardae_sgd_step")
        elif self.kind == "rmsprop":
            L.check(lib.ardae_rmsprop_step(p, g, L.ptr(self.a), L.ptr(self.b), self.n, self.lr, 0.99, 1e-8, self.momentum, st), "ardae_rmsprop_step")
        elif in_step:      # t and the bias corrections come from the device block (advanced inside the step)
            L.check(lib.ardae_adam_ref_step_dev(p, g, L.ptr(self.a), L.ptr(self.b), L.ptr(self.c) if self.c is not None else None, self.n,
                                                self.beta1, 0.999, 1e-8, ctypes.c_void_p(self.state.data_ptr()), st), "ardae_adam_ref_step_dev")
        else:
            L.check(lib.ardae_adam_ref_step(p, g, L.ptr(self.a), L.ptr(self.b), L.ptr(self.c) if self.c is not None else None, self.n,
                                            self.lr, self.beta1, 0.999, 1e-8, self.steps + 1, st), "ardae_adam_ref_step")

    # torch.optim.Optimizer.state_dict() pieces (per-parameter views of the flat buffers)
    def state_names(self):
        return {"sgd": (), "adam": ("exp_avg", "exp_avg_sq"), "amsgrad": ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"),
                "rmsprop": ("square_avg", "momentum_buffer")}[self.kind]

    def buffers(self):
        return [t for t in (self.a, self.b, self.c) if t is not None]

    def param_group(self, nparams):
        if self.kind == "sgd":
            g = {"lr": self.lr, "momentum": 0, "dampening": 0, "weight_decay": 0, "nesterov": False}
        elif self.adam:
            g = {"lr": self.lr, "betas": (self.beta1, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": self.kind == "amsgrad"}
        else:
            g = {"lr": self.lr, "momentum": self.momentum, "alpha": 0.99, "eps": 1e-8, "centered": False, "weight_decay": 0}
        g["params"] = list(range(nparams))
        return g


class ArdaeEngine:
    """`graph=True` (default): `step()` captures the iteration at its third call and replays it afterwards.  What changes from
    step to step lives in device memory: a 32-byte step state (Philox base offset, Adam's t and bias corrections,
    `ardae_step_state_advance`) and the static image buffers the caller's batches are copied into.  Noise injection (parity
    tests) runs the same launches eagerly.

    A step is a PLAN of units (`_plan` / `_units`): stretches of launches on ONE stream each, ordered by events between
    them, with the gradient all-reduces (world > 1) as eager items in between.  Every unit is captured as its own LINEAR HIP
    graph: ROCm submits a single-stream graph as one batch of AQL packets (measured on MI355X / ROCm 7.2: 3.3 us of host time
    per node), while a graph with a forked stream is enqueued node by node at the cost of eager launches (9.3 us per node:
    the 8-rank shard of config #2, 113 launches in 1.2 ms, was HOST-bound in round 2, and the side branch only reached its
    queue 0.7 ms into the step).  Concurrency between the VAE forward half and the cDAE phase therefore comes from two
    linear graphs on two streams, not from a fork inside one graph; the same units run eagerly when graphs are off, so
    replayed == eager bit for bit, for any world size and any backend (RCCL over xGMI in `bench.py --gpus N`, gloo in the
    rehearsal tests).  `graph=True` makes a refused capture an error; `graph="auto"` falls back to eager launches with a warning."""

    RNG_STRIDE = 16   # Philox offsets reserved per step (draws use base + 0, 1, 2, ...)

    def __init__(self, model, cdae, cfg: TrainConfig, batch_size, process_group=None, graph=True, force_dp=False, dp_comm="auto"):
        model._require_gpu()
        cdae._require_gpu()
        self.model, self.cdae, self.cfg = model, cdae, cfg
        self.B = int(batch_size)                       # per-rank image batch
        self.dev = model._flat.device
        self.lib = L.lib()
        self.pg = process_group
        self.world = dist.world_size(process_group)
        self.rank = dist.rank(process_group)
        # force_dp: run the data-parallel plan (graphs cut at the gradient all-reduces, the collectives issued between them) even
        # with ONE rank - exercises the RCCL path on a single GPU (tests/test_dp_gpu.py)
        self.dp = self.world > 1 or bool(force_dp)
        # The gradient exchange (SURVEY 8(b) `dp_allreduce_flat`): an RCCL communicator behind the C ABI (`dist.DpComm`) whose all-reduce
        # is a stream-ordered call like any kernel launch and is CAPTURED into the step's graphs - the multi-rank step is then the same
        # three linear graphs as the single-rank one.  dp_comm: a `dist.DpComm`, None (torch.distributed's collectives as eager items
        # between the graphs: the gloo rehearsal, two ranks sharing a GPU), or "auto": a communicator when the process group is RCCL
        # ("nccl" backend), none otherwise.
        if isinstance(dp_comm, str):
            if dp_comm != "auto":
                raise ValueError(f"dp_comm must be a dist.DpComm, None or 'auto', got {dp_comm!r}")
            dp_comm = None
            if self.dp and torch.distributed.is_available() and torch.distributed.is_initialized() and \
                    "nccl" in str(torch.distributed.get_backend(process_group)):
                dp_comm = dist.DpComm(process_group)
        self.comm = dp_comm if self.dp else None
        if self.comm is not None and (self.comm.world, self.comm.rank) != (self.world, self.rank):
            raise ValueError(f"dp_comm is rank {self.comm.rank} of {self.comm.world}, the process group says {self.rank} of {self.world}")
        md, cd = model._desc, cdae._desc
        B, nzc, nzm = self.B, cfg.nz_cdae, cfg.nz_model
        N = B * nzc
        z, nd = model.z_dim, model._noise_width          # floats per row of a sampler draw (aux models: [eps0 | eps])
        if cfg.cdae_ctx_type not in ("lt0", "hidden1a", "data"):
            raise NotImplementedError(f"cdae_ctx_type {cfg.cdae_ctx_type!r}")          # ivae_ardae.py:743-744
        if cfg.cdae_ctx_type == "hidden1a" and not model.hidden_dim:
            raise NotImplementedError("hidden1a is the aux models' context (ivae_ardae.py:572-580)")
        self.hidden_ctx = cfg.cdae_ctx_type == "hidden1a"
        self.data_ctx = cfg.cdae_ctx_type == "data"
        # MNISTResConvAuxIPVAEClipped: the two std = 0 calls that open each phase are RANDOM draws (z0 keeps an unscaled eps0,
        # ivae/auxresconv2.py:91) - one for the context, one for the latent mean - so they cannot share a pass
        self.clipped = bool(getattr(model, "_clipped", False))
        if self.clipped and not self.hidden_ctx:
            raise NotImplementedError("MNISTResConvAuxIPVAEClipped is built with --cdae-ctx-type hidden1a (the aux models' context, ivae_ardae.py:572-580)")
        ctx_dim = model.hidden_dim if self.hidden_ctx else int(model.input_dim) if self.data_ctx else z
        if int(cdae.context_dim) != ctx_dim:
            raise ValueError(f"cdae.context_dim = {cdae.context_dim}, but the {cfg.cdae_ctx_type} context has {ctx_dim} columns")
        f = lambda *s: torch.empty(*s, device=self.dev, dtype=torch.float32)
        lib = self.lib
        S = nzc * int(cfg.nstd_cdae)            # rows per image of the cDAE update (ivae_ardae.py:759-767)
        ws_floats = max(lib.ardae_cdae_workspace_floats(ctypes.byref(cd), B, S, 1),
                        lib.ardae_model_workspace_floats(ctypes.byref(md), B, nzc, 3))
        self.ws = f(ws_floats)
        self.ws_vae = f(lib.ardae_model_workspace_floats(ctypes.byref(md), B, nzm, 1))
        self.ws_small = f(max(lib.ardae_cdae_workspace_floats(ctypes.byref(cd), B, nzm, 0),
                              lib.ardae_model_workspace_floats(ctypes.byref(md), B, 1, 0)))
        self.ws_small_v = f(lib.ardae_model_workspace_floats(ctypes.byref(md), B, 1, 0))   # VAE-side encode(std=0): may run beside the cDAE phase
        self.z0, self.latent = f(B, z), f(N, z)
        self.noise_s, self.xi, self.eps = f(model._noise_numel(B, nzc)), f(B * S), f(B * S, z)
        self.xbar, self.sigma, self.std_b = f(B * S, z), f(B * S), f(B)
        self.noise_v, self.zv, self.z0v, self.u, self.g = f(model._noise_numel(B, nzm)), f(B * nzm, z), f(B, z), f(B * nzm, z), f(B * nzm, z)
        self.sigma0 = torch.zeros(B * nzm, device=self.dev)
        if self.clipped:        # [context draw | latent-mean draw] of a phase's two std = 0 calls, [2, B, z0_dim]
            self.raw_c, self.raw_v = f(2, B, model.noise_dim), f(2, B, model.noise_dim)
        self.ctx_c, self.ctx_v = (f(B, ctx_dim), f(B, ctx_dim)) if (self.hidden_ctx or self.data_ctx) else (self.z0, self.z0v)
        if self.data_ctx:      # 2x - 1 = 2 (x - 1/2) through ardae_center_scale; uncentred: x - 0
            self._ctx_half = torch.full((B, ctx_dim), 0.5 if cfg.ctx_data_center else 0.0, device=self.dev)
        self.loss_c, self.losses_m = f(1), f(3)
        self.grads_c = torch.zeros_like(cdae._flat)
        self.grads_m = torch.zeros_like(model._flat)
        # optimiser state (flat; the cDAE's last tensor, neglogprob.fc.bias, gets no gradient in the reference and is skipped)
        self.n_c = cdae._flat.numel() - (1 if cdae._kind == "grad" else 0)
        self.step_count = 0
        # device-resident step state (Philox base offset + the model optimiser's Adam block) + graph bookkeeping
        self.state = torch.zeros(4, dtype=torch.int64, device=self.dev)
        # --m-optimizer / --d-optimizer (ivae_ardae.py:545-556,612-622); the model's RMSprop is built with d_momentum there (:553)
        self.opt_m = _FlatOpt(cfg.m_optimizer, model._flat, model._flat.numel(), cfg.m_lr, cfg.m_beta1, cfg.d_momentum, state=self.state)
        self.opt_c = _FlatOpt(cfg.d_optimizer, cdae._flat, self.n_c, cfg.d_lr, cfg.d_beta1, cfg.d_momentum)
        if graph not in (True, False, "auto"):
            raise ValueError(f"graph must be True, False or 'auto', got {graph!r}")
        self.use_graph = bool(graph) and L.debug_knob("ARDAE_GRAPH", "1") != "0"
        self.graph_strict = graph is True       # a refused capture is an error (bench.py must not silently time eager launches)
        self._graph, self._graph_key, self._xc, self._xv = None, None, None, None      # _graph: the captured plan; _xc: static batch buffers
        self._in_step, self._draws, self._warmed = False, 0, False
        self._last_beta, self._beta_stable = None, 0
        # The first half of the VAE update (sampler + decoder + ELBO pieces on the VAE batch, ~25 per-image launches) needs
        # nothing from the cDAE update: step() runs it on a side stream next to the cDAE phase's N-row kernels.
        self.overlap = L.debug_knob("ARDAE_OVERLAP", "1") != "0"
        # MLP models: the decoder half of the VAE backward (down to dL/dz) joins the forward half on the side stream
        self.split_backward = int(md.kind) < 2 and L.debug_knob("ARDAE_SPLIT_BACKWARD", "1") != "0"
        self._side = torch.cuda.Stream(device=self.dev) if self.overlap else None
        self.fused_draws = L.debug_knob("ARDAE_FUSED_DRAW", "1") != "0"     # sigma / eps draws inside the perturbation kernel
        self.fused_first_layer = L.debug_knob("ARDAE_FUSED_A1", "1") != "0"   # ... and the score network's first layer on the perturbed rows
        self._cap_stream = torch.cuda.Stream(device=self.dev)
        self._stamps = None                             # diagnostics: see enable_stamps()
        self._log = None                                # scalar log channel (scalar_log.ScalarLog), one more launch at the end of the step
        self.opt_m.advance(self.lib, self.RNG_STRIDE)   # the step state always describes the COMING step (t = 1, first Philox block)
        self.repack()

    def attach_log(self, log):
        """Append `log.record(beta)` to every step (a new launch: a captured step graph is rebuilt)."""
        self._log, self._graph = log, None

    # ------------------------------------------------------------------------------------------------------------
    def repack(self):
        self.model._packed = None
        self.cdae._packed = None
        self.pk_m = self.model._packed_weights()
        self.pk_c = self.cdae._packed_weights()

    def _pack_model(self):
        L.check(self.lib.ardae_model_pack(ctypes.byref(self.model._desc), L.ptr(self.model._flat), L.ptr(self.pk_m), L.stream_ptr()))

    def _pack_cdae(self):
        L.check(self.lib.ardae_cdae_pack(ctypes.byref(self.cdae._desc), L.ptr(self.cdae._flat), L.ptr(self.pk_c), L.stream_ptr()))

    def _check_batch(self, x, what):
        """The kernels read exactly B * input_dim contiguous floats from each batch pointer: anything else (a ragged last batch
        of a loader without drop_last, a strided view, a host tensor) must be refused here, not read out of bounds."""
        if not torch.is_tensor(x):
            raise TypeError(f"{what}: expected a tensor, got {type(x).__name__}")
        if x.dtype != torch.float32:
            raise ValueError(f"{what}: expected a float32 tensor, got {x.dtype}")
        if x.dim() < 2 or x.size(0) != self.B or x.numel() != self.B * self.model.input_dim:
            raise ValueError(f"{what}: expected {self.B} images of {self.model.input_dim} values (the engine was built with batch_size={self.B}; "
                             f"use drop_last or pad the last batch), got shape {tuple(x.shape)}")
        if not x.is_contiguous():
            raise ValueError(f"{what}: the batch must be contiguous (got strides {tuple(x.stride())}); call .contiguous()")
        if not x.is_cuda or x.device != self.dev:
            raise ValueError(f"{what}: expected a tensor on {self.dev}, got one on {x.device}")

    def _encode(self, x, noise, nz, out, ws):
        L.check(self.lib.ardae_model_encode(ctypes.byref(self.model._desc), L.ptr(self.model._flat), L.ptr(self.pk_m), L.ptr(x),
                                            L.ptr(noise) if noise is not None else None, self.B, nz, L.ptr(ws), ws.numel(),
                                            L.ptr(out), L.stream_ptr()), "ardae_model_encode")

    def _hidden(self, x, z0_out, out, ws, raws=None):
        """The std = 0 pass of an aux sampler: latent mean z0 AND the hidden1a context in one go.  raws [2, B, z0_dim] (clipped class):
        two passes - the context with draw 0, the latent mean with draw 1."""
        if self.clipped:
            md, st = self.model._desc, L.stream_ptr()
            for k, (zo, ho) in enumerate(((None, out), (z0_out, None))):
                L.check(self.lib.ardae_model_encode_hidden_raw(ctypes.byref(md), L.ptr(self.model._flat), L.ptr(self.pk_m), L.ptr(x), L.ptr(raws[k]), self.B,
                                                               L.ptr(ws), ws.numel(), None if zo is None else L.ptr(zo), None if ho is None else L.ptr(ho), st),
                        "ardae_model_encode_hidden_raw")
            return
        L.check(self.lib.ardae_model_encode_hidden(ctypes.byref(self.model._desc), L.ptr(self.model._flat), L.ptr(self.pk_m), L.ptr(x), self.B,
                                                   L.ptr(ws), ws.numel(), L.ptr(z0_out), L.ptr(out), L.stream_ptr()), "ardae_model_encode_hidden")

    # ------------------------------------------------------------------------------------------------------------
    # The step as a plan.  A segment is ("run", name, stream, deps, fn) or ("allreduce", tensor); `_units` merges neighbouring
    # segments of a stream into units (= what one linear graph holds).  A unit ends where another stream waits for it.
    def _plan(self, xs, x_vae, noise, beta):
        cfg = self.cfg
        vae_draw = self.RNG_STRIDE - 1            # the VAE sampler's noise keeps its own offset whatever the launch order
        # injected noise: one dict for everything, or a list with one dict per cDAE update (the last one also holds "vae")
        nlist = list(noise) if isinstance(noise, (list, tuple)) else [noise] * len(xs)
        nv = nlist[-1]["vae"] if nlist[-1] else self.noise_v
        side = "side" if self.overlap else "main"
        # (the device step state is advanced at the END of a step, for the next one: nothing stands between the start of a step and
        # the two streams' first launches)
        segs = []
        vae_fwd = ("run", "vae_fwd", side, (), lambda: self.vae_forward_part(x_vae, nlist[-1], beta, draw=vae_draw))
        if self.overlap:
            # first in launch order: its ~30 per-image launches run beside the head of the cDAE phase (per-image launches too),
            # before the N-row kernels take every CU
            segs.append(vae_fwd)
        for i, xc in enumerate(xs):
            segs.append(("run", f"cdae_grads{i}", "main", (), lambda xc=xc, i=i: self._cdae_grads(xc, nlist[i], 3 * i)))
            if self.dp:
                segs.append(self._allreduce_seg(f"allreduce_c{i}", self.grads_c[:self.n_c]))
            segs.append(("run", f"cdae_update{i}", "main", (), self._cdae_update))
        if not self.overlap:
            segs.append(vae_fwd)
        segs.append(("run", "vae_bwd", "main", ("vae_fwd",) if self.overlap else (), lambda: self._vae_backward_grads(x_vae, nv, beta)))
        if self.dp:
            segs.append(self._allreduce_seg("allreduce_m", self.grads_m))

        def model_update():
            self._model_update()
            if self._log is not None:
                self._log.record(cfg.beta if beta is None else beta)
            self.opt_m.advance(self.lib, self.RNG_STRIDE)       # for the NEXT step: Philox base += stride, model optimiser's t += 1
        segs.append(("run", "model_update", "main", (), model_update))
        if self._stamps is not None:
            def wrap(name, fn):
                def run():
                    self._stamp(name + " >")
                    fn()
                    self._stamp(name + " <")
                return run
            segs = [s if s[0] != "run" else (s[0], s[1], s[2], s[3], wrap(s[1], s[4])) for s in segs]
        return segs

    def _allreduce_seg(self, name, buf):
        """The gradient mean over the ranks as a plan segment: a launch on the main stream when the C ABI owns the communicator (captured
        with its neighbours), an eager item between the graphs otherwise (torch.distributed)."""
        if self.comm is not None:
            return ("run", name, "main", (), lambda: self.comm.allreduce_mean_(buf))
        return ("allreduce", buf)

    def _allreduce(self, buf):
        """Outside the plan (phase calls made directly)."""
        if self.comm is not None:
            self.comm.allreduce_mean_(buf)
        else:
            dist.allreduce_mean_(buf, self.pg, force=True)

    @staticmethod
    def _units(segs):
        """[("run", stream, wait_for (unit indices on other streams), [fns], record (bool)) | ("allreduce", tensor)] in launch order."""
        needed = {d for s in segs if s[0] == "run" for d in s[3]}          # segments another stream waits for
        units, where, open_unit = [], {}, {}                                # where: segment name -> unit index; open_unit: stream -> index
        for s in segs:
            if s[0] == "allreduce":
                units.append(["allreduce", s[1]])
                open_unit.clear()
                continue
            _, name, stream, deps, fn = s
            waits = sorted({where[d] for d in deps if units[where[d]][1] != stream})
            k = open_unit.get(stream)
            if k is None or waits:
                units.append(["run", stream, waits, [], False])
                k = len(units) - 1
            units[k][3].append(fn)
            where[name] = k
            open_unit[stream] = None if name in needed else k               # a unit somebody waits for ends here
            if name in needed:
                units[k][4] = True
        return units

    def _run_units(self, units, graphs=None, capture=False):
        """Launch a plan: eagerly (graphs None), capturing every unit into its own linear graph (capture=True: returns them),
        or replaying captured graphs."""
        main = torch.cuda.current_stream()
        streams = {"main": main, "side": self._side}
        events, out = {}, []
        for k, u in enumerate(units):
            if u[0] == "allreduce":
                dist.allreduce_mean_(u[1], self.pg, force=self.dp)
                out.append(None)
                continue
            _, sname, waits, fns, record = u
            st = streams[sname]
            for w in waits:
                st.wait_event(events[w])
            if sname != "main" and not waits:        # a side unit without explicit dependencies still follows what main has queued
                st.wait_stream(main)
            if capture:
                g = torch.cuda.CUDAGraph()
                # (captured on a stream of its own - the default stream cannot capture -, replayed on the unit's stream)
                with torch.cuda.graph(g, stream=self._cap_stream):
                    for fn in fns:
                        fn()
                out.append(g)
                with torch.cuda.stream(st):
                    g.replay()
            elif graphs is not None:
                if st is main:
                    graphs[k].replay()
                else:
                    with torch.cuda.stream(st):
                        graphs[k].replay()
            elif st is main:
                for fn in fns:
                    fn()
            else:
                with torch.cuda.stream(st):
                    for fn in fns:
                        fn()
            if record:
                events[k] = torch.cuda.Event()
                events[k].record(st)
        return out

    def _normal(self, out, draw=None):
        """One standard-normal draw.  Inside `step()` the offset comes from the device step state (graph-replayable: the block is
        advanced by the last launch of every step, for the next one);
        phase calls made directly use the host-side stream of `rng`.  `draw`: fixed index of the draw inside the step
        (so that the numbers do not depend on the order in which concurrent parts of the step are launched)."""
        first = self.rank * out.numel()        # equal shards: this rank's rows of the global draw (independent of the rank count)
        if not self._in_step:
            return rng.normal(None, self.dev, out=out, first_element=first)
        if draw is None:
            k = self._draws
            self._draws += 1
        else:
            k = draw
        if k >= self.RNG_STRIDE:
            raise RuntimeError("more Philox draws in one step than RNG_STRIDE reserves")
        L.check(self.lib.ardae_philox_normal_at(L.ptr(out), out.numel(), ctypes.c_uint64(rng.get_state()["seed"]), ctypes.c_uint64(k),
                                                ctypes.c_void_p(self.state.data_ptr()), ctypes.c_uint64(first), L.stream_ptr()), "ardae_philox_normal_at")
        return out

    # ------------------------------------------------------------------------------------------------------------
    def cdae_phase(self, x, noise=None, apply_update=True):
        """ivae_ardae.py:713-779 (one cDAE update).  noise: optional dict(sampler [N,nd], sigma [B,nz,1], eps [N,z])."""
        self._cdae_grads(x, noise)
        if self.dp:
            self._allreduce(self.grads_c[:self.n_c])
        if apply_update:
            self._cdae_update()

    def _cdae_grads(self, x, noise=None, draw0=None):
        """ivae_ardae.py:713-776: sampler on N rows, latent statistics, perturbation, cDAE loss and its gradients (local shard).
        draw0: index of the first of this update's three Philox draws inside the step."""
        self._check_batch(x, "cdae_phase")
        cfg, lib, st = self.cfg, self.lib, L.stream_ptr()
        B, nz, z = self.B, cfg.nz_cdae, self.model.z_dim
        nstd = int(cfg.nstd_cdae)
        # inside a step the sigma- and eps-draws are made by the perturbation kernel itself where the shape allows (same Philox
        # keying as the separate draws: same numbers, two launches less)
        fused = (not noise) and self._in_step and draw0 is not None and self.fused_draws and bool(lib.ardae_latent_perturb_draw_ok(nz, nstd, z))
        if noise:
            ns, xi, eps = noise["sampler"], noise["sigma"].reshape(-1), noise["eps"]
        else:
            d = (None, None, None) if draw0 is None else (draw0, draw0 + 1, draw0 + 2)
            ns = self._normal(self.noise_s, d[0])
            xi, eps = (self.xi, self.eps) if fused else (self._normal(self.xi, d[1]), self._normal(self.eps, d[2]))
        if self.hidden_ctx:
            # aux models: hidden = model.encode.forward_hidden(x, std=0) and latent_mean = model.encode(x, std=0) are ONE std = 0 pass
            # (ivae_ardae.py:737-739,748), then the N-row pass
            raws = None
            if self.clipped:
                if noise:
                    raws = torch.stack([noise["ctx_raw"].reshape(B, -1), noise["z0_raw"].reshape(B, -1)]).float().contiguous()
                else:      # one Philox draw for both (index 9 + update: behind the 3 x 3 draws of up to three cDAE updates)
                    raws = self._normal(self.raw_c, None if draw0 is None else 9 + draw0 // 3)
            self._hidden(x, self.z0, self.ctx_c, self.ws_small, raws)
            self._encode(x, ns, nz, self.latent, self.ws)
        else:
            # context == latent_mean == encode(x, std=0) (lt0) and forward_hidden(x, nz) share the per-image trunk: one pass
            L.check(lib.ardae_model_encode_pair(ctypes.byref(self.model._desc), L.ptr(self.model._flat), L.ptr(self.pk_m), L.ptr(x), L.ptr(ns), B, nz,
                                                L.ptr(self.ws), self.ws.numel(), L.ptr(self.z0), L.ptr(self.latent), 0, st),
                    "ardae_model_encode_pair")
        if self.data_ctx:
            self._data_context(x, self.ctx_c)
        self._stamp("  sampler done")
        if fused and self.fused_first_layer and bool(lib.ardae_cdae_perturb_fused_ok(ctypes.byref(self.cdae._desc), nz, nstd)):
            # north star's "fused per-sample Gaussian-perturb + sigma-scaling + DAE-forward kernel", then the cDAE from its second layer on
            L.check(lib.ardae_cdae_perturb_loss_grads(ctypes.byref(self.cdae._desc), L.ptr(self.cdae._flat), L.ptr(self.pk_c), L.ptr(self.latent),
                                                      L.ptr(self.z0), L.ptr(self.ctx_c), B, nz, cfg.std_scale, cfg.delta,
                                                      ctypes.c_uint64(rng.get_state()["seed"]), ctypes.c_uint64(d[1]), ctypes.c_uint64(d[2]),
                                                      ctypes.c_void_p(self.state.data_ptr()), ctypes.c_uint64(self.rank * B * nz),
                                                      L.ptr(self.xbar), L.ptr(self.sigma), L.ptr(eps), L.ptr(self.std_b), L.ptr(self.ws),
                                                      self.ws.numel(), L.ptr(self.loss_c), L.ptr(self.grads_c), st),
                    "ardae_cdae_perturb_loss_grads")
            return
        if fused:
            L.check(lib.ardae_latent_perturb_draw(L.ptr(self.latent), L.ptr(self.z0), B, nz, z, cfg.std_scale, cfg.delta,
                                                  ctypes.c_uint64(rng.get_state()["seed"]), ctypes.c_uint64(d[1]), ctypes.c_uint64(d[2]),
                                                  ctypes.c_void_p(self.state.data_ptr()), ctypes.c_uint64(self.rank * B * nz),
                                                  L.ptr(self.xbar), L.ptr(self.sigma), L.ptr(eps), L.ptr(self.std_b), st),
                    "ardae_latent_perturb_draw")
        else:
            L.check(lib.ardae_latent_perturb_nstd(L.ptr(self.latent), L.ptr(self.z0), L.ptr(xi), L.ptr(eps), B, nz, nstd, z, cfg.std_scale, cfg.delta,
                                                  L.ptr(self.xbar), L.ptr(self.sigma), L.ptr(self.std_b), st), "ardae_latent_perturb_nstd")
        L.check(lib.ardae_cdae_loss_grads(ctypes.byref(self.cdae._desc), L.ptr(self.cdae._flat), L.ptr(self.pk_c), L.ptr(self.xbar),
                                          L.ptr(self.sigma), L.ptr(eps), L.ptr(self.ctx_c), B, nz * nstd, L.ptr(self.ws), self.ws.numel(),
                                          L.ptr(self.loss_c), L.ptr(self.grads_c), None, st), "ardae_cdae_loss_grads")

    def _cdae_update(self):
        """ivae_ardae.py:777-779: the cDAE optimiser's step on the (rank-averaged) gradients, then the weight re-pack."""
        if self._in_step and self.opt_c.adam:
            self.opt_c.advance(self.lib)           # the cDAE's own Adam block: t advances once per cDAE update
        self.opt_c.apply(self.lib, self.grads_c, self._in_step)
        if not self._in_step:
            self.opt_c.steps += 1
        self._pack_cdae()

    def _data_context(self, x, out):
        """--cdae-ctx-type data: the flattened image, centred to 2x - 1 for the MNIST family (ivae_ardae.py:730-734,809-813)."""
        D = int(self.model.input_dim)
        L.check(self.lib.ardae_center_scale(L.ptr(x), L.ptr(self._ctx_half), self.B, 1, D, 2.0 if self.cfg.ctx_data_center else 1.0, L.ptr(out),
                                            L.stream_ptr()), "ardae_center_scale")

    def vae_forward_part(self, x, noise=None, beta=None, draw=None):
        """ivae_ardae.py:781-827: everything of the VAE update that does not involve the cDAE (forward, ELBO pieces, z0, u)."""
        self._check_batch(x, "vae_forward_part")
        cfg, lib, st = self.cfg, self.lib, L.stream_ptr()
        beta = cfg.beta if beta is None else beta
        B, nz, md = self.B, cfg.nz_model, self.model._desc
        nv = noise["vae"] if noise else self._normal(self.noise_v, draw)
        L.check(lib.ardae_model_vae_forward(ctypes.byref(md), L.ptr(self.model._flat), L.ptr(self.pk_m), L.ptr(x), L.ptr(nv), B, nz,
                                            float(beta), L.ptr(self.ws_vae), self.ws_vae.numel(), L.ptr(self.zv), L.ptr(self.losses_m), st),
                "ardae_model_vae_forward")
        if self.hidden_ctx:      # context and latent mean of the VAE batch: one std = 0 pass (ivae_ardae.py:815-817,826)
            raws = None
            if self.clipped:
                if noise:
                    raws = torch.stack([noise["vctx_raw"].reshape(B, -1), noise["vz0_raw"].reshape(B, -1)]).float().contiguous()
                else:
                    raws = self._normal(self.raw_v, None if draw is None else 12)
            self._hidden(x, self.z0v, self.ctx_v, self.ws_small_v, raws)
        else:
            self._encode(x, None, 1, self.z0v, self.ws_small_v)
        if self.data_ctx:
            self._data_context(x, self.ctx_v)
        L.check(lib.ardae_center_scale(L.ptr(self.zv), L.ptr(self.z0v), B, nz, self.model.z_dim, cfg.std_scale, L.ptr(self.u), st))
        if self.split_backward:
            # model_loss.backward() through the decoder down to dL/dz (ivae_ardae.py:804) needs nothing from the cDAE either
            L.check(lib.ardae_model_vae_backward_decoder(ctypes.byref(md), L.ptr(self.model._flat), L.ptr(self.pk_m), L.ptr(x), L.ptr(nv), B, nz,
                                                         float(beta), 1.0, L.ptr(self.ws_vae), self.ws_vae.numel(), st),
                    "ardae_model_vae_backward_decoder")
        return nv

    def vae_backward_part(self, x, nv, beta=None, apply_update=True):
        """ivae_ardae.py:829-846: entropy gradient through the (updated) cDAE, backward, Adam."""
        self._vae_backward_grads(x, nv, beta)
        if self.dp:
            self._allreduce(self.grads_m)
        if apply_update:
            self._model_update()

    def _vae_backward_grads(self, x, nv, beta=None):
        self._check_batch(x, "vae_backward_part")
        cfg, lib, st = self.cfg, self.lib, L.stream_ptr()
        beta = cfg.beta if beta is None else beta
        B, nz, md = self.B, cfg.nz_model, self.model._desc
        L.check(lib.ardae_cdae_score(ctypes.byref(self.cdae._desc), L.ptr(self.cdae._flat), L.ptr(self.pk_c), L.ptr(self.u),
                                     L.ptr(self.sigma0), L.ptr(self.ctx_v), B, nz, L.ptr(self.ws_small), self.ws_small.numel(),
                                     L.ptr(self.g), st), "ardae_cdae_score")
        # seed of (s (z - z0)).backward(beta g / (B nz)) w.r.t. z  (ivae_ardae.py:834); B is the per-rank batch because the
        # ranks' gradients are averaged afterwards (mean over ranks of 1/B_local == 1/B_global sum)
        seed_scale = dist.entropy_seed_scale(cfg.std_scale, beta, B, nz)
        if self.split_backward:      # the decoder half already ran in vae_forward_part
            L.check(lib.ardae_model_vae_backward_sampler(ctypes.byref(md), L.ptr(self.model._flat), L.ptr(self.pk_m), L.ptr(x), L.ptr(nv), B, nz,
                                                         L.ptr(self.g), float(seed_scale), L.ptr(self.ws_vae), self.ws_vae.numel(),
                                                         L.ptr(self.grads_m), 0.0, st), "ardae_model_vae_backward_sampler")
        else:
            self.g.mul_(seed_scale)
            L.check(lib.ardae_model_vae_backward(ctypes.byref(md), L.ptr(self.model._flat), L.ptr(self.pk_m), L.ptr(x), L.ptr(nv), B, nz,
                                                 float(beta), 1.0, L.ptr(self.g), L.ptr(self.ws_vae), self.ws_vae.numel(), L.ptr(self.grads_m),
                                                 0.0, st), "ardae_model_vae_backward")

    def _model_update(self):
        self.opt_m.apply(self.lib, self.grads_m, self._in_step)   # in a step: t and the bias corrections come from the device step state
        if not self._in_step:
            self.step_count += 1
            self.opt_m.steps = self.step_count
            self.opt_m.advance(self.lib, self.RNG_STRIDE)   # phase calls made directly keep the device block (t of the coming step) in step
        self._pack_model()

    def vae_phase(self, x, noise=None, beta=None, apply_update=True):
        """ivae_ardae.py:781-846.  noise: optional dict(vae [B*nz_model, nd])."""
        nv = self.vae_forward_part(x, noise, beta)
        self.vae_backward_part(x, nv, beta, apply_update)

    def _step_body(self, xs, x_vae, noise, beta, capture=False):
        """One iteration: eager launches of the plan's units, or (capture=True) one linear graph per unit - returns the replay list."""
        self._in_step, self._draws = True, 0
        try:
            units = self._units(self._plan(xs, x_vae, noise, beta))
            graphs = self._run_units(units, capture=capture)
            return (units, graphs) if capture else None
        finally:
            self._in_step = False

    def _replay(self):
        units, graphs = self._graph
        self._run_units(units, graphs=graphs)

    # ---- diagnostics: an unprofiled timeline of the step (device clock stamps between the pieces of the plan) ----------------
    def enable_stamps(self, on=True):
        """Insert one-thread timestamp kernels (`ardae_debug_stamp`, the device's constant 100 MHz clock) at the boundaries of the
        plan's segments and between the calls of the cDAE phase; `read_stamps()` returns the last step's (name, microseconds)
        pairs.  Each stamp is a launch of its own (~5 us on its stream): a diagnostic, never on in timed runs."""
        self._stamps = {"buf": torch.zeros(256, dtype=torch.int64, device=self.dev), "names": []} if on else None
        self._graph = None

    def _stamp(self, name):
        st = self._stamps
        if st is None:
            return
        if name not in st["names"]:
            st["names"].append(name)
        L.check(self.lib.ardae_debug_stamp(ctypes.c_void_p(st["buf"].data_ptr()), st["names"].index(name), L.stream_ptr()), "ardae_debug_stamp")

    def read_stamps(self):
        st = self._stamps
        torch.cuda.synchronize()
        v = st["buf"].cpu().tolist()
        t0 = min(v[i] for i in range(len(st["names"])))
        return sorted(((n, (v[i] - t0) / 100.0) for i, n in enumerate(st["names"])), key=lambda kv: kv[1])

    def plan_summary(self):
        """What a replayed step submits, in launch order: "graph:<stream>" per linear graph, "allreduce" per collective."""
        if self._graph is None:
            return None
        return ["allreduce" if u[0] == "allreduce" else f"graph:{u[1]}" for u in self._graph[0]]

    def step(self, x_cdae, x_vae, noise=None, beta=None):
        """One iteration of the reference loop: num_cdae_updates cDAE updates (each on its own batch in the reference; the
        caller passes a list of batches when num_cdae_updates > 1) followed by one VAE update."""
        many = isinstance(x_cdae, (list, tuple))
        xs = list(x_cdae) if many else [x_cdae] * self.cfg.num_cdae_updates
        for x in xs:
            self._check_batch(x, "step(x_cdae)")
        self._check_batch(x_vae, "step(x_vae)")
        if 3 * len(xs) >= self.RNG_STRIDE:
            raise ValueError("at most %d cDAE updates per step (Philox offsets reserved per step)" % ((self.RNG_STRIDE - 1) // 3))
        if self.clipped and len(xs) > 3:
            raise ValueError("at most 3 cDAE updates per step with MNISTResConvAuxIPVAEClipped (its std = 0 draws use Philox offsets 9 .. 12)")
        if self.use_graph and noise is None:
            b = float(self.cfg.beta if beta is None else beta)
            # static copies of the batches (one per DISTINCT batch object: --num-cdae-updates k on one tensor shares its copy)
            if self._xc is None or len(self._xc) != len(xs):
                flat = lambda: torch.empty(self.B, self.model.input_dim, device=self.dev, dtype=torch.float32)
                self._xc, self._xv, self._graph = [flat() for _ in xs], flat(), None
            for buf, x in zip(self._xc, xs):      # batches were validated above: B x input_dim contiguous floats, whatever their view shape
                if x is not buf:
                    buf.copy_(x.view(self.B, -1))
            if x_vae is not self._xv:
                self._xv.copy_(x_vae.view(self.B, -1))
            key = (b, tuple(tuple(x.shape) for x in self._xc), tuple(self._xv.shape))
            # beta annealing (utils/msc.py:53-55) changes a frozen kernel argument every step: capture only once beta has stood
            # still for two steps, run eagerly while it moves (a capture per step would cost far more than replay saves)
            self._beta_stable = self._beta_stable + 1 if b == self._last_beta else 0
            self._last_beta = b
            if self._graph is not None and self._graph_key == key:
                self._replay()
            elif self._warmed and self._beta_stable < 2:
                self._step_body(self._xc, self._xv, None, b)
            elif not self._warmed:
                # first iteration of this engine eagerly: every kernel gets loaded outside of a capture
                self._step_body(self._xc, self._xv, None, b)
                self._warmed = True
            else:
                try:
                    g = self._step_body(self._xc, self._xv, None, b, capture=True)      # runs the step too (each unit is replayed once captured)
                except Exception as exc:
                    if self.graph_strict:
                        raise RuntimeError(f"ArdaeEngine(graph=True): HIP graph capture failed ({exc}); pass graph='auto' to fall back to "
                                           "eager launches, or graph=False") from exc
                    self.use_graph = False      # graph="auto": eager from now on
                    self._graph = None
                    import warnings
                    warnings.warn(f"ArdaeEngine: HIP graph capture failed ({exc}); continuing with eager launches")
                    self._step_body(self._xc, self._xv, None, b)
                else:
                    self._graph, self._graph_key = g, key
            self._count_step(len(xs))
            return
        self._step_body(xs, x_vae, noise, beta)
        self._count_step(len(xs))

    def input_buffers(self, n_cdae=None):
        """The engine's static batch buffers ([B, input_dim] each: one per cDAE update, and the VAE batch's): a producer that
        already works on the device (dynamic binarisation, a gather from a resident table) writes the next batches THERE and
        passes the same tensors to step(), which then has nothing to copy."""
        n = self.cfg.num_cdae_updates if n_cdae is None else int(n_cdae)
        if self._xc is None or len(self._xc) != n:
            flat = lambda: torch.empty(self.B, self.model.input_dim, device=self.dev, dtype=torch.float32)
            self._xc, self._xv, self._graph = [flat() for _ in range(n)], flat(), None
        return list(self._xc), self._xv

    def _count_step(self, n_cdae_updates):
        self.step_count += 1
        self.opt_m.steps = self.step_count
        self.opt_c.steps += n_cdae_updates

    # ------------------------------------------------------------------------------------------------------------
    # Checkpoints in the reference's format (ivae_ardae.py:931-950,1120-1139; utils/msc.py:67-93): one dict per network with
    # 'state_dict' and 'optimizer' (torch.optim.Optimizer.state_dict() layout: per-parameter 'step' / 'exp_avg' / 'exp_avg_sq',
    # resp. 'step' / 'square_avg' / 'momentum_buffer', and 'param_groups'), so that files written by the reference loop, by the
    # drop-in modules + net.Adam / net.RMSprop, and by the fused engine are interchangeable.  The caller adds its own
    # bookkeeping keys ('epoch', 'batch_idx', 'best_val_loss', ...) exactly as the reference does.
    def _per_param(self, module, flat, n_used):
        out, off = [], 0
        for name, p in module.named_parameters():
            k = p.numel()
            out.append(flat[off:off + k].view_as(p) if off + k <= n_used else None)
            off += k
        return out

    def _opt_state(self, module, opt):
        """torch.optim state_dict()['state'] of one network: {param index: {'step', <buffers of the optimiser>}}; parameters without
        gradients (the cDAE's trailing bias) and optimisers without state (SGD) contribute nothing."""
        if not opt.steps or opt.kind == "sgd":
            return {}
        views = [self._per_param(module, t, opt.n) for t in opt.buffers()]
        names = opt.state_names()
        out = {}
        for i in range(len(views[0])):
            if views[0][i] is not None:
                out[i] = dict({"step": opt.steps}, **{nm: v[i].clone() for nm, v in zip(names, views)})
        return out

    def model_checkpoint(self):
        nparams = len(list(self.model.named_parameters()))
        return {"state_dict": {k: t.clone() for k, t in self.model.state_dict().items()},
                "optimizer": {"state": self._opt_state(self.model, self.opt_m), "param_groups": [self.opt_m.param_group(nparams)]},
                # state_version 2 (round 3 on): the device step block describes the COMING step (the last launch of a step advances it)
                "engine": {"state_version": 2, "step_count": self.step_count, "cdae_steps": self.opt_c.steps, "rng_seed": rng.get_state()["seed"],
                           "rng_host_offset": rng.get_state()["offset"], "step_state": self.state.cpu().clone()}}

    def cdae_checkpoint(self):
        nparams = len(list(self.cdae.named_parameters()))
        return {"state_dict": {k: t.clone() for k, t in self.cdae.state_dict().items()},
                "optimizer": {"state": self._opt_state(self.cdae, self.opt_c), "param_groups": [self.opt_c.param_group(nparams)]}}

    @staticmethod
    def _kind_of_group(group):
        """Which optimiser wrote this torch.optim param_group (utils.Adam / torch.optim.RMSprop / torch.optim.SGD layouts)."""
        if "betas" in group:
            return "amsgrad" if group.get("amsgrad") else "adam"
        return "rmsprop" if "alpha" in group else "sgd"

    def _load_opt_state(self, module, opt, state, what, groups=None):
        if groups:
            wrote = self._kind_of_group(groups[0])
            if wrote != opt.kind:
                raise ValueError(f"{what}: written by optimiser {wrote!r}, but this engine was built with {opt.kind!r} for that network")
        for t in opt.buffers():
            t.zero_()
        steps = {int(st["step"]) for st in state.values()}
        if len(steps) > 1:
            raise ValueError(f"the fused engine keeps one step count per network ({what}: {sorted(steps)})")
        names = opt.state_names()
        if state and not all(nm in next(iter(state.values())) for nm in names[:1]):
            raise ValueError(f"{what}: the checkpoint's optimiser state does not belong to {opt.kind!r} (engine built with TrainConfig."
                             f"{'m' if opt is self.opt_m else 'd'}_optimizer={opt.kind!r})")
        with torch.no_grad():
            for nm, buf in zip(names, opt.buffers()):
                for i, t in enumerate(self._per_param(module, buf, opt.n)):
                    if t is not None and i in state and state[i].get(nm) is not None:
                        t.copy_(state[i][nm])
        return steps.pop() if steps else 0

    def load_checkpoints(self, model_ckpt, cdae_ckpt):
        """Inverse of model_checkpoint() / cdae_checkpoint(); also accepts files written by the reference loop."""
        self.model.load_state_dict(model_ckpt["state_dict"])
        self.cdae.load_state_dict(cdae_ckpt["state_dict"])
        eng = model_ckpt.get("engine")
        m_steps = self._load_opt_state(self.model, self.opt_m, model_ckpt["optimizer"]["state"], "model checkpoint",
                                       model_ckpt["optimizer"].get("param_groups"))
        c_steps = self._load_opt_state(self.cdae, self.opt_c, cdae_ckpt["optimizer"]["state"], "cdae checkpoint",
                                       cdae_ckpt["optimizer"].get("param_groups"))
        # optimisers without per-parameter state (SGD) carry no step count: the engine's own record, if the file has one
        self.step_count = m_steps if (m_steps or eng is None) else int(eng["step_count"])
        self.opt_m.steps = self.step_count
        self.opt_c.steps = c_steps if (c_steps or eng is None) else int(eng.get("cdae_steps", 0))
        if eng is not None:     # written by this engine: continue the same noise stream (the saved block already describes the coming step)
            rng.manual_seed(eng["rng_seed"], eng.get("rng_host_offset", rng.get_state()["offset"]))
            self.state.copy_(eng["step_state"].to(self.dev))
            if int(eng.get("state_version", 1)) < 2:
                # written before round 3: the block still described the step just DONE (t == step_count, that step's Philox offsets) -
                # advance it once, or Adam's t would lag by one for good and the first resumed step would repeat the last step's noise
                self.opt_m.advance(self.lib, self.RNG_STRIDE)
        else:                   # written by the reference / the module path: the optimisers' t, and Philox offsets this run has not used yet
            self.state.zero_()      # (a resumed run with an unchanged seed would otherwise replay the draws of steps 1..step_count)
            self.state[0] = self.RNG_STRIDE * self.step_count
            self.state[1] = self.step_count
            self.opt_m.advance(self.lib, self.RNG_STRIDE)
        self.opt_c.state.zero_()
        self.opt_c.state[1] = self.opt_c.steps
        self._graph = None      # parameters were rewritten outside of the captured step
        if self._log is not None:
            self._log.resync()   # the log's iteration numbers come from the device t: first unreported iteration = step_count + 1
        self.repack()

    def stats(self):
        """Host copy of the logged scalars of ivae_ardae.py:756-758,774,837-841 (this is the only synchronising call)."""
        v = torch.cat([self.loss_c, self.losses_m, self.std_b.mean().reshape(1), self.std_b.max().reshape(1), self.std_b.min().reshape(1)]).tolist()
        return dict(cdae_loss=v[0], model_loss=v[1], recon=v[2], prior=v[3], std_mean=v[4], std_max=v[5], std_min=v[6])
