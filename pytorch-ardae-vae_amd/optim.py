"""Optimisers of the reference loop as fused multi-tensor HIP updates.

    Adam    <- utils/optim.py:9-108 (vendored old-style Adam: `denom = (sqrt(v) + eps) / sqrt(1 - beta2^t)`, :102-106)
    RMSprop <- torch.optim.RMSprop as constructed at ivae_ardae.py:625-626 (momentum form, alpha=.99, eps=1e-8)

Same constructor signatures, `param_groups`, `state_dict()` layout (`exp_avg`, `exp_avg_sq`, `step`, `max_exp_avg_sq`;
`square_avg`, `momentum_buffer`, `step`).  Parameters that are adjacent views of one flat buffer (the modules in
modules.py) and whose gradients are adjacent too are updated by ONE kernel launch per run; a parameter whose `.grad`
is None is skipped and keeps no state, exactly like the reference (`neglogprob.fc.bias`).
"""
import torch
from torch.optim.optimizer import Optimizer

from . import _lib as L


def _runs(params):
    """Group (param, grad, state tensors...) tuples into maximal runs that are contiguous in memory."""
    runs, cur = [], None
    for tensors in params:
        if cur is not None and all(c[-1].data_ptr() + c[-1].numel() * 4 == t.data_ptr() for c, t in zip(cur, tensors)):
            for c, t in zip(cur, tensors):
                c.append(t)
        else:
            if cur is not None:
                runs.append(cur)
            cur = [[t] for t in tensors]
    if cur is not None:
        runs.append(cur)
    return runs


def _bump(p):
    torch.autograd.graph.increment_version(p)


class _FlatStateOptimizer(Optimizer):
    _state_names = ()

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        # these optimisers bump the parameters' version counters after every step, so the owning modules may keep their packed
        # weight image until a version moves instead of rebuilding it at every forward (modules.py::FlatParamModule._pack_is_current)
        for group in self.param_groups:
            for p in group["params"]:
                owner = getattr(p, "_ardae_owner", None)
                owner = owner() if owner is not None else None
                if owner is not None:
                    owner._tracked = True

    def _state_for(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            for n in self._state_names(p):
                st[n] = None
        return st

    def _alloc_states(self, group_params, names):
        """Allocate state tensors of consecutive parameters as views of one buffer so the update stays one launch."""
        need = [p for p in group_params if any(self.state[p].get(n) is None for n in names)]
        if not need:
            return
        total = sum(p.numel() for p in need)
        for n in names:
            buf = torch.zeros(total, device=need[0].device, dtype=torch.float32)
            off = 0
            for p in need:
                self.state[p][n] = buf[off:off + p.numel()].view_as(p)
                off += p.numel()


class Adam(_FlatStateOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        if weight_decay != 0:
            raise NotImplementedError("weight_decay != 0 is never used by ivae_ardae.py and is not implemented on the HIP path")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))

    def _state_names(self, p):
        return ("exp_avg", "exp_avg_sq")

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = L.lib()
        for group in self.param_groups:
            active = [p for p in group["params"] if p.grad is not None]
            if not active:
                continue
            names = ["exp_avg", "exp_avg_sq"] + (["max_exp_avg_sq"] if group["amsgrad"] else [])
            for p in active:
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                self._state_for(p)
                for n in names:
                    self.state[p].setdefault(n, None)
            self._alloc_states(active, names)
            # parameters that share a step count and sit next to each other go out in one launch
            tuples = []
            for p in active:
                st = self.state[p]
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                tuples.append((p.data, g, st["exp_avg"], st["exp_avg_sq"]) + ((st["max_exp_avg_sq"],) if group["amsgrad"] else ()) )
            steps = [self.state[p]["step"] for p in active]
            start = 0
            while start < len(active):
                end = start
                while end < len(active) and steps[end] == steps[start]:
                    end += 1
                for run in _runs(tuples[start:end]):
                    n = sum(t.numel() for t in run[0])
                    vmax = L.ptr(run[4][0]) if group["amsgrad"] else None
                    L.check(lib.ardae_adam_ref_step(L.ptr(run[0][0]), L.ptr(run[1][0]), L.ptr(run[2][0]), L.ptr(run[3][0]), vmax, n,
                                                    float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]),
                                                    float(group["eps"]), int(steps[start]), L.stream_ptr()), "ardae_adam_ref_step")
                start = end
            for p in active:
                _bump(p)
        return loss


class RMSprop(_FlatStateOptimizer):
    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0, momentum=0, centered=False):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= momentum:
            raise ValueError("Invalid momentum value: {}".format(momentum))
        if not 0.0 <= alpha:
            raise ValueError("Invalid alpha value: {}".format(alpha))
        if weight_decay != 0 or centered:
            raise NotImplementedError("weight_decay / centered RMSprop are never used by ivae_ardae.py and are not implemented on the HIP path")
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay, momentum=momentum, centered=centered))

    def _state_names(self, p):
        return ("square_avg", "momentum_buffer")

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = L.lib()
        for group in self.param_groups:
            active = [p for p in group["params"] if p.grad is not None]
            if not active:
                continue
            names = ["square_avg"] + (["momentum_buffer"] if group["momentum"] > 0 else [])
            for p in active:
                self._state_for(p)
                for n in names:
                    self.state[p].setdefault(n, None)
            self._alloc_states(active, names)
            tuples = []
            for p in active:
                st = self.state[p]
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                tuples.append((p.data, g, st["square_avg"]) + ((st["momentum_buffer"],) if group["momentum"] > 0 else ()))
            for run in _runs(tuples):
                n = sum(t.numel() for t in run[0])
                buf = L.ptr(run[3][0]) if group["momentum"] > 0 else None
                L.check(lib.ardae_rmsprop_step(L.ptr(run[0][0]), L.ptr(run[1][0]), L.ptr(run[2][0]), buf, n, float(group["lr"]),
                                               float(group["alpha"]), float(group["eps"]), float(group["momentum"]), L.stream_ptr()),
                        "ardae_rmsprop_step")
            for p in active:
                _bump(p)
        return loss
