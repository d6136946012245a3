"""The engine's own counter-based RNG stream (Philox4x32-10 in csrc/elementwise.hip).

Replaces the `torch.randn` calls of the reference hot path (models/ivae/mnist.py:73, ivae_ardae.py:761,
models/graddae/mlp.py:22).  GPU Philox cannot reproduce torch-CPU draws, so parity tests inject noise tensors instead;
this stream is what training and the benchmark use.  Every draw consumes one `offset`; element i of a draw depends
only on (seed, offset, i), so results do not depend on how rows are partitioned over ranks.
"""
import ctypes

import torch

from . import _lib as L

_state = {"seed": 0x5EED, "offset": 0}


def manual_seed(seed, offset=0):
    _state["seed"], _state["offset"] = int(seed) & (2 ** 64 - 1), int(offset)


def get_state():
    return dict(_state)


def _next_offset():
    o = _state["offset"]
    _state["offset"] = o + 1
    return o


def normal(shape, device, out=None, first_element=0):
    """first_element: this tensor is the slice [first_element, first_element + numel) of a larger draw (data parallelism:
    rank r of R equal shards passes r * numel and gets the numbers a single process would have generated there)."""
    t = out if out is not None else torch.empty(shape, device=device, dtype=torch.float32)
    L.check(L.lib().ardae_philox_normal_at(L.ptr(t), t.numel(), ctypes.c_uint64(_state["seed"]), ctypes.c_uint64(_next_offset()), None,
                                           ctypes.c_uint64(first_element), L.stream_ptr()), "ardae_philox_normal_at")
    return t


def uniform(shape, device, out=None):
    t = out if out is not None else torch.empty(shape, device=device, dtype=torch.float32)
    L.check(L.lib().ardae_philox_uniform(L.ptr(t), t.numel(), ctypes.c_uint64(_state["seed"]), ctypes.c_uint64(_next_offset()),
                                         L.stream_ptr()), "ardae_philox_uniform")
    return t
