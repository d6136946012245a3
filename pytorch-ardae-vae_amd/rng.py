"""The engine's own counter-based RNG stream (Philox4x32-10 in csrc/elementwise.hip).

Replaces the `torch.randn` calls of the reference hot path (models/ivae/mnist.py:73, ivae_ardae.py:761,
models/graddae/mlp.py:22).  GPU Philox cannot reproduce torch-CPU draws, so parity tests inject noise tensors instead;
this stream is what training and the benchmark use.  Every draw consumes one `offset`; element i of a draw depends
only on (seed, offset, i), so results do not depend on how rows are partitioned over ranks.

Two consumers share one seed and must never share a Philox word: the fused engine's in-step draws use offsets
`RNG_STRIDE * step + k` (device-resident step state, engine.py), host-side draws (this module: `normal`, `uniform`,
`data.dynamic_binarize`, `model.generate/logprob`, the drop-in modules' noise) use `HOST_STREAM | n` with the top bit of
the 64-bit offset set.  The host counter `n` is part of the engine checkpoint (engine.py::model_checkpoint).
"""
import ctypes

import torch

from . import _lib as L

HOST_STREAM = 1 << 63          # offset space of host-side draws; the engine's in-step offsets stay below it

_state = {"seed": 0x5EED, "offset": 0}


def manual_seed(seed, offset=0):
    _state["seed"], _state["offset"] = int(seed) & (2 ** 64 - 1), int(offset)


def get_state():
    return dict(_state)


def _next_offset():
    """Philox offset of the next host-side draw (disjoint from every in-step offset of the engine)."""
    o = _state["offset"]
    if o >= HOST_STREAM:
        raise OverflowError("host Philox stream exhausted")
    _state["offset"] = o + 1
    return HOST_STREAM | o


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
