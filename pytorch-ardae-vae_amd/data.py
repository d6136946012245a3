"""On-device synthetic / binarised batches for the train loop: no host-to-device copy per step.

    dynamic_binarize  <- datasets/mnist.py:36-40 (`torch.bernoulli(probs)` per batch on the host in the reference)
    gaussians25       <- datasets/toy.py:193-227 (`exp4`: 5 x 5 grid of isotropic Gaussians on [-4, 4]^2, variance 0.1,
                         the same number of points per mixture component, labels 0..24 in grid order)
Both draw from the library's Philox stream (`rng.manual_seed`), like the engine's noise.
"""
import ctypes
import math

import torch

from . import _lib as L
from . import rng


def dynamic_binarize(probs, out=None):
    """x ~ Bernoulli(probs) element-wise on the GPU.  probs: [B, D] float32 CUDA tensor (pixel intensities in [0, 1])."""
    p2 = probs.reshape(probs.size(0), -1)
    if not (p2.is_cuda and p2.dtype == torch.float32 and p2.is_contiguous()):
        raise RuntimeError("dynamic_binarize: expected a contiguous float32 CUDA tensor")
    out = torch.empty_like(p2) if out is None else out
    # ardae_bernoulli takes one probability per column; a full [B, D] table is B*D columns of a single row
    st = rng.get_state()
    L.check(L.lib().ardae_bernoulli(L.ptr(p2), 1, p2.numel(), L.ptr(out), ctypes.c_uint64(st["seed"]), ctypes.c_uint64(rng._next_offset()),
                                    L.stream_ptr()), "ardae_bernoulli")
    return out.view_as(probs)


def gaussians25(num_data, device="cuda", var=0.1, extent=4.0, n=5):
    """(x [num_data, 2], label [num_data]) of the reference's 25-Gaussians toy set, generated on `device`."""
    N = n * n
    if num_data % N != 0:
        raise ValueError("num_data should be multiple of {} (num_data = {})".format(N, num_data))
    per = num_data // N
    lin = torch.linspace(-extent, extent, n, device=device)
    yv, xv = torch.meshgrid(lin, lin, indexing="ij")                 # np.meshgrid(x, y) order: x varies fastest
    mu = torch.stack([xv.reshape(-1), yv.reshape(-1)], 1)             # [25, 2]
    eps = rng.normal((num_data, 2), device)
    label = torch.arange(N, device=device).repeat_interleave(per)
    x = mu[label] + math.sqrt(var) * eps
    return x, label
