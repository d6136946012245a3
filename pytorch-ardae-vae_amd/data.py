"""On-device synthetic / binarised batches for the train loop: no host-to-device copy per step.

    dynamic_binarize  <- datasets/mnist.py:36-40 (`torch.bernoulli(probs)` per batch on the host in the reference)
    StaticBinarizedSource <- datasets/sbmnist.py:34-60,97-130 (Larochelle's statically binarised MNIST: a FIXED table of 0/1 rows,
                         shuffled every epoch by the DataLoader; BASELINE config #3)
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


class StaticBinarizedSource:
    """Statically binarised images: the rows are drawn (or read) ONCE and never change (datasets/sbmnist.py:46-60); batches are
    gathered from the device-resident table by a fresh permutation per epoch (the reference's `DataLoader(shuffle=True)`,
    datasets/sbmnist.py:97-130), with no host copy per step.  The fused engine works on a fixed batch size, so an epoch ends
    when fewer than `batch_size` unseen rows remain (drop_last semantics; the reference's loop restarts its iterator at that
    point too, ivae_ardae.py:718-722)."""

    def __init__(self, table, seed=0):
        if not (table.is_cuda and table.dtype == torch.float32 and table.dim() == 2 and table.is_contiguous()):
            raise RuntimeError("StaticBinarizedSource: expected a contiguous float32 CUDA table [rows, input_dim]")
        self.table = table
        self._gen = torch.Generator(device=table.device).manual_seed(int(seed))
        self._perm = torch.randperm(table.size(0), device=table.device, generator=self._gen)
        self._pos, self.epoch, self.last_indices = 0, 0, None

    @classmethod
    def synthetic(cls, num_rows=50000, input_dim=784, device="cuda", seed=0):
        """A synthetic stand-in of the 50 000 x 784 training table (there is no dataset on the box): a fixed "mean image" with
        MNIST-like sparsity, binarised ONCE from the library's Philox stream."""
        g = torch.Generator().manual_seed(int(seed))
        p = ((torch.rand(input_dim, generator=g) < 0.2).float() * 0.6 + 0.03).to(device)
        table = torch.empty(num_rows, input_dim, device=device)
        st = rng.get_state()
        L.check(L.lib().ardae_bernoulli(L.ptr(p), num_rows, input_dim, L.ptr(table), ctypes.c_uint64(st["seed"]), ctypes.c_uint64(rng._next_offset()),
                                        L.stream_ptr()), "ardae_bernoulli")
        return cls(table, seed)

    @classmethod
    def from_amat(cls, path, device="cuda", seed=0):
        """Rows of a `binarized_mnist_{train,valid,test}.amat` text file (the files datasets/sbmnist.py:34-42 downloads and :46-49 parses)."""
        import numpy as np
        rows = np.loadtxt(path).astype("float32")
        return cls(torch.from_numpy(rows.reshape(rows.shape[0], -1)).to(device).contiguous(), seed)

    def next_batch(self, batch_size, out=None):
        n = self.table.size(0)
        if batch_size > n:
            raise ValueError(f"batch_size {batch_size} > {n} rows")
        if self._pos + batch_size > n:                       # epoch finished: reshuffle
            self._perm = torch.randperm(n, device=self.table.device, generator=self._gen)
            self._pos, self.epoch = 0, self.epoch + 1
        idx = self._perm[self._pos:self._pos + batch_size]
        self._pos += batch_size
        self.last_indices = idx
        out = torch.empty(batch_size, self.table.size(1), device=self.table.device) if out is None else out
        L.check(L.lib().ardae_gather_rows(L.ptr(self.table), ctypes.c_void_p(idx.data_ptr()), batch_size, self.table.size(1), L.ptr(out),
                                          L.stream_ptr()), "ardae_gather_rows")
        return out
