"""K6w (batched weight gradients, C ABI `ardae_wgrad_batch`) against a float64 restatement:
dW[o][i] = sum_pairs sum_m G[m][o] X[m][i], bias[o] = sum_m G[bias_pair][m][o], rowscale[o] = sum_m sigma[m] G[bias_pair][m][o].

One batch mixes the three kernels behind the entry point: the software-pipelined 256x256 and 256x32 geometries
(wgrad_wide.hip: M % 32 == 0, O % 256 == 0) and the generic ragged kernel (wgrad.hip).  Tolerance: fp32 accumulation
over M rows in a fixed order; |err| <= 1e-5 of the output scale at M = 16384.
"""
import ctypes

import pytest
import torch

import ardae_amd
from ardae_amd import _lib as L

pytestmark = pytest.mark.gpu


def run_batch(specs, seed):
    """specs: list of (M, O, I, npairs, want_bias, want_rowscale)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    lib = L.lib()
    probs = (L.WgradProblem * len(specs))()
    keep, refs = [], []
    for k, (M, O, I, npairs, wb, wr) in enumerate(specs):
        p = probs[k]
        p.M, p.O, p.I, p.npairs = M, O, I, npairs
        Gs = [torch.randn(M, O, device="cuda", generator=g) for _ in range(npairs)]
        Xs = [torch.randn(M, I, device="cuda", generator=g) for _ in range(npairs)]
        for q in range(npairs):
            p.G[q], p.ldG[q], p.X[q], p.ldX[q] = Gs[q].data_ptr(), O, Xs[q].data_ptr(), I
        sig = torch.randn(M, device="cuda", generator=g)
        p.bias_pair = npairs - 1 if (wb or wr) else -1
        p.rowscale = sig.data_ptr() if wr else None
        p.splits = lib.ardae_wgrad_splits(M, O, I, len(specs))
        part = torch.empty(p.splits * O * I, device="cuda"); pvec = torch.empty(p.splits * 2 * O, device="cuda")
        out = torch.full((O, I + 3), float("nan"), device="cuda")        # strided output view, like a block of a wider matrix
        ob = torch.full((O,), float("nan"), device="cuda"); ors = torch.full((O, 5), float("nan"), device="cuda")
        p.partial, p.partial_vec = part.data_ptr(), pvec.data_ptr()
        p.out, p.ldout = out.data_ptr(), I + 3
        p.out_bias = ob.data_ptr() if wb else None
        p.out_rowscale, p.ld_rowscale = (ors[:, 2:].data_ptr(), 5) if wr else (None, 0)
        p.beta = 0.0
        keep += Gs + Xs + [sig, part, pvec, out, ob, ors]
        dW = sum(Gs[q].double().T @ Xs[q].double() for q in range(npairs))
        Gb = Gs[p.bias_pair].double() if p.bias_pair >= 0 else None
        refs.append((out, ob, ors, dW, Gb.sum(0) if wb else None, (sig.double()[:, None] * Gb).sum(0) if wr else None, I))
    L.check(lib.ardae_wgrad_batch(probs, len(specs), L.stream_ptr()), "ardae_wgrad_batch")
    torch.cuda.synchronize()
    for out, ob, ors, dW, bsum, rsum, I in refs:
        assert float((out[:, :I].double() - dW).abs().max() / dW.abs().max()) < 1e-5
        assert torch.isnan(out[:, I:]).all()                              # nothing written outside the [O, I] view
        if bsum is not None:
            assert float((ob.double() - bsum).abs().max() / bsum.abs().max()) < 1e-5
        if rsum is not None:
            assert float((ors[:, 2].double() - rsum).abs().max() / rsum.abs().max()) < 1e-5


def test_wgrad_batch_mixed_kernels():
    # the cDAE update's shapes at 16384 rows: two-pair 256x256 problems (one with bias + sigma column), a first-layer
    # 256x32 problem, and per-image / ragged problems for the generic kernel
    run_batch([(16384, 256, 256, 2, True, True), (16384, 256, 256, 2, True, False), (16384, 256, 32, 2, True, False),
               (64, 256, 256, 1, False, False), (16384, 32, 256, 1, True, False), (1000, 96, 130, 2, True, True)], seed=1)


def test_wgrad_wide_uneven_splits():
    # 3 tiles -> 85 splits of 512+512 chunks: the last split of each pair range is short, some slices straddle the pair boundary
    run_batch([(16384, 256, 256, 2, True, True), (16384, 256, 256, 1, False, False), (16384, 256, 256, 2, True, False)], seed=2)
    run_batch([(2048, 512, 256, 1, True, False)], seed=3)


def test_wgrad_full_size():
    # BASELINE config #2: 131072 rows (reference in float64 on the GPU)
    run_batch([(131072, 256, 256, 2, True, True), (131072, 256, 32, 2, True, False)], seed=4)


def test_wgrad_many_tiles_several_launches():
    # h_dim 1024 (BASELINE config #5): 16 tiles of 256 x 256 per matrix, 32 per software-pipelined launch - three such
    # problems go out as two launches, each with the row splits that fill the chip (one of them with bias + sigma column);
    # plus h_dim 512 shapes (4 tiles) and a toy-model first layer (I = 2: the ragged N-row problem on the generic kernel)
    run_batch([(4096, 1024, 1024, 2, True, True), (4096, 1024, 1024, 1, True, False), (4096, 1024, 1024, 2, False, False),
               (4096, 512, 512, 2, True, False), (4096, 1024, 32, 2, True, False)], seed=5)
    run_batch([(32768, 256, 2, 2, True, False), (32768, 512, 512, 1, True, False)], seed=6)
