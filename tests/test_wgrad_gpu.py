"""K6w (batched weight gradients, C ABI `ardae_wgrad_batch`) against a float64 restatement:
dW[o][i] = sum_pairs sum_m G[m][o] X[m][i], bias[o] = sum_m G[bias_pair][m][o], rowscale[o] = sum_m sigma[m] G[bias_pair][m][o].

One batch mixes the kernels behind the entry point: the 256x256-tile problems (wgrad_x9.hip since round 4: fp32 products formed exactly on
the BF16 matrix cores), the software-pipelined 256x32 geometry (wgrad_wide.hip: M % 32 == 0, O % 256 == 0) and the generic ragged kernel
(wgrad.hip).  Tolerance: fp32 accumulation
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


# ---- wgrad_x9_kernel: fp32 products formed EXACTLY from three bf16 pieces per operand on the BF16 matrix cores (round 4)
def _x9_problem(G, X, sig=None):
    """One 256 x 256-tile problem through ardae_wgrad_batch; returns (dW, bias sums, sigma-weighted sums) as float64 CPU tensors."""
    lib = L.lib()
    M, O = G.shape; I = X.shape[1]
    probs = (L.WgradProblem * 1)()
    p = probs[0]
    p.M, p.O, p.I, p.npairs = M, O, I, 1
    p.G[0], p.ldG[0], p.X[0], p.ldX[0] = G.data_ptr(), O, X.data_ptr(), I
    p.bias_pair = 0
    p.rowscale = sig.data_ptr() if sig is not None else None
    p.splits = lib.ardae_wgrad_splits(M, O, I, 1)
    part = torch.empty(p.splits * O * I, device="cuda"); pvec = torch.empty(p.splits * 2 * O, device="cuda")
    out = torch.full((O, I), float("nan"), device="cuda"); ob = torch.full((O,), float("nan"), device="cuda"); ors = torch.full((O,), float("nan"), device="cuda")
    p.partial, p.partial_vec, p.out, p.ldout, p.out_bias = part.data_ptr(), pvec.data_ptr(), out.data_ptr(), I, ob.data_ptr()
    p.out_rowscale, p.ld_rowscale = (ors.data_ptr(), 1) if sig is not None else (None, 0)
    p.beta = 0.0
    lib.ardae_profile_enable(1)
    L.check(lib.ardae_wgrad_batch(probs, 1, L.stream_ptr()), "ardae_wgrad_batch")
    names = [e["name"] for e in L.profile_report()]
    lib.ardae_profile_enable(0)
    assert any(n.startswith("wgrad_x9_kernel") for n in names), names          # the kernel under test is the one that ran
    return out.double().cpu(), ob.double().cpu(), ors.double().cpu()


def test_wgrad_x9_exact_products_on_adversarial_values():
    """The claim behind wgrad_x9.hip: every fp32 operand is the EXACT sum of three bf16 pieces and the nine piece products are exact in
    fp32, so the kernel's only rounding is the fp32 accumulation - the error bound of an fp32 dot product, |err| <= c M eps sum |g| |x|,
    holds for ANY operand values, not only for well-scaled ones.  Inputs that would expose a piece being rounded or dropped: values with all 24
    significand bits set, magnitudes from 1e-18 to 1e18 in one column (per-row scales that cancel between G and X, so that every product is
    O(1) while the operands are not), exact zeros, negative values, and rows where G x X cancels to a small sum."""
    torch.manual_seed(5)
    M, O, I = 4096, 256, 256
    g = torch.Generator(device="cuda").manual_seed(17)
    G = torch.randn(M, O, device="cuda", generator=g); X = torch.randn(M, I, device="cuda", generator=g)
    # full significands: odd multiples of 2^-23 next to a power of two
    G[::7] = (1.0 + (2.0 ** -23) * torch.randint(1, 1 << 23, (G[::7].shape), device="cuda", generator=g).float()) * torch.sign(G[::7])
    X[::5] = (1.0 + (2.0 ** -23) * torch.randint(1, 1 << 23, (X[::5].shape), device="cuda", generator=g).float()) * torch.sign(X[::5])
    scale = 10.0 ** torch.randint(-18, 19, (M, 1), device="cuda", generator=g).float()
    G = (G * scale).contiguous(); X = (X / scale).contiguous()                     # products O(1), operands 1e-18 .. 1e18
    G[100:164] = 0.0; X[300:364, ::2] = 0.0
    sig = torch.randn(M, device="cuda", generator=g)
    dW, bs, rs = _x9_problem(G, X, sig)
    Gd, Xd = G.double().cpu(), X.double().cpu()
    ref = Gd.T @ Xd
    bound = 4.0 * M * 2.0 ** -24 * (Gd.abs().T @ Xd.abs())                       # fp32 accumulation in a fixed, blocked order: far inside
    assert bool(((dW - ref).abs() <= bound * 0.05 + 1e-30).all()), float(((dW - ref).abs() / (bound + 1e-30)).max())
    # and in relative terms against what an fp32 FMA chain achieves on well-scaled data (6e-7 of the result scale): the x9 kernel is not worse
    assert float((dW - ref).abs().max() / ref.abs().max()) < 1e-6
    assert float((bs - Gd.sum(0)).abs().max() / Gd.abs().sum(0).max()) < 1e-6
    assert float((rs - (sig.double().cpu()[:, None] * Gd).sum(0)).abs().max() / (sig.double().cpu().abs()[:, None] * Gd.abs()).sum(0).max()) < 1e-6


def test_wgrad_x9_not_less_accurate_than_an_fp32_fma_chain():
    """Same data through the x9 kernel and through torch's fp32 matmul (an fp32 FMA / MFMA chain per output element), both against float64:
    the x9 result's rms error must not exceed the fp32 product's by more than 20 % (measured: it is lower - the piece products are exact, only
    the accumulation rounds)."""
    g = torch.Generator(device="cuda").manual_seed(3)
    M = 16384
    G = torch.randn(M, 256, device="cuda", generator=g); X = torch.randn(M, 256, device="cuda", generator=g) * 3
    dW, _, _ = _x9_problem(G, X)
    ref = G.double().T @ X.double()
    f32 = (G.T @ X).double().cpu()
    rms = lambda a: float((a - ref.cpu()).pow(2).mean().sqrt())
    assert rms(dW) <= 1.2 * rms(f32), (rms(dW), rms(f32))
