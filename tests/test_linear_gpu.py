"""K1 (fused linear on FP32 MFMA) against a float64 CPU restatement of the same operator.

Tolerance: the kernel is an fp32 fmaf chain; |err| <= 2e-6 * sum|a*b| per output (guide: ~1e-7 at K<=1024),
checked as max-abs error relative to the output scale, 2e-5.
"""
import ctypes

import pytest
import torch

import ardae_amd
from ardae_amd import _lib as L

pytestmark = pytest.mark.gpu


def pack(W, transpose=False):
    nout, k = (W.shape[1], W.shape[0]) if transpose else W.shape
    out = torch.empty(L.lib().ardae_packed_floats(nout, k), device="cuda", dtype=torch.float32)
    L.check(L.lib().ardae_pack_weight(L.ptr(W), W.stride(0), nout, k, int(transpose), L.ptr(out), L.stream_ptr()))
    return out


def run_linear(epi, M, Nout, srcs, **kw):
    a = L.LinearArgs()
    a.M, a.Nout, a.nsrc = M, Nout, len(srcs)
    keep = []
    for i, (x, wp) in enumerate(srcs):
        a.src[i].x = x.data_ptr(); a.src[i].ld = x.stride(0); a.src[i].K = x.shape[1]; a.src[i].wp = wp.data_ptr()
        keep += [x, wp]
    for k, v in kw.items():
        if torch.is_tensor(v):
            keep.append(v)
            setattr(a, k, v.data_ptr())
            ldname = {"S": "ldS", "R": "ldR", "Q": "ldQ", "eps": "ldeps", "Y": "ldY", "Y2": "ldY2", "rowbias": "rowbias_ld"}.get(k)
            if ldname:
                setattr(a, ldname, v.stride(0))
        else:
            setattr(a, k, v)
    L.check(L.lib().ardae_linear(ctypes.byref(a), epi, L.stream_ptr()), "ardae_linear")
    torch.cuda.synchronize()


def relerr(a, b):
    return float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))


def d1(act, a):
    if act == "softplus":
        return -torch.expm1(-a)
    if act == "relu":
        return (a > 0).double()
    return torch.ones_like(a)


@pytest.mark.parametrize("M,K,Nout", [(64, 256, 256), (200, 100, 256), (131, 37, 96), (512, 784, 256),
                                      (96, 256, 32), (300, 64, 8), (64, 2, 64), (1024, 512, 512), (70, 266, 2),
                                      # big-M shapes: the software-pipelined wide kernel where the shape allows, else the generic 64 x 256 tiling
                                      (4096, 256, 256), (8192, 32, 256), (4128, 256, 64), (4096, 64, 192)])
@pytest.mark.parametrize("act", ["none", "relu", "softplus"])
def test_linear_act(M, K, Nout, act):
    g = torch.Generator().manual_seed(M * 7 + K * 3 + Nout)
    X = torch.randn(M, K, generator=g)
    W = torch.randn(Nout, K, generator=g) / K ** 0.5
    b = torch.randn(Nout, generator=g)
    rpg = 8
    rb = torch.randn((M + rpg - 1) // rpg, Nout, generator=g)
    sig = torch.randn(M, generator=g)
    wsig = torch.randn(Nout, generator=g)
    pre = X.double() @ W.double().T + b.double() + rb.double().repeat_interleave(rpg, 0)[:M] + sig.double()[:, None] * wsig.double()
    ref = {"none": pre, "relu": pre.clamp(min=0), "softplus": torch.nn.functional.softplus(pre)}[act]
    Xd, Wd = X.cuda(), W.cuda()
    Y = torch.full((M, Nout), float("nan"), device="cuda")
    run_linear(L.EPI_ACT, M, Nout, [(Xd, pack(Wd))], act=L.ACT[act], bias=b.cuda(), rowbias=rb.cuda(), rows_per_group=rpg,
               rowscale=sig.cuda(), rowscale_w=wsig.cuda(), Y=Y)
    assert relerr(Y, ref) < 2e-5


def test_linear_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C/D or operand map (guide section 3)."""
    n = 64
    X = torch.eye(n)
    W = (torch.arange(n * n, dtype=torch.float32).view(n, n) % 97) - 40.0
    Y = torch.empty(n, n, device="cuda")
    run_linear(L.EPI_ACT, n, n, [(X.cuda(), pack(W.cuda()))], act=0, Y=Y)
    assert torch.equal(Y.cpu(), W.T.contiguous())


def test_linear_transposed_pack_and_two_sources():
    g = torch.Generator().manual_seed(5)
    M, K1, K2, Nout = 192, 256, 10, 256
    X1, X2 = torch.randn(M, K1, generator=g), torch.randn(M, K2, generator=g)
    Wfull = torch.randn(Nout, K1 + K2, generator=g) / 16            # one [out, in] matrix, two column slices
    Wt = torch.randn(K1, Nout, generator=g) / 16                     # used transposed (backward-style product)
    Wf = Wfull.cuda()
    Y = torch.empty(M, Nout, device="cuda")
    run_linear(L.EPI_ACT, M, Nout, [(X1.cuda(), pack(Wf[:, :K1])), (X2.cuda(), pack(Wf[:, K1:]))], act=0, Y=Y)
    ref = torch.cat([X1, X2], 1).double() @ Wfull.double().T
    assert relerr(Y, ref) < 2e-5
    run_linear(L.EPI_ACT, M, Nout, [(X1.cuda(), pack(Wt.cuda(), transpose=True))], act=0, Y=Y)
    assert relerr(Y, X1.double() @ Wt.double()) < 2e-5


@pytest.mark.parametrize("M,K1,K2,Nout,act", [(64, 256, 100, 256, "softplus"), (128, 256, 100, 256, "relu"), (32, 784, 100, 300, "softplus"), (16, 256, 10, 48, "none"),
                                              (512, 256, 100, 256, "softplus")])
def test_linear_two_source_per_image_layers(M, K1, K2, Nout, act):
    """Concat inputs [hidden | noise] of the per-image sampler layers (models/layers.py:501-515 on a few rows): two (activation, weight)
    pairs summed into one output - on the 16 x 16 blocks where the layer is small (each source's steps split over the four waves, partial
    last step for K = 100 / 10), on the generic block otherwise (512 rows); against float64."""
    g = torch.Generator().manual_seed(M + K1 + K2 + Nout)
    X1, X2 = torch.randn(M, K1, generator=g), torch.randn(M, K2, generator=g)
    W = torch.randn(Nout, K1 + K2, generator=g) / (K1 + K2) ** 0.5
    b = torch.randn(Nout, generator=g)
    pre = torch.cat([X1, X2], 1).double() @ W.double().T + b.double()
    ref = {"none": pre, "relu": pre.clamp(min=0), "softplus": torch.nn.functional.softplus(pre)}[act]
    Wd = W.cuda()
    Y = torch.full((M, Nout), float("nan"), device="cuda")
    run_linear(L.EPI_ACT, M, Nout, [(X1.cuda(), pack(Wd[:, :K1])), (X2.cuda(), pack(Wd[:, K1:]))], act=L.ACT[act], bias=b.cuda(), Y=Y)
    assert relerr(Y, ref) < 2e-5


@pytest.mark.parametrize("M,Nout", [(2048, 1024), (8192, 1024), (6144, 512), (20480, 128)])
@pytest.mark.parametrize("epi", ["dact", "act_rowbias"])
def test_linear_wide_kernel_k1024_rolling_slab(M, Nout, epi):
    """K = 1024 (config #5's mlp-res h 1024 layers): the wave's slab, 32 columns x 1024, is twice its AGPR file, so the kernel keeps a
    rolling WINDOW of 64 chunks - every slot is reloaded one chunk after the MFMAs that read it, with the fragment 64 chunks ahead
    (this tile's second half, then the next tile's first).  One tile per workgroup (2048 rows x 8 column panels), four, a ragged
    number of tiles per workgroup and a single column panel; against float64.  The instantiated epilogues: forward with bias /
    per-image row bias + sigma term, backward DACT."""
    K = 1024
    g = torch.Generator().manual_seed(M + Nout)
    X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / K ** 0.5
    v = X.double() @ W.double().T
    wpk = pack(W.cuda())
    Y = torch.full((M, Nout), float("nan"), device="cuda")
    if epi == "dact":
        S = torch.nn.functional.softplus(torch.randn(M, Nout, generator=g) * 3)
        run_linear(L.EPI_DACT, M, Nout, [(X.cuda(), wpk)], act=2, S=S.cuda(), Y=Y)
        ref = v * d1("softplus", S.double())
    else:
        rpg = 256
        rb, sig, ws = torch.randn(M // rpg, Nout, generator=g), torch.randn(M, generator=g).abs(), torch.randn(Nout, generator=g)
        run_linear(L.EPI_ACT, M, Nout, [(X.cuda(), wpk)], act=2, rowbias=rb.cuda(), rows_per_group=rpg, rowscale=sig.cuda(), rowscale_w=ws.cuda(), Y=Y)
        ref = torch.nn.functional.softplus(v + rb.double().repeat_interleave(rpg, 0) + sig.double()[:, None] * ws.double()[None, :])
    assert not torch.isnan(Y).any()
    assert relerr(Y, ref) < 2e-5


@pytest.mark.parametrize("M,K,Nout", [(8192, 256, 256), (24576, 256, 256), (65536, 256, 256), (16384, 32, 256),
                                      (8192, 512, 512), (20480, 512, 512), (4096, 512, 1024), (8192, 32, 512), (12288, 512, 128),
                                      (65536, 32, 256), (40960, 32, 512)])
@pytest.mark.parametrize("epi", ["dact", "dact_q", "chain", "act_seed"])
def test_linear_wide_kernel_epilogues(M, K, Nout, epi):
    """The weight-stationary N-row kernel (linear_wide_kernel: >= 128 tiles, K in {256, 32} with 256 columns per workgroup, K = 512
    with 128 columns per workgroup - configs #4's h 512 layers - in 2, 4 and 8 column panels, the XCD-aware and the plain
    workgroup -> tile maps): one tile per workgroup (8192 rows), one or two (24576: deferred stores carried into the next
    tile), four (65536; with K = 32 the tile loop is a rolled loop whose spilled SGPRs once reached a buffer instruction too early)."""
    test_linear_big_m_epilogues(M, K, Nout, epi, rpg=64)


@pytest.mark.parametrize("epi", ["act", "dact_q", "chain"])
def test_linear_wide_kernel_full_size_matches_generic_kernel(epi):
    """BASELINE config #2 size (131072 rows = 512 images x 256 samples, 8 tiles per workgroup): the software-pipelined
    kernel against the generic kernel on the same data.  The generic path is forced by a 4-byte-misaligned copy of X (the
    wide kernel needs 16-byte aligned rows); both are fp32 fmaf chains over k in the same order, so they agree to rounding
    of the epilogue only."""
    M, K, Nout = 131072, 256, 256
    g = torch.Generator(device="cuda").manual_seed(7)
    X = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(Nout, K, device="cuda", generator=g) / K ** 0.5
    S = torch.nn.functional.softplus(torch.randn(M, Nout, device="cuda", generator=g) * 3)
    Q = torch.randn(M, Nout, device="cuda", generator=g); R = torch.randn(M, Nout, device="cuda", generator=g)
    b = torch.randn(Nout, device="cuda", generator=g)
    Xu = torch.empty(M * K + 1, device="cuda")[1:].view(M, K)   # same values, rows start 4 bytes off a 16-byte boundary
    Xu.copy_(X)
    wpk = pack(W)
    outs = []
    for x in (X, Xu):
        Y = torch.full((M, Nout), float("nan"), device="cuda"); Y2 = torch.full((M, Nout), float("nan"), device="cuda")
        if epi == "act":
            run_linear(L.EPI_ACT, M, Nout, [(x, wpk)], act=2, bias=b, Y=Y)
        elif epi == "dact_q":
            Y.copy_(Q)
            run_linear(L.EPI_DACT, M, Nout, [(x, wpk)], act=2, S=S, Q=Y, Y=Y)       # in place
        else:
            run_linear(L.EPI_CHAIN, M, Nout, [(x, wpk)], act=2, S=S, R=R, Y=Y, Y2=Y2)
        outs.append((Y, Y2))
    (Ya, Y2a), (Yb, Y2b) = outs
    assert not torch.isnan(Ya).any()
    assert float((Ya - Yb).abs().max() / Yb.abs().max()) < 2e-6
    if epi == "chain":
        assert float((Y2a - Y2b).abs().max() / Y2b.abs().max()) < 2e-6


@pytest.mark.parametrize("M,K,Nout", [(4096, 256, 256), (12288, 32, 256), (4096, 256, 128)])
@pytest.mark.parametrize("epi", ["dact", "dact_q", "chain", "act_seed"])
def test_linear_big_m_epilogues(M, K, Nout, epi, rpg=16):
    """The derivative / score-seed epilogues at a few thousand rows (M >= 4096, K % 32 == 0), against float64."""
    g = torch.Generator().manual_seed(M + K + Nout)
    X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / K ** 0.5
    S = torch.nn.functional.softplus(torch.randn(M, Nout, generator=g) * 3)
    Q = torch.randn(M, Nout, generator=g); R = torch.randn(M, Nout, generator=g)
    v = X.double() @ W.double().T
    s1 = d1("softplus", S.double())
    Y = torch.full((M, Nout), float("nan"), device="cuda"); Y2 = torch.full((M, Nout), float("nan"), device="cuda")
    wpk = pack(W.cuda())
    if epi == "dact":
        run_linear(L.EPI_DACT, M, Nout, [(X.cuda(), wpk)], act=2, S=S.cuda(), Y=Y)
        assert relerr(Y, v * s1) < 2e-5
    elif epi == "dact_q":
        Qd = Q.cuda()
        run_linear(L.EPI_DACT, M, Nout, [(X.cuda(), wpk)], act=2, S=S.cuda(), Q=Qd, Y=Qd)       # in place, as the backward uses it
        assert relerr(Qd, v * s1 + Q.double()) < 2e-5
    elif epi == "chain":
        run_linear(L.EPI_CHAIN, M, Nout, [(X.cuda(), wpk)], act=2, S=S.cuda(), R=R.cuda(), Y=Y, Y2=Y2)
        assert relerr(Y, v * s1) < 2e-5
        assert relerr(Y2, v * R.double() * (1 - s1)) < 2e-5
    else:
        w = torch.randn(Nout, generator=g); b = torch.randn(Nout, generator=g)
        rb = torch.randn(M // rpg, Nout, generator=g); sig = torch.randn(M, generator=g); wsig = torch.randn(Nout, generator=g)
        pre = v + b.double() + rb.double().repeat_interleave(rpg, 0) + sig.double()[:, None] * wsig.double()
        run_linear(L.EPI_ACT, M, Nout, [(X.cuda(), wpk)], act=2, bias=b.cuda(), rowbias=rb.cuda(), rows_per_group=rpg,
                   rowscale=sig.cuda(), rowscale_w=wsig.cuda(), R=w.cuda(), Y=Y, Y2=Y2)
        assert relerr(Y, torch.nn.functional.softplus(pre)) < 2e-5
        assert relerr(Y2, -w.double() * torch.sigmoid(pre)) < 2e-5


@pytest.mark.parametrize("M,K,Nout", [(64, 256, 256),      # 16 blocks of 32 x 32: the 16 x 16 blocks (few rows: small_block16)
                                      (16, 32, 48), (128, 784, 256), (48, 256, 784),     # K = 32: two waves idle; 784 = 49 blocks of 16 columns
                                      (512, 256, 256), (256, 784, 256),                 # 32 x 32 lean blocks (small_block_fast), 24.5 chunks per wave
                                      (64, 100, 256), (64, 256, 100), (128, 300, 300), (32, 784, 300), (48, 20, 36),   # ragged K (a partial last step of 16 k) / ragged columns: still the 16 x 16 blocks
                                      (100, 256, 256), (64, 102, 256), (512, 256, 100)])  # ragged rows / K % 4 / many ragged blocks: the generic block
@pytest.mark.parametrize("epi", ["dact", "dact_q", "chain", "act_seed"])
def test_linear_per_image_kernels(M, K, Nout, epi):
    """The latency kernels of the per-image (B-row) layers (linear_small.hip) - 16 x 16 blocks for layers of at most 64 blocks of 32 x 32,
    the lean 32 x 32 block for regular shapes, the generic block for the rest - all epilogues, against float64."""
    test_linear_big_m_epilogues(M, K, Nout, epi, rpg=4 if M % 4 == 0 else 1)


@pytest.mark.parametrize("Ks,Nout", [((256,), 256), ((784,), 256), ((256,), 784), ((256,), 32), ((100,), 256), ((24,), 64), ((8,), 32),
                                     ((256, 100), 256), ((256, 10), 256), ((784, 100), 300), ((40, 12), 48)])
@pytest.mark.parametrize("epi", ["act", "dact_q", "chain"])
def test_linear_per_image_blocks_bit_identical(Ks, Nout, epi):
    """A per-image (B-row) layer gives a row the SAME BITS whatever block shape it runs on, i.e. however many rows (images per rank) the
    launch has: 16 x 16 blocks (few rows), the lean 32 x 32 block (regular shapes) and the generic 32 x 32 block (ragged rows / K /
    two sources) add one sequence of products per output element (linear_small.hip, `CANONICAL k ORDER`; an FP32 MFMA is a chain of
    fused multiply-adds over its k ascending - scratch/mfma/order.hip).  This is what makes the data-parallel step independent of the
    shard size up to the order of the sums over ROWS (the seed's B-row layers: ivae_ardae.py:826-834)."""
    g = torch.Generator().manual_seed(sum(Ks) + Nout)
    Mbig = 512
    Xs = [torch.randn(Mbig + 32, K, generator=g).cuda() for K in Ks]
    W = (torch.randn(Nout, sum(Ks), generator=g) / sum(Ks) ** 0.5).cuda()
    wps, off = [], 0
    for K in Ks:
        wps.append(pack(W[:, off:off + K])); off += K
    S = torch.nn.functional.softplus(torch.randn(Mbig + 32, Nout, generator=g) * 3).cuda()
    Q = torch.randn(Mbig + 32, Nout, generator=g).cuda(); R = torch.randn(Mbig + 32, Nout, generator=g).cuda()
    b = torch.randn(Nout, generator=g).cuda(); rb = torch.randn((Mbig + 32) // 4, Nout, generator=g).cuda()
    sig = torch.randn(Mbig + 32, generator=g).cuda(); wsig = torch.randn(Nout, generator=g).cuda()

    def run(M):
        Y = torch.full((M, Nout), float("nan"), device="cuda"); Y2 = torch.full((M, Nout), float("nan"), device="cuda")
        srcs = [(X[:M], wp) for X, wp in zip(Xs, wps)]
        if epi == "act":
            run_linear(L.EPI_ACT, M, Nout, srcs, act=2, bias=b, rowbias=rb, rows_per_group=4, rowscale=sig, rowscale_w=wsig, Y=Y)
            return (Y,)
        if epi == "dact_q":
            run_linear(L.EPI_DACT, M, Nout, srcs, act=2, S=S, Q=Q, Y=Y)
            return (Y,)
        run_linear(L.EPI_CHAIN, M, Nout, srcs, act=2, S=S, R=R, Y=Y, Y2=Y2)
        return (Y, Y2)

    # 512 rows: 32 x 32 blocks (lean where the shape is regular, generic otherwise); 500 / 529: ragged rows, always the generic block;
    # 16 .. 64 rows: 16 x 16 blocks (at most 64 blocks of 32 x 32 in the layer; Nout = 784 only up to 64 rows)
    ref = run(Mbig)
    for M in (16, 32, 64, 128, 500, 529):
        got = run(M)
        m = min(M, Mbig)
        for r, o in zip(ref, got):
            assert not torch.isnan(o).any()
            assert torch.equal(r[:m], o[:m]), (M, float((r[:m] - o[:m]).abs().max()))


@pytest.mark.parametrize("act", ["relu", "softplus"])
def test_linear_dact_and_colsum(act):
    g = torch.Generator().manual_seed(11)
    M, K, Nout = 200, 256, 256
    X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / 16
    S = torch.nn.functional.softplus(torch.randn(M, Nout, generator=g) * 3) if act == "softplus" else torch.randn(M, Nout, generator=g).clamp(min=0)
    Q = torch.randn(M, Nout, generator=g)
    ref = (X.double() @ W.double().T) * d1(act, S.double()) + Q.double()
    Y = torch.empty(M, Nout, device="cuda")
    tiles = L.lib().ardae_linear_row_tiles(M, Nout)
    cs = torch.zeros(tiles, Nout, device="cuda")
    run_linear(L.EPI_DACT, M, Nout, [(X.cuda(), pack(W.cuda()))], act=L.ACT[act], S=S.cuda(), Q=Q.cuda(), Y=Y, colsum=cs)
    assert relerr(Y, ref) < 2e-5
    assert relerr(cs.sum(0), ref.sum(0)) < 2e-5


def test_linear_chain():
    g = torch.Generator().manual_seed(12)
    M, K, Nout = 128, 256, 256
    X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / 16
    S = torch.nn.functional.softplus(torch.randn(M, Nout, generator=g) * 3)
    R = torch.randn(M, Nout, generator=g)
    v = X.double() @ W.double().T
    s = d1("softplus", S.double())
    Y = torch.empty(M, Nout, device="cuda"); Y2 = torch.empty(M, Nout, device="cuda")
    run_linear(L.EPI_CHAIN, M, Nout, [(X.cuda(), pack(W.cuda()))], act=2, S=S.cuda(), R=R.cuda(), Y=Y, Y2=Y2)
    assert relerr(Y, v * s) < 2e-5
    assert relerr(Y2, v * R.double() * (1 - s)) < 2e-5


@pytest.mark.parametrize("z", [32, 8, 2])
def test_linear_dae_loss(z):
    g = torch.Generator().manual_seed(13)
    M, K = 300, 256
    X = torch.randn(M, K, generator=g); W = torch.randn(z, K, generator=g) / 16
    sig = torch.randn(M, generator=g); eps = torch.randn(M, z, generator=g)
    gref = X.double() @ W.double().T
    rho = sig.double()[:, None] * gref + eps.double()
    scale = 1.0 / (M * z)
    Y = torch.empty(M, z, device="cuda"); Y2 = torch.empty(M, z, device="cuda")
    tiles = L.lib().ardae_linear_row_tiles(M, z) * L.lib().ardae_linear_col_panels(M, z)
    tl = torch.zeros(tiles, device="cuda")
    run_linear(L.EPI_DAE_LOSS, M, z, [(X.cuda(), pack(W.cuda()))], sigma=sig.cuda(), eps=eps.cuda(), scale=scale, Y=Y, Y2=Y2, tile_loss=tl)
    assert relerr(Y, gref) < 2e-5
    assert relerr(Y2, 2 * sig.double()[:, None] * rho * scale) < 2e-5
    assert abs(float(tl.sum().cpu()) - float((rho ** 2).sum())) / float((rho ** 2).sum()) < 1e-5


@pytest.mark.parametrize("M,z,with_bias", [(1024, 32, False), (4096, 8, True), (65536, 32, True)])
def test_linear_narrow_kernel_dae_loss(M, z, with_bias):
    """linear_narrow_kernel (K = 256 -> Nout <= 32 on whole 128-row tiles, A fragments straight into registers) with the
    DAE-loss epilogue, against float64: the mlp-grad score product (no bias) and the mlp-res head (bias)."""
    g = torch.Generator().manual_seed(M + z)
    K = 256
    X = torch.randn(M, K, generator=g); W = torch.randn(z, K, generator=g) / 16
    b = torch.randn(z, generator=g) if with_bias else None
    sig = torch.randn(M, generator=g); eps = torch.randn(M, z, generator=g)
    gref = X.double() @ W.double().T + (b.double() if with_bias else 0.0)
    rho = sig.double()[:, None] * gref + eps.double()
    scale = 1.0 / (M * z)
    Y = torch.full((M, z), float("nan"), device="cuda"); Y2 = torch.full((M, z), float("nan"), device="cuda")
    tiles = L.lib().ardae_linear_row_tiles(M, z) * L.lib().ardae_linear_col_panels(M, z)
    assert tiles == M // 128
    tl = torch.zeros(tiles, device="cuda")
    kw = dict(bias=b.cuda()) if with_bias else {}
    run_linear(L.EPI_DAE_LOSS, M, z, [(X.cuda(), pack(W.cuda()))], sigma=sig.cuda(), eps=eps.cuda(), scale=scale, Y=Y, Y2=Y2, tile_loss=tl, **kw)
    assert relerr(Y, gref) < 2e-5
    assert relerr(Y2, 2 * sig.double()[:, None] * rho * scale) < 2e-5
    assert abs(float(tl.double().sum().cpu()) - float((rho ** 2).sum())) / float((rho ** 2).sum()) < 1e-5


@pytest.mark.parametrize("M,z", [(32768, 32), (65536, 20)])
def test_linear_narrow_kernel_bias_epilogue(M, z):
    """The sampler's output layer on N rows (256 -> z, bias, no activation): too many rows for the per-image kernel, so it
    streams through linear_narrow_kernel."""
    g = torch.Generator().manual_seed(M + z)
    X = torch.randn(M, 256, generator=g); W = torch.randn(z, 256, generator=g) / 16; b = torch.randn(z, generator=g)
    Y = torch.full((M, z), float("nan"), device="cuda")
    run_linear(L.EPI_ACT, M, z, [(X.cuda(), pack(W.cuda()))], act=0, bias=b.cuda(), Y=Y)
    assert relerr(Y, X.double() @ W.double().T + b.double()) < 2e-5


@pytest.mark.parametrize("K,Nout,rpg", [(100, 256, 256), (100, 256, 48), (36, 64, 32), (124, 512, 128)])
@pytest.mark.parametrize("act", ["none", "relu", "softplus"])
def test_linear_shortk_kernel(K, Nout, rpg, act):
    """linear_shortk_kernel (N-row forward layers with K <= 128 that is not a multiple of 32: the sampler's noise layer,
    K = 100): bias + per-image row bias (groups aligned to the 32-row wave tiles or not) + activation, against float64."""
    M = 32768
    g = torch.Generator().manual_seed(K * 3 + Nout + rpg)
    X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / K ** 0.5; b = torch.randn(Nout, generator=g)
    rb = torch.randn((M + rpg - 1) // rpg, Nout, generator=g)
    pre = X.double() @ W.double().T + b.double() + rb.double().repeat_interleave(rpg, 0)[:M]
    ref = {"none": pre, "relu": pre.clamp(min=0), "softplus": torch.nn.functional.softplus(pre)}[act]
    Y = torch.full((M, Nout), float("nan"), device="cuda")
    run_linear(L.EPI_ACT, M, Nout, [(X.cuda(), pack(W.cuda()))], act=L.ACT[act], bias=b.cuda(), rowbias=rb.cuda(), rows_per_group=rpg, Y=Y)
    assert relerr(Y, ref) < 2e-5


def test_linear_generic_kernel_fallback_paths():
    """The shape-specialised kernels (per-image split-K, narrow, short-K) take their problems away from linear_kernel; with
    them switched off (the library reads its knobs once per process) every test of this file must still pass on the generic
    kernel, which stays the path for everything ragged."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, ARDAE_DEBUG_KNOBS="1", ARDAE_SMALL="0", ARDAE_NARROW="0", ARDAE_SHORTK="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.abspath(__file__), "-k", "not fallback_paths"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_linear_rejects_bad_args():
    a = L.LinearArgs()
    with pytest.raises(ValueError):
        L.check(L.lib().ardae_linear(ctypes.byref(a), 0, None))


# ---- the activations beyond relu / softplus (get_nonlinear_func, utils/models.py:14-32): elu, tanh, leaky_relu(0.2) on the generic
# ---- kernels - forward, derivative from the saved output, and the second-over-first derivative ratio of EPI_CHAIN
def _fwd64(act, pre):
    F = torch.nn.functional
    return {"elu": F.elu(pre), "tanh": torch.tanh(pre), "leaky_relu": F.leaky_relu(pre, 0.2), "relu": pre.clamp(min=0), "softplus": F.softplus(pre),
            "swish": pre * torch.sigmoid(pre)}[act]


def _d1_ratio64(act, pre):
    """s = act'(pre) and s'/s as float64 functions of the PRE-activation."""
    x = pre.clone().requires_grad_(True)
    y = _fwd64(act, x)
    (s,) = torch.autograd.grad(y.sum(), x, create_graph=True)
    s2 = torch.autograd.grad(s.sum(), x, allow_unused=True)[0] if s.requires_grad else None     # piecewise linear: s is a constant of x
    s2 = torch.zeros_like(x) if s2 is None else s2
    return s.detach(), torch.where(s.detach() != 0, s2 / s.detach(), torch.zeros_like(s2))


@pytest.mark.parametrize("M,K,Nout", [(200, 100, 256), (64, 256, 256), (4096, 256, 256), (96, 256, 32), (8192, 32, 256)])
@pytest.mark.parametrize("act", ["elu", "tanh", "leaky_relu", "swish"])
def test_linear_act_more_activations(M, K, Nout, act):
    g = torch.Generator().manual_seed(M + K + Nout)
    X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / K ** 0.5 * 2; b = torch.randn(Nout, generator=g)
    w = torch.randn(Nout, generator=g)
    pre = X.double() @ W.double().T + b.double()
    Y = torch.full((M, Nout), float("nan"), device="cuda"); Y2 = torch.full((M, Nout), float("nan"), device="cuda")
    run_linear(L.EPI_ACT, M, Nout, [(X.cuda(), pack(W.cuda()))], act=L.ACT[act], bias=b.cuda(), R=w.cuda(), Y=Y, Y2=Y2)
    assert relerr(Y, _fwd64(act, pre)) < 2e-5
    s, _ = _d1_ratio64(act, pre)
    # score seed -w (.) act'(a), rebuilt from the saved output (swish: inverted on the branch its lowest bit names; ill-conditioned within
    # ~1e-3 of the minimum, where |swish' error| <= 1e-4)
    assert relerr(Y2, -w.double() * s) < (2e-4 if act == "swish" else 5e-5)


@pytest.mark.parametrize("M", [200, 4096])
@pytest.mark.parametrize("act", ["elu", "tanh", "leaky_relu", "relu"])
def test_linear_dact_chain_more_activations(M, act):
    g = torch.Generator().manual_seed(21 + M)
    K, Nout = 256, 256
    X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / 16
    pre = torch.randn(M, Nout, generator=g).double() * 2
    S = _fwd64(act, pre).float()                            # the saved OUTPUT of the layer
    s, ratio = _d1_ratio64(act, pre)
    Q = torch.randn(M, Nout, generator=g); R = torch.randn(M, Nout, generator=g)
    v = X.double() @ W.double().T
    Y = torch.empty(M, Nout, device="cuda"); Y2 = torch.empty(M, Nout, device="cuda")
    run_linear(L.EPI_DACT, M, Nout, [(X.cuda(), pack(W.cuda()))], act=L.ACT[act], S=S.cuda(), Q=Q.cuda(), Y=Y)
    assert relerr(Y, v * s + Q.double()) < 5e-5
    run_linear(L.EPI_CHAIN, M, Nout, [(X.cuda(), pack(W.cuda()))], act=L.ACT[act], S=S.cuda(), R=R.cuda(), Y=Y, Y2=Y2)
    assert relerr(Y, v * s) < 5e-5
    assert relerr(Y2, v * R.double() * ratio) < 5e-5


@pytest.mark.parametrize("M", [200, 4096])
def test_linear_dact_chain_swish(M):
    """swish (utils/models.py:8-10) through the derivative epilogues: the saved output carries the branch of x sigmoid(x) in its lowest
    mantissa bit (csrc/common.h::swish_f), act' and act''/act' are rebuilt by inverting it.  S is written by the FORWARD epilogue here
    (as in a step); R carries act' like the tensor the forward-mode pass multiplies (so that R act''/act' = t act'' is well conditioned
    at the minimum, where act' vanishes).  Also: the stored output is swish(pre) to fp32 accuracy, over a dense sweep of both branches,
    the minimum and the tails."""
    g = torch.Generator().manual_seed(33 + M)
    K, Nout = 256, 256
    X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / 16
    pre = (torch.randn(M, Nout, generator=g) * 2.5).float()
    pre.view(-1)[:4001] = torch.linspace(-100.0, 30.0, 4001)                                        # tails on both sides
    pre.view(-1)[4001:8002] = torch.linspace(-1.2784645 - 2e-3, -1.2784645 + 2e-3, 4001)           # around the minimum
    eye = torch.eye(Nout)
    S = torch.full((M, Nout), float("nan"), device="cuda")
    run_linear(L.EPI_ACT, M, Nout, [(pre.cuda(), pack(eye.cuda()))], act=L.ACT["swish"], Y=S)        # S = swish(pre), branch bit included
    ref_y = _fwd64("swish", pre.double())
    # fp32 accuracy of x sigmoid(x) on the hardware exp unit (the argument's rounding costs |x| 6e-8 relative in exp(-|x|)), plus the one
    # ulp the branch bit may move the stored value by; exp(-|x|) below the normal range is flushed (results under ~1e-36 come out as the bare branch bit)
    tol = torch.maximum(ref_y.abs() * (5e-7 + 2e-7 * pre.double().abs()), torch.tensor(1e-35, dtype=torch.float64))
    assert bool(((S.cpu().double() - ref_y).abs() <= tol).all())
    s, _ = _d1_ratio64("swish", pre.double())
    x = pre.double().clone().requires_grad_(True)
    (s_x,) = torch.autograd.grad(_fwd64("swish", x).sum(), x, create_graph=True)
    (s2,) = torch.autograd.grad(s_x.sum(), x)
    Q = torch.randn(M, Nout, generator=g); t = torch.randn(M, Nout, generator=g)
    v = X.double() @ W.double().T
    Y = torch.empty(M, Nout, device="cuda"); Y2 = torch.empty(M, Nout, device="cuda")
    # R = t act'(S) as the score pass leaves it: made by the derivative epilogue itself (the inverted x is the same in both launches,
    # so act' cancels exactly in R act''/act' - also within 1e-3 of the minimum, where the inversion cannot resolve act')
    Rd = torch.empty(M, Nout, device="cuda")
    run_linear(L.EPI_DACT, M, Nout, [(t.cuda(), pack(eye.cuda()))], act=L.ACT["swish"], S=S, Y=Rd)
    assert relerr(Rd, t.double() * s) < 1e-4
    run_linear(L.EPI_DACT, M, Nout, [(X.cuda(), pack(W.cuda()))], act=L.ACT["swish"], S=S, Q=Q.cuda(), Y=Y)
    assert relerr(Y, v * s + Q.double()) < 1e-4
    run_linear(L.EPI_CHAIN, M, Nout, [(X.cuda(), pack(W.cuda()))], act=L.ACT["swish"], S=S, R=Rd, Y=Y, Y2=Y2)
    assert relerr(Y, v * s) < 1e-4
    assert torch.isfinite(Y2).all()
    assert relerr(Y2, v * t.double() * s2) < 5e-4


@pytest.mark.parametrize("M,K,Nout,rpg", [(80000, 100, 256, 625),      # short-K kernel, groups of 625 rows (the shipped recipes' nz_cdae)
                                          (80000, 256, 256, 625), (40000, 512, 512, 625), (16000, 256, 256, 125),   # N-row kernel, "group tiles"
                                          (4000, 64, 96, 40), (4096, 256, 256, 100), (8192, 512, 512, 625), (640, 256, 64, 33), (300, 37, 40, 7)])
def test_linear_rowbias_groups_not_tile_aligned(M, K, Nout, rpg):
    """Per-image row bias with groups that are no multiple of the 32-row blocks / 64-row tiles: a block meets two images (one division
    per block + a compare per row for groups of >= 32 rows, the per-element form below that)."""
    g = torch.Generator().manual_seed(M + K + rpg)
    X = torch.randn(M, K, generator=g); W = torch.randn(Nout, K, generator=g) / K ** 0.5; b = torch.randn(Nout, generator=g)
    ngroups = (M + rpg - 1) // rpg
    rb = torch.randn(ngroups, Nout, generator=g) * 2
    pre = X.double() @ W.double().T + b.double() + rb.double().repeat_interleave(rpg, 0)[:M]
    Y = torch.full((M, Nout), float("nan"), device="cuda")
    run_linear(L.EPI_ACT, M, Nout, [(X.cuda(), pack(W.cuda()))], act=L.ACT["softplus"], bias=b.cuda(), rowbias=rb.cuda(), rows_per_group=rpg, Y=Y)
    assert relerr(Y, torch.nn.functional.softplus(pre)) < 2e-5


@pytest.mark.parametrize("M,K,Nout,rpg", [(80000, 256, 256, 625), (80000, 100, 256, 625), (65536, 256, 256, 256)])
def test_nrow_kernels_are_deterministic(M, K, Nout, rpg):
    """Same inputs, five launches, bit-identical outputs: the group-tile mode writes the rows two tiles of an image share twice (from two
    workgroups, identical values), the short-K kernel selects between two row-bias values per row - neither may depend on timing."""
    g = torch.Generator().manual_seed(K + rpg)
    X = torch.randn(M, K, generator=g).cuda(); W = (torch.randn(Nout, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(Nout, generator=g).cuda()
    rb = (torch.randn((M + rpg - 1) // rpg, Nout, generator=g) * 2).cuda()
    wp = pack(W)
    outs = []
    for _ in range(5):
        Y = torch.full((M, Nout), float("nan"), device="cuda")
        run_linear(L.EPI_ACT, M, Nout, [(X, wp)], act=L.ACT["softplus"], bias=b, rowbias=rb, rows_per_group=rpg, Y=Y)
        outs.append(Y)
    assert not torch.isnan(outs[0]).any()
    for Y in outs[1:]:
        assert torch.equal(Y, outs[0])


def _make_args(M, Nout, x, wp, **kw):
    a = L.LinearArgs()
    a.M, a.Nout, a.nsrc = M, Nout, 1
    a.src[0].x = x.data_ptr(); a.src[0].ld = x.stride(0); a.src[0].K = x.shape[1]; a.src[0].wp = wp.data_ptr()
    for k, v in kw.items():
        if torch.is_tensor(v):
            setattr(a, k, v.data_ptr())
            ldname = {"S": "ldS", "R": "ldR", "Q": "ldQ", "Y": "ldY", "Y2": "ldY2", "rowbias": "rowbias_ld"}.get(k)
            if ldname:
                setattr(a, ldname, v.stride(0))
        else:
            setattr(a, k, v)
    return a


@pytest.mark.parametrize("M", [8192, 16384, 24576])
@pytest.mark.parametrize("kind", ["act", "act_energy", "act_energy_noseed", "dact", "dact_q", "chain"])
@pytest.mark.parametrize("nl", [2, 3, 5])
def test_linear_chain_kernel_equals_per_layer_launches(M, kind, nl):
    """linear_chain_kernel (a row-local run of h x h layers in ONE launch, the tile handed from layer to layer through LDS, the
    next layer's weight slab reloaded behind the MFMAs) against nl launches of ardae_linear on the same data: same MFMA order,
    same epilogue code - bit-identical, every tensor of every layer.  One tile per workgroup (16384 rows = the 8-rank shard of
    config #2), half a chip (8192) and one-or-two tiles (24576); 2 layers (first + last body), 3 (one pass of the rolled middle
    body) and 5 (the score / forward-mode / backward runs of L = 3); first layer with row bias + sigma term, last with the score
    seed; the last CHAIN layer with per-tile column sums."""
    h, act = 256, 2
    g = torch.Generator(device="cuda").manual_seed(M + nl)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    X = rn(M, h)
    Ws = [rn(h, h) / h ** 0.5 for _ in range(nl)]
    wps = [pack(W) for W in Ws]
    S = [torch.nn.functional.softplus(rn(M, h) * 3) for _ in range(nl)]
    R = [rn(M, h) for _ in range(nl)]
    Q0 = [rn(M, h) for _ in range(nl)]
    bias = [rn(h) for _ in range(nl)]
    rpg = 256
    rowbias, sigma, wsig, wfc = rn(M // rpg, h), rn(M).abs(), rn(h), rn(h)
    epi = {"act": L.EPI_ACT, "act_energy": L.EPI_ACT, "act_energy_noseed": L.EPI_ACT, "dact": L.EPI_DACT, "dact_q": L.EPI_DACT, "chain": L.EPI_CHAIN}[kind]
    ntile = M // 64

    def build(tag):
        """LinearArgs of the nl layers writing into fresh output tensors; returns (array, outputs to compare)."""
        Y = [torch.full((M, h), float("nan"), device="cuda") for _ in range(nl)]
        Y2 = [torch.full((M, h), float("nan"), device="cuda") for _ in range(nl)]
        Q = [q.clone() for q in Q0]
        cs = torch.full((ntile, h), float("nan"), device="cuda")
        arr = (L.LinearArgs * nl)()
        for l in range(nl):
            x = X if l == 0 else Y[l - 1]
            if kind == "act":
                a = _make_args(M, h, x, wps[l], act=act, bias=bias[l], Y=Y[l])
            elif kind in ("act_energy", "act_energy_noseed"):
                kw = dict(act=act, Y=Y[l])
                if l == 0:
                    kw.update(rowbias=rowbias, rows_per_group=rpg, rowscale=sigma, rowscale_w=wsig)
                else:
                    kw.update(bias=bias[l])
                if l == nl - 1 and kind == "act_energy":
                    kw.update(Y2=Y2[l], R=wfc)
                a = _make_args(M, h, x, wps[l], **kw)
            elif kind == "dact":
                a = _make_args(M, h, x, wps[l], act=act, S=S[l], Y=Y[l])
            elif kind == "dact_q":
                Y[l] = Q[l]                                     # in place, as the backward pass runs it (qhat_l overwrites qbar_l)
                a = _make_args(M, h, x, wps[l], act=act, S=S[l], Q=Q[l], Y=Q[l])
            else:
                kw = dict(act=act, S=S[l], R=R[l], Y=Y[l], Y2=Y2[l])
                if l == nl - 1:
                    kw.update(colsum=cs)
                a = _make_args(M, h, x, wps[l], **kw)
            arr[l] = a
        return arr, Y, Y2, cs

    lib = L.lib()
    arr, Ya, Y2a, csa = build("per layer")
    for l in range(nl):
        L.check(lib.ardae_linear(ctypes.byref(arr[l]), epi, L.stream_ptr()), "ardae_linear")
    arr, Yb, Y2b, csb = build("chain")
    assert lib.ardae_linear_chain_eligible(arr, nl, epi) == 1
    L.check(lib.ardae_linear_chain(arr, nl, epi, L.stream_ptr()), "ardae_linear_chain")
    torch.cuda.synchronize()
    for l in range(nl):
        assert not torch.isnan(Yb[l]).any(), f"layer {l}: rows never written"
        assert torch.equal(Ya[l], Yb[l]), f"Y of layer {l}: max |diff| {float((Ya[l] - Yb[l]).abs().max())}"
        if kind == "chain" or (kind == "act_energy" and l == nl - 1):
            assert torch.equal(Y2a[l], Y2b[l]), f"Y2 of layer {l}"
    if kind == "chain":
        assert torch.equal(csa, csb)


@pytest.mark.parametrize("M", [131072, 80000, 32768])
@pytest.mark.parametrize("kind", ["act", "dact", "dact_q", "chain"])
@pytest.mark.parametrize("nl", [2, 5])
def test_linear_wide_layers_kernel_equals_per_layer_launches(M, kind, nl):
    """linear_wide_layers_kernel (a row-local run of h x h layers LAYER-major in one launch of the weight-stationary kernel: a workgroup
    walks all its row tiles for layer l, then reads back what it wrote for layer l + 1) against nl launches of ardae_linear: bit-identical,
    every tensor of every layer.  131072 rows = config #2's shard (8 tiles per workgroup), 80000 = the shipped recipes' 128 x 625 (1250
    tiles: workgroups with 4 and 5 tiles), 32768 = 2 tiles per workgroup; the backward run in place (Y = Q), the last CHAIN layer with per-tile
    column sums.  Run twice: a second pass over the same buffers must not see stale rows."""
    h, act = 256, 2
    g = torch.Generator(device="cuda").manual_seed(M + nl)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    X = rn(M, h)
    wps = [pack(rn(h, h) / h ** 0.5) for _ in range(nl)]
    S = [torch.nn.functional.softplus(rn(M, h) * 3) for _ in range(nl)]
    R = [rn(M, h) for _ in range(nl)]
    Q0 = [rn(M, h) for _ in range(nl)]
    bias = [rn(h) for _ in range(nl)]
    epi = {"act": L.EPI_ACT, "dact": L.EPI_DACT, "dact_q": L.EPI_DACT, "chain": L.EPI_CHAIN}[kind]
    ntile = M // 64

    def build():
        Y = [torch.full((M, h), float("nan"), device="cuda") for _ in range(nl)]
        Y2 = [torch.full((M, h), float("nan"), device="cuda") for _ in range(nl)]
        Q = [q.clone() for q in Q0]
        cs = torch.full((ntile, h), float("nan"), device="cuda")
        arr = (L.LinearArgs * nl)()
        for l in range(nl):
            x = X if l == 0 else Y[l - 1]
            if kind == "act":
                a = _make_args(M, h, x, wps[l], act=act, bias=bias[l], Y=Y[l])
            elif kind == "dact":
                a = _make_args(M, h, x, wps[l], act=act, S=S[l], Y=Y[l])
            elif kind == "dact_q":
                Y[l] = Q[l]
                a = _make_args(M, h, x, wps[l], act=act, S=S[l], Q=Q[l], Y=Q[l])
            else:
                kw = dict(act=act, S=S[l], R=R[l], Y=Y[l], Y2=Y2[l])
                if l == nl - 1:
                    kw.update(colsum=cs)
                a = _make_args(M, h, x, wps[l], **kw)
            arr[l] = a
        return arr, Y, Y2, cs

    lib = L.lib()
    arr, Ya, Y2a, csa = build()
    for l in range(nl):
        L.check(lib.ardae_linear(ctypes.byref(arr[l]), epi, L.stream_ptr()), "ardae_linear")
    arr, Yb, Y2b, csb = build()
    assert lib.ardae_linear_wide_layers_eligible(arr, nl, epi) == 1
    L.check(lib.ardae_linear_wide_layers(arr, nl, epi, L.stream_ptr()), "ardae_linear_wide_layers")
    torch.cuda.synchronize()
    for l in range(nl):
        assert not torch.isnan(Yb[l]).any(), f"layer {l}: rows never written"
        assert torch.equal(Ya[l], Yb[l]), f"Y of layer {l}: max |diff| {float((Ya[l] - Yb[l]).abs().max())}"
        if kind == "chain":
            assert torch.equal(Y2a[l], Y2b[l]), f"Y2 of layer {l}"
    if kind == "chain":
        assert torch.equal(csa, csb)
    if kind != "dact_q":        # (in place the second pass would start from the first one's results)
        Ykeep = [y.clone() for y in Yb]
        for y in Yb:
            y.fill_(float("nan"))
        L.check(lib.ardae_linear_wide_layers(arr, nl, epi, L.stream_ptr()), "ardae_linear_wide_layers")
        torch.cuda.synchronize()
        for l in range(nl):
            assert torch.equal(Yb[l], Ykeep[l]), f"second pass, layer {l}"
    # what it refuses: a layer that does not read its predecessor's output, mixed operands
    if kind == "dact_q" and nl == 2:
        arr2, _, _, _ = build()
        arr2[1].Q = None
        assert lib.ardae_linear_wide_layers_eligible(arr2, nl, epi) == 0
        arr3, _, _, _ = build()
        arr3[1].src[0].x = X.data_ptr()
        assert lib.ardae_linear_wide_layers_eligible(arr3, nl, epi) == 0


def test_linear_chain_refuses_what_it_cannot_run():
    h, M = 256, 16384
    X = torch.randn(M, h, device="cuda"); W = pack(torch.randn(h, h, device="cuda"))
    Y = [torch.empty(M, h, device="cuda") for _ in range(2)]
    arr = (L.LinearArgs * 2)()
    arr[0] = _make_args(M, h, X, W, act=2, Y=Y[0])
    arr[1] = _make_args(M, h, X, W, act=2, Y=Y[1])               # second layer does not read the first one's output
    assert L.lib().ardae_linear_chain_eligible(arr, 2, L.EPI_ACT) == 0
    with pytest.raises(ValueError):
        L.check(L.lib().ardae_linear_chain(arr, 2, L.EPI_ACT, L.stream_ptr()))
    arr[1] = _make_args(M, h, Y[0], W, act=2, Y=Y[1])
    assert L.lib().ardae_linear_chain_eligible(arr, 2, L.EPI_ACT) == 1
    assert L.lib().ardae_linear_chain_eligible(arr, 1, L.EPI_ACT) == 0        # a chain has at least two layers
    arr[0] = _make_args(4096, h, X, W, act=2, Y=Y[0]); arr[1] = _make_args(4096, h, Y[0], W, act=2, Y=Y[1])
    assert L.lib().ardae_linear_chain_eligible(arr, 2, L.EPI_ACT) == 0        # 64 tiles: the small-M tiling fills the chip better
