"""The C-ABI library loads and exports every symbol include/ardae_hip.h declares (no compute: runs without a GPU)."""
import ctypes
import os
import re

import pytest

import ardae_amd
from ardae_amd import _lib as L
from ardae_amd import layout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ardae_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ardae_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 30
    h = ctypes.CDLL(L.LIB_PATH)
    for s in syms:
        assert hasattr(h, s), f"{s} declared in include/ardae_hip.h but not exported by libardae_hip.so"
        assert s in L.EXPORTS, f"{s} has no ctypes binding in _lib.EXPORTS"
    assert sorted(L.EXPORTS) == syms


def test_abi_version_and_error_channel():
    lib = L.lib()
    assert lib.ardae_abi_version() == 1
    a = L.LinearArgs()
    assert lib.ardae_linear(ctypes.byref(a), 0, None) < 0          # argument validation happens before any HIP call
    assert b"empty problem" in lib.ardae_last_error()
    with pytest.raises(ValueError):
        L.check(lib.ardae_pack_weight(None, 0, 0, 0, 0, None, None))


@pytest.mark.parametrize("kind,args", [("mnist", (784, 100, 256, 32, 2)), ("toy", (2, 10, 256, 2, 2)), ("mnist", (24, 10, 64, 8, 2)),
                                       ("conv", (784, 100, 800, 32, 1)), ("auxmnist", (784, 100, 300, 32, 2)), ("auxmnist", (24, 10, 48, 8, 3)), ("auxconv", (784, 100, 800, 32, 1)),
                                       ("resconv", (784, 100, 512, 32, 1)), ("auxresconv", (784, 100, 450, 32, 1))])
def test_model_layout_matches_c_side(kind, args):
    spec = layout.model_spec(kind, *args)
    _, total = layout.offsets(spec)
    d = L.ModelDesc({"mnist": 0, "toy": 1, "conv": 2, "auxmnist": 3, "auxconv": 4, "resconv": 5, "auxresconv": 6}[kind], *args,
                    3 if "resconv" in kind else 2)
    if kind == "resconv":       # every sampler head of ivae_ardae.py's --model resconv* choices, one to three hidden layers
        for et, code in layout.RESCONV_HEADS.items():
            for nl in (1, 2, 3):
                _, tot = layout.offsets(layout.model_spec(kind, *args[:4], nl, enc_type=et))
                dd = L.ModelDesc(5, *args[:4], nl, 3, code << L.MODEL_HEAD_SHIFT)
                assert L.lib().ardae_model_param_floats(ctypes.byref(dd)) == tot, (et, nl)
                assert L.lib().ardae_model_workspace_floats(ctypes.byref(dd), 4, 8, 1) > 0
    assert L.lib().ardae_model_param_floats(ctypes.byref(d)) == total
    assert L.lib().ardae_model_packed_floats(ctypes.byref(d)) > total
    assert L.lib().ardae_model_workspace_floats(ctypes.byref(d), 8, 16, 1) > 0


@pytest.mark.parametrize("kind,args", [("grad", (32, 32, 256, 3)), ("res", (32, 32, 1024, 6)), ("grad", (2, 2, 256, 3)), ("grad", (32, 32, 512, 4))])
def test_cdae_layout_matches_c_side(kind, args):
    spec = layout.cdae_spec(kind, *args)
    _, total = layout.offsets(spec)
    d = L.CdaeDesc(0 if kind == "grad" else 1, *args, 2)
    assert L.lib().ardae_cdae_param_floats(ctypes.byref(d)) == total
    assert L.lib().ardae_cdae_workspace_floats(ctypes.byref(d), 4, 8, 1) > L.lib().ardae_cdae_workspace_floats(ctypes.byref(d), 4, 8, 0) > 0


def test_survey_parameter_counts():
    """839 472 / 543 489 (cfg #2), 271 386 / 528 129 (cfg #1), 2 923 521 (cfg #4 cDAE), 17 943 584 (cfg #5 cDAE)."""
    tot = lambda spec: layout.offsets(spec)[1]
    assert tot(layout.model_spec("mnist", 784, 100, 256, 32, 2)) == 839472
    assert tot(layout.cdae_spec("grad", 32, 32, 256, 3)) == 543489
    assert tot(layout.model_spec("toy", 2, 10, 256, 2, 2)) == 271386
    assert tot(layout.cdae_spec("grad", 2, 2, 256, 3)) == 528129
    assert tot(layout.cdae_spec("grad", 32, 32, 512, 4)) == 2923521
    assert tot(layout.model_spec("conv", 784, 100, 800, 32, 1)) == 757773
    assert tot(layout.cdae_spec("res", 32, 32, 1024, 6)) == 17943584
    # the reference's own parameter counts of the two residual-conv models (oracle/gen_golden.py asserts names and shapes against them)
    assert tot(layout.model_spec("resconv", 784, 100, 512, 32, 1)) == 2961487
    assert tot(layout.model_spec("auxresconv", 784, 100, 450, 32, 1)) == 2151637


def test_modules_refuse_cpu_execution():
    """No silent fallback: the HIP path is the only path."""
    import torch
    m = ardae_amd.MNISTIPVAE(input_dim=24, noise_dim=10, h_dim=64, num_hidden_layers=2, nonlinearity="softplus", enc_type="concat", z_dim=8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.encode(torch.zeros(2, 24), std=0)
    c = ardae_amd.MLPGradCARDAE(input_dim=8, context_dim=8, std=1., h_dim=64, num_hidden_layers=3, nonlinearity="softplus")
    with pytest.raises(RuntimeError, match="no CPU path"):
        c(torch.zeros(2, 3, 8), torch.zeros(2, 1, 8))
    with pytest.raises(NotImplementedError):
        ardae_amd.MLPGradCARDAE(input_dim=8, context_dim=8, h_dim=64, num_hidden_layers=3, nonlinearity="softplus", noise_type="laplace")
    with pytest.raises(NotImplementedError):
        ardae_amd.ToyIPVAE(enc_type="scale")


def test_state_dict_roundtrip_keeps_flat_views():
    import torch
    m = ardae_amd.MNISTIPVAE(input_dim=24, noise_dim=10, h_dim=64, num_hidden_layers=2, nonlinearity="softplus", enc_type="concat", z_dim=8)
    sd = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    off = 0
    for n, p in m.named_parameters():
        assert p.data_ptr() == m.flat_params().data_ptr() + 4 * off      # still views of the one flat buffer
        assert torch.equal(p, sd[n])
        off += p.numel()
    m2 = m.double().float()                                               # _apply keeps the views linked
    assert all(p.data_ptr() >= m2.flat_params().data_ptr() for p in m2.parameters())


def test_argument_validation_of_the_step_entry_points():
    """Error behaviour of the entry points a caller can misuse (all checks run before any HIP call, so no GPU is needed):
    negative status + a message on the error channel; nothing is launched, nothing is written."""
    lib = L.lib()
    md = L.ModelDesc(0, 24, 10, 64, 8, 2, 2)
    conv = L.ModelDesc(2, 784, 100, 800, 32, 1, 2)
    cd = L.CdaeDesc(0, 8, 8, 64, 3, 2)
    one = ctypes.c_void_p(64)                     # any non-null address: validation must fail before it is dereferenced
    big = ctypes.c_size_t(1 << 40)

    def fails(rc, fragment):
        assert rc < 0
        assert fragment.encode() in lib.ardae_last_error(), lib.ardae_last_error()

    # empty / oversized batches
    fails(lib.ardae_cdae_loss_grads(ctypes.byref(cd), one, one, one, one, one, one, 0, 8, one, big, one, one, None, None), "bad batch")
    fails(lib.ardae_cdae_loss_grads(ctypes.byref(cd), one, one, one, one, one, one, 1 << 20, 1 << 12, one, big, one, one, None, None), "bad batch")
    # workspace smaller than ardae_*_workspace_floats asks for
    fails(lib.ardae_cdae_loss_grads(ctypes.byref(cd), one, one, one, one, one, one, 4, 8, one, ctypes.c_size_t(16), one, one, None, None),
          "workspace too small")
    fails(lib.ardae_model_encode_pair(ctypes.byref(md), one, one, one, one, 4, 8, one, ctypes.c_size_t(16), one, one, 0, None), "workspace too small")
    # encode_pair: phase out of range, missing noise for the phases that need it
    fails(lib.ardae_model_encode_pair(ctypes.byref(md), one, one, one, one, 4, 8, one, big, one, one, 7, None), "phase")
    fails(lib.ardae_model_encode_pair(ctypes.byref(md), one, one, one, None, 4, 8, one, big, one, one, 2, None), "null pointer")
    # the split backward exists for the MLP models only
    fails(lib.ardae_model_vae_backward_decoder(ctypes.byref(conv), one, one, one, one, 4, 1, 1.0, 1.0, one, big, None), "conv model")
    fails(lib.ardae_model_vae_backward_sampler(ctypes.byref(conv), one, one, one, one, 4, 1, one, 1.0, one, big, one, 0.0, None), "conv model")
    # unknown network kinds / activations (0 = none and anything beyond swish)
    for bad_act in (0, 7):
        bad = L.CdaeDesc(0, 8, 8, 64, 3, bad_act)
        assert lib.ardae_cdae_workspace_floats(ctypes.byref(bad), 4, 8, 1) == 0
        fails(lib.ardae_cdae_pack(ctypes.byref(bad), one, one, None), "unknown activation")
    assert lib.ardae_cdae_workspace_floats(ctypes.byref(L.CdaeDesc(0, 8, 8, 64, 3, 1)), 4, 8, 1) > 0      # relu in mlp-grad: --cdae-nonlin's default
    assert lib.ardae_cdae_param_floats(ctypes.byref(L.CdaeDesc(5, 8, 8, 64, 3, 2))) == 0
    # weight-gradient batches: problem count and shapes
    probs = (L.WgradProblem * 1)()
    fails(lib.ardae_wgrad_batch(probs, 0, None), "problems per batch")
    fails(lib.ardae_wgrad_batch(probs, 1, None), "empty problem")


def test_engine_batch_validation_rules():
    """what ArdaeEngine.step / the phase entry points accept as a batch (checked before any pointer reaches a kernel).
    The rules are host logic, so they are exercised here on CPU tensors against a stand-in for the engine's fields."""
    import types
    import torch
    from ardae_amd.engine import ArdaeEngine
    eng = types.SimpleNamespace(B=4, dev=torch.device("cuda", 0), model=types.SimpleNamespace(input_dim=24))
    check = lambda x: ArdaeEngine._check_batch(eng, x, "step")
    good = torch.zeros(4, 24)
    with pytest.raises(ValueError, match="on cuda:0"):
        check(good)                                               # everything right but the device
    with pytest.raises(ValueError, match="batch_size=4"):
        check(torch.zeros(3, 24))                                 # ragged last batch of a loader without drop_last
    with pytest.raises(ValueError, match="batch_size=4"):
        check(torch.zeros(4, 25))
    with pytest.raises(ValueError, match="batch_size=4"):
        check(torch.zeros(1, 24).expand(4, 24)[:1])
    with pytest.raises(ValueError, match="contiguous"):
        check(torch.zeros(4, 48)[:, ::2])                         # strided view
    with pytest.raises(ValueError, match="contiguous"):
        check(torch.zeros(24, 4).t())
    with pytest.raises(ValueError, match="float32"):
        check(good.double())
    with pytest.raises(TypeError):
        check([[0.0] * 24] * 4)
    with pytest.raises(ValueError, match="on cuda:0"):
        check(torch.zeros(4, 1, 4, 6))                            # image-shaped batches pass the shape rule


def test_host_and_in_step_philox_offsets_are_disjoint():
    """host-side draws (rng.normal / uniform, data.*, model.generate / logprob) and the engine's in-step draws share one
    seed; their Philox offsets must never coincide.  In-step offsets are RNG_STRIDE * step + k < 2^63, host offsets have bit 63 set."""
    from ardae_amd import rng
    from ardae_amd.engine import ArdaeEngine
    rng.manual_seed(7, 0)
    host = [rng._next_offset() for _ in range(40)]
    assert all(o & rng.HOST_STREAM for o in host) and len(set(host)) == 40
    assert [o & ~rng.HOST_STREAM for o in host] == list(range(40))
    in_step = {ArdaeEngine.RNG_STRIDE * step + k for step in range(1, 10_000, 97) for k in range(ArdaeEngine.RNG_STRIDE)}
    assert max(in_step) < rng.HOST_STREAM and not (in_step & set(host))
    assert rng.get_state() == {"seed": 7, "offset": 40}           # the host counter itself (what the engine checkpoint stores)
    rng.manual_seed(7, 40)
    assert rng._next_offset() == rng.HOST_STREAM | 40


def test_activation_names_follow_get_nonlinear_func():
    """utils/models.py:14-32: relu, elu, tanh, softplus, csoftplus (= softplus; evaluated in the accurate form), leaky_relu, swish - every
    name get_nonlinear_func knows constructs; anything else is refused by the host classes."""
    import torch
    import ardae_amd as net
    assert L.ACT["csoftplus"] == L.ACT["softplus"] == 2 and L.ACT["relu"] == 1
    assert {L.ACT[k] for k in ("elu", "tanh", "leaky_relu", "swish")} == {3, 4, 5, 6}
    x = torch.linspace(-30, 30, 2001, dtype=torch.float64)
    from oracle import ardae_oracle as O
    # log(exp(x) + 1) is softplus; its literal fp32 evaluation only loses the tail below ~1e-7 (1 + e^x rounds to 1)
    assert torch.allclose(O.act("csoftplus")(x.float()), O.act("softplus")(x.float()), rtol=1e-6, atol=2e-7)
    net.MLPGradCARDAE(input_dim=2, context_dim=2, h_dim=16, num_hidden_layers=1, nonlinearity="swish")
    net.ToyIPVAE(input_dim=2, noise_dim=2, h_dim=16, z_dim=2, nonlinearity="swish", enc_type="concat")
    with pytest.raises(NotImplementedError):
        net.MLPGradCARDAE(input_dim=2, context_dim=2, h_dim=16, num_hidden_layers=1, nonlinearity="gelu")
    with pytest.raises(NotImplementedError):
        net.ToyIPVAE(input_dim=2, noise_dim=2, h_dim=16, z_dim=2, nonlinearity="gelu", enc_type="concat")
    # the reference's class defaults (tanh) construct
    net.MLPGradCARDAE(input_dim=2, context_dim=2, h_dim=16, num_hidden_layers=1)
    net.ToyIPVAE(input_dim=2, noise_dim=2, h_dim=16, z_dim=2, enc_type="concat")


def test_three_piece_cut_is_exact():
    """csrc/wgrad_x9.hip::cut3, restated in numpy: h = the top 16 bits of x, m = the top 16 bits of x - h, l = x - h - m.  Both subtractions are
    exact and h + m + l == x bit for bit for EVERY finite fp32 value (a million random bit patterns plus edge cases: all 24 significand bits
    set, powers of two, tiny and huge magnitudes, zeros).  For |x| >= 2^-100 the remainder l has at most 8 significand bits - its low half is
    zero by itself, i.e. it is a bf16 value as it stands and nothing is lost when the kernel keeps only its high half; below that (pieces in
    the denormal range, |x| < ~1e-30) the kernel's l loses its low half when it is a denormal: less than 2^-133 in absolute terms, far below an
    fp32 underflow."""
    import numpy as np
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 1 << 32, size=1 << 20, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x)]
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.9999999, -1.9999999, 3.4e38, -3.4e38, 1.2e-38, 1e-30, 16777215.0, 0.1, 1.0 / 3.0], dtype=np.float32)
    x = np.concatenate([x, edge])
    hi = lambda v: (v.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    h = hi(x)
    r1 = (x - h).astype(np.float32)
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))          # exact
    m = hi(r1)
    l = (r1 - m).astype(np.float32)
    assert np.array_equal(l.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))          # exact
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    normal = np.abs(x) >= np.float32(2.0 ** -100)
    assert np.all((l[normal].view(np.uint32) & np.uint32(0xffff)) == 0)                                 # l is a bf16 value as it stands
    lost = np.abs(l.astype(np.float64) - hi(l).astype(np.float64))                                     # what keeping only l's high half drops
    assert np.all(lost[normal] == 0) and np.all(lost < 2.0 ** -133)
