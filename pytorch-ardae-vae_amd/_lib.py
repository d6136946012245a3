"""ctypes binding of libardae_hip.so (C ABI: include/ardae_hip.h).

The HIP library is the product: there is no CPU or eager-PyTorch fallback.  Importing this module
without a built library raises; calling into it without a GPU raises from the HIP runtime.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARDAE_LIB") or os.path.join(_HERE, "libardae_hip.so")   # ARDAE_LIB: experiment builds

c_float_p = ctypes.POINTER(ctypes.c_float)


def debug_knob(name, default=None):
    """Test / experiment switches (ARDAE_GRAPH, ARDAE_OVERLAP, ... and the kernel-selection switches of the library) are honoured
    only together with ARDAE_DEBUG_KNOBS=1: a stray ARDAE_* variable cannot change what a production process runs."""
    if os.environ.get("ARDAE_DEBUG_KNOBS") != "1":
        return default
    return os.environ.get(name, default)


class LinSrc(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("ld", ctypes.c_int), ("K", ctypes.c_int), ("wp", ctypes.c_void_p)]


class LinearArgs(ctypes.Structure):
    _fields_ = [
        ("M", ctypes.c_int), ("Nout", ctypes.c_int),
        ("nsrc", ctypes.c_int),
        ("src", LinSrc * 2),
        ("act", ctypes.c_int),
        ("bias", ctypes.c_void_p),
        ("rowbias", ctypes.c_void_p), ("rowbias_ld", ctypes.c_int), ("rows_per_group", ctypes.c_int),
        ("rowscale", ctypes.c_void_p), ("rowscale_w", ctypes.c_void_p),
        ("S", ctypes.c_void_p), ("ldS", ctypes.c_int),
        ("R", ctypes.c_void_p), ("ldR", ctypes.c_int),
        ("Q", ctypes.c_void_p), ("ldQ", ctypes.c_int),
        ("sigma", ctypes.c_void_p),
        ("eps", ctypes.c_void_p), ("ldeps", ctypes.c_int),
        ("scale", ctypes.c_float),
        ("Y", ctypes.c_void_p), ("ldY", ctypes.c_int),
        ("Y2", ctypes.c_void_p), ("ldY2", ctypes.c_int),
        ("colsum", ctypes.c_void_p),
        ("tile_loss", ctypes.c_void_p),
    ]


class WgradProblem(ctypes.Structure):
    _fields_ = [
        ("M", ctypes.c_int), ("O", ctypes.c_int), ("I", ctypes.c_int), ("npairs", ctypes.c_int),
        ("G", ctypes.c_void_p * 2), ("ldG", ctypes.c_int * 2),
        ("X", ctypes.c_void_p * 2), ("ldX", ctypes.c_int * 2),
        ("bias_pair", ctypes.c_int),
        ("rowscale", ctypes.c_void_p),
        ("splits", ctypes.c_int),
        ("partial", ctypes.c_void_p),
        ("partial_vec", ctypes.c_void_p),
        ("out", ctypes.c_void_p), ("ldout", ctypes.c_int),
        ("out_bias", ctypes.c_void_p),
        ("out_rowscale", ctypes.c_void_p), ("ld_rowscale", ctypes.c_int),
        ("beta", ctypes.c_float),
    ]


class CdaeDesc(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("input_dim", ctypes.c_int), ("context_dim", ctypes.c_int),
                ("h_dim", ctypes.c_int), ("n_layers", ctypes.c_int), ("act", ctypes.c_int)]


class ProfileEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 96), ("calls", ctypes.c_int), ("total_ms", ctypes.c_double), ("flops", ctypes.c_double),
                ("bytes", ctypes.c_double)]


class ModelDesc(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("input_dim", ctypes.c_int), ("noise_dim", ctypes.c_int), ("h_dim", ctypes.c_int),
                ("z_dim", ctypes.c_int), ("n_layers", ctypes.c_int), ("act", ctypes.c_int), ("flags", ctypes.c_int)]


MODEL_NO_CENTER = 1      # ardae_model_desc.flags (residual-conv kinds: do_center=False)
MODEL_HEAD_SHIFT = 1     # kind 5: sampler-head type in flags bits 1-3 (layout.RESCONV_HEADS)
MODEL_CLIPPED = 16       # kind 6: MNISTResConvAuxIPVAEClipped (no 'spm4' clip, z0 keeps an unscaled eps0)
# kinds 3 / 7: NormalDistribution.clip_logvar of the z0 / z heads (models/reparam.py:17-41; flags bits 8-11 / 12-15)
LOGVAR_CLIP = {None: 0, "none": 0, "hard": 1, "softplus": 2, "spm10": 3, "spm6": 4, "spm5": 5, "spm4": 6, "spm3": 7, "spm2": 8, "tanh": 9, "2tanh": 10}
MODEL_CLIP_Z0_SHIFT, MODEL_CLIP_Z_SHIFT = 8, 12


# utils/models.py:14-32 (get_nonlinear_func): all seven names; 'csoftplus' = log(exp(x) + 1) is softplus (evaluated in its accurate form)
ACT = {"none": 0, None: 0, "relu": 1, "softplus": 2, "csoftplus": 2, "elu": 3, "tanh": 4, "leaky_relu": 5, "swish": 6}
LOG_RECORD_FLOATS = 16
EPI_ACT, EPI_DACT, EPI_CHAIN, EPI_DAE_LOSS = 0, 1, 2, 3

# every symbol include/ardae_hip.h declares (checked by tests/test_abi.py)
EXPORTS = {
    "ardae_last_error": (ctypes.c_char_p, []),
    "ardae_abi_version": (ctypes.c_int, []),
    "ardae_packed_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "ardae_linear_row_tiles": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "ardae_linear_col_panels": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "ardae_pack_weight": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_linear": (ctypes.c_int, [ctypes.POINTER(LinearArgs), ctypes.c_int, ctypes.c_void_p]),
    "ardae_linear_chain_eligible": (ctypes.c_int, [ctypes.POINTER(LinearArgs), ctypes.c_int, ctypes.c_int]),
    "ardae_linear_chain": (ctypes.c_int, [ctypes.POINTER(LinearArgs), ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "ardae_linear_wide_layers_eligible": (ctypes.c_int, [ctypes.POINTER(LinearArgs), ctypes.c_int, ctypes.c_int]),
    "ardae_linear_wide_layers": (ctypes.c_int, [ctypes.POINTER(LinearArgs), ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "ardae_wgrad_splits": (ctypes.c_int, [ctypes.c_int] * 4),
    "ardae_wgrad_batch": (ctypes.c_int, [ctypes.POINTER(WgradProblem), ctypes.c_int, ctypes.c_void_p]),
    "ardae_latent_perturb": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_float] * 2 + [ctypes.c_void_p] * 4),
    "ardae_latent_perturb_nstd": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_float] * 2 + [ctypes.c_void_p] * 4),
    "ardae_latent_perturb_draw_ok": (ctypes.c_int, [ctypes.c_int] * 3),
    "ardae_latent_perturb_draw": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_float] * 2 + [ctypes.c_uint64] * 3 +
                                  [ctypes.c_void_p, ctypes.c_uint64] + [ctypes.c_void_p] * 5),
    "ardae_cdae_perturb_fused_ok": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "ardae_cdae_perturb_loss_grads": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 2 + [ctypes.c_float] * 2 + [ctypes.c_uint64] * 3 +
                                      [ctypes.c_void_p, ctypes.c_uint64] + [ctypes.c_void_p] * 5 + [ctypes.c_size_t] + [ctypes.c_void_p] * 3),
    "ardae_center_scale": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_float] + [ctypes.c_void_p] * 2),
    "ardae_philox_normal": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]),
    "ardae_philox_uniform": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]),
    "ardae_bernoulli": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64,
                                       ctypes.c_uint64, ctypes.c_void_p]),
    "ardae_step_state_advance": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]),
    "ardae_philox_normal_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]),
    "ardae_philox_normal_at": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                               ctypes.c_void_p]),
    "ardae_adam_ref_step_dev": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int64] + [ctypes.c_double] * 3 + [ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_adam_ref_step": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int64] + [ctypes.c_double] * 4 + [ctypes.c_int, ctypes.c_void_p]),
    "ardae_rmsprop_step": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_double] * 4 + [ctypes.c_void_p]),
    "ardae_sgd_step": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int64, ctypes.c_double, ctypes.c_void_p]),
    "ardae_cdae_param_floats": (ctypes.c_size_t, [ctypes.POINTER(CdaeDesc)]),
    "ardae_cdae_packed_floats": (ctypes.c_size_t, [ctypes.POINTER(CdaeDesc)]),
    "ardae_cdae_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(CdaeDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "ardae_cdae_pack": (ctypes.c_int, [ctypes.POINTER(CdaeDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_cdae_loss_grads": (ctypes.c_int, [ctypes.POINTER(CdaeDesc)] + [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_void_p] * 4),
    "ardae_model_param_floats": (ctypes.c_size_t, [ctypes.POINTER(ModelDesc)]),
    "ardae_model_packed_floats": (ctypes.c_size_t, [ctypes.POINTER(ModelDesc)]),
    "ardae_model_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(ModelDesc), ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "ardae_model_pack": (ctypes.c_int, [ctypes.POINTER(ModelDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_model_encode": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_model_encode_pair": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "ardae_model_encode_hidden": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_model_encode_hidden_raw": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_model_decode": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_model_loss_rows": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_model_vae_forward": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int,
                                                ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p]),
    "ardae_model_vae_backward": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                                 ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]),
    "ardae_model_vae_backward_decoder": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int,
                                                         ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "ardae_model_vae_backward_sampler": (ctypes.c_int, [ctypes.POINTER(ModelDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int,
                                                         ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t,
                                                         ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]),
    "ardae_relaxed_bernoulli": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p]),
    "ardae_gaussian_sample": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_cholesky_batched": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_log_scalars": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int, ctypes.c_void_p]),
    "ardae_gather_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "ardae_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "ardae_debug_stamp": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "ardae_profile_report": (ctypes.c_int, [ctypes.POINTER(ProfileEntry), ctypes.c_int]),
    "ardae_dp_backend": (ctypes.c_char_p, []),
    "ardae_dp_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "ardae_dp_comm_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]),
    "ardae_dp_comm_query": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3),
    "ardae_dp_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "ardae_dp_allreduce_mean": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "ardae_cdae_score": (ctypes.c_int, [ctypes.POINTER(CdaeDesc)] + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_size_t] + [ctypes.c_void_p] * 2),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what="ardae call"):
    if rc != 0:
        msg = lib().ardae_last_error().decode("utf-8", "replace")
        if rc < 0:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what}: HIP error {rc}: {msg}")


def ptr(t):
    """Device pointer of an fp32 CUDA(HIP) tensor (or None).  Layout is the caller's business (strided views are passed with their
    leading dimension); the engine validates its batches in ArdaeEngine._check_batch."""
    if t is None:
        return None
    import torch
    if not (t.is_cuda and t.dtype == torch.float32):
        raise TypeError(f"expected a float32 tensor on the GPU, got {t.dtype} on {t.device}")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def profile_report(max_entries=64):
    """Per-kernel (name, calls, total_ms, flops, bytes) since ardae_profile_enable(1); synchronises and clears the log."""
    buf = (ProfileEntry * max_entries)()
    n = lib().ardae_profile_report(buf, max_entries)
    return [dict(name=buf[i].name.decode(), calls=buf[i].calls, total_ms=buf[i].total_ms, flops=buf[i].flops, bytes=buf[i].bytes)
            for i in range(min(n, max_entries))]
