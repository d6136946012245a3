"""Host-side mirror of the reference's model / cDAE object surface, backed by the HIP library.

    net.MNISTIPVAE / net.ToyIPVAE        <- models/ivae/mnist.py:201-301, models/ivae/toy.py:739-873 (enc_type='concat')
    net.MLPGradCARDAE / net.MLPResCARDAE <- models/graddae/mlp.py:341-483, models/resdae/mlp.py:286-413

Same constructor kwargs, same `state_dict()` keys and `[out, in]` layouts (reference checkpoints load), same method
names and argument meaning, same exception types.  Parameters are `nn.Parameter` views into ONE flat fp32 buffer
(the layout the C ABI consumes); forward/backward are `torch.autograd.Function`s that call the ABI.  There is no
PyTorch fallback: on a CPU tensor or without the built library these modules raise.
"""
import ctypes
import math
import weakref

import torch
import torch.nn as nn

from . import _lib as L
from . import layout
from . import rng


class _Box(nn.Module):
    """Name-only container (gives parameters their dotted reference names)."""


class _EncodeBox(_Box):
    """`model.encode(x, std=0)` -> [B, nz, z] (Encoder.forward, ivae/mnist.py:102-121).  Holds the `encode.*` parameters by
    name; the computation is the owning model's sampler (the owner is kept out of the module tree on purpose)."""

    def forward(self, x, noise=None, std=None, nz=1):
        o = self._owner_ref()
        return o._sample(o._x(x), nz, std, noise)

    def forward_hidden(self, x, std=None, nz=1):
        """Aux models: `model.encode.forward_hidden(x, std=0)` -> cat(h0, h) [B, 2 h], the `hidden1a` cDAE context
        (ivae/auxmnist.py:125-132, ivae_ardae.py:737-739)."""
        assert nz == 1                                    # ivae/auxmnist.py:126
        if std is None or float(std) != 0.0:
            raise NotImplementedError("forward_hidden is implemented for std=0 (its only use in ivae_ardae.py)")
        return self._owner_ref()._hidden(x)


class FlatParamModule(nn.Module):
    def _build_params(self, spec, boxes=None):
        boxes = boxes or {}
        self._spec = spec
        self._offs, total = layout.offsets(spec)
        self.register_buffer("_flat", torch.zeros(total), persistent=False)
        for name, (off, n, shape) in self._offs.items():
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, boxes.get(p, _Box)() if mod is self else _Box())
                mod = mod._modules[p]
            par = nn.Parameter(self._flat[off:off + n].view(shape))
            par._ardae_owner = weakref.ref(self)          # net.Adam / net.RMSprop find the owning module through it (optim.py)
            mod.register_parameter(parts[-1], par)
        self._packed = None
        self._tracked = False        # see _pack_is_current
        self._packed_ver = None

    # Weight image policy of the module path.  Default: re-pack at EVERY use (one launch over ~2-12 MB), because tensor version
    # counters cannot be trusted to see an update - `p.data.add_(...)`, which is how the reference's vendored Adam writes
    # (utils/optim.py:106), bumps none - so any optimiser, the reference's included, may be used with these modules.  Once one of
    # THIS package's optimisers (optim.py: they bump the counters) is built over the parameters the module switches to tracking:
    # the image is rebuilt only when a parameter's version has moved (optimiser step, load_state_dict, any in-place op on p).
    # Writing through `p.data` by hand in that mode needs `mark_dirty()`.
    def _pack_is_current(self):
        if self._packed is None or not self._tracked:
            return False
        return self._packed_ver == tuple(p._version for p in self.parameters())

    def _note_packed(self):
        self._packed_ver = tuple(p._version for p in self.parameters())

    def mark_dirty(self):
        """Parameters were written behind autograd's back (`p.data...`) while one of net's optimisers tracks them."""
        self._packed_ver = None

    # keep the parameters views of the flat buffer across .to()/.cuda()/.float()
    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._relink()
        return self

    def _relink(self):
        flat = self._flat
        for name, p in self.named_parameters():
            off, n, shape = self._offs[name]
            p.data = flat[off:off + n].view(shape)
        self._packed = None

    def flat_params(self):
        return self._flat

    def _default_init(self):
        """nn.Linear's default init (kaiming_uniform(a=sqrt 5) == U(+-1/sqrt(fan_in)) for weight and bias)."""
        spec = dict(self._spec)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith(".scale") or name.endswith(".direction") or (name[:-len("bias")] + "direction") in spec:
                    continue                                        # weight-normalised operators: reset_parameters() of the owning model
                wshape = p.shape if name.endswith("weight") else spec[name[:-len("bias")] + "weight"]
                fan_in = wshape[1] * (wshape[2] * wshape[3] if len(wshape) == 4 else 1)   # torch's fan_in for (transposed) convs too
                bound = 1.0 / math.sqrt(fan_in)
                p.uniform_(-bound, bound)

    def _require_gpu(self, *tensors):
        if not self._flat.is_cuda:
            raise RuntimeError(f"{type(self).__name__}: parameters are on {self._flat.device}; the HIP engine needs .to('cuda') "
                               "(there is no CPU path)")
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise RuntimeError(f"{type(self).__name__}: got a {t.device} tensor; inputs must be on the GPU")

    def _ws(self, nfloats):
        return torch.empty(nfloats, device=self._flat.device, dtype=torch.float32)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


KIND_IDS = {"mnist": 0, "toy": 1, "conv": 2, "auxmnist": 3, "auxconv": 4, "resconv": 5, "auxresconv": 6, "auxtoy": 7}     # ardae_model_desc.kind
AUX_KINDS = ("auxmnist", "auxconv", "auxresconv", "auxtoy")                                                 # hierarchical samplers
GAUSSIAN_DECODERS = ("toy", "auxtoy")                                                                        # decode.reparam.{mean_fn, logvar_fn}


# ---------------------------------------------------------------------------------------------------------------
# conditional AR-DAE
# ---------------------------------------------------------------------------------------------------------------
class _CdaeLossFn(torch.autograd.Function):
    """forward = ConditionalARDAE.forward's loss; the double backward runs in the same ABI call, so backward() only scales."""

    @staticmethod
    def forward(ctx, mod, xbar, sigma, eps, context, B, S, *params):
        lib = L.lib()
        d = mod._desc
        ws = mod._ws(lib.ardae_cdae_workspace_floats(ctypes.byref(d), B, S, 1))
        loss = torch.empty(1, device=xbar.device)
        grads = torch.zeros_like(mod._flat)
        L.check(lib.ardae_cdae_loss_grads(ctypes.byref(d), L.ptr(mod._flat), L.ptr(mod._packed_weights()), L.ptr(xbar), L.ptr(sigma),
                                          L.ptr(eps), L.ptr(context), B, S, L.ptr(ws), ws.numel(), L.ptr(loss), L.ptr(grads), None,
                                          L.stream_ptr()), "ardae_cdae_loss_grads")
        ctx.mod, ctx.grads = mod, grads
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        mod = ctx.mod
        out = []
        for name, p in mod.named_parameters():
            if name in mod._no_grad_names:
                out.append(None)      # the reference leaves .grad = None here
                continue
            off, n, shape = mod._offs[name]
            out.append(ctx.grads[off:off + n].view(shape) * gloss)
        return (None,) * 7 + tuple(out)


class ConditionalARDAE(FlatParamModule):
    _kind = None

    def __init__(self, input_dim=2, h_dim=128, context_dim=2, std=0.01, num_hidden_layers=1, nonlinearity="tanh",
                 noise_type="gaussian", enc_input=True, enc_ctx=True, std_method="default"):
        super().__init__()
        if noise_type != "gaussian":
            raise NotImplementedError            # reference: graddae/mlp.py:392-393 for unknown types; only gaussian is on the path
        if not (enc_input and enc_ctx):
            raise NotImplementedError("enc_input=enc_ctx=True is the only configuration ivae_ardae.py:583-606 constructs")
        if nonlinearity not in L.ACT or nonlinearity in ("none", None):
            raise NotImplementedError(f"nonlinearity {nonlinearity!r}: get_nonlinear_func (utils/models.py:14-32) knows relu, softplus / csoftplus, elu, tanh, leaky_relu and swish")
        self.input_dim, self.h_dim, self.context_dim, self.std = input_dim, h_dim, context_dim, std
        self.num_hidden_layers, self.nonlinearity, self.noise_type = num_hidden_layers, nonlinearity, noise_type
        self.enc_input, self.enc_ctx = enc_input, enc_ctx
        self._desc = L.CdaeDesc(0 if self._kind == "grad" else 1, input_dim, context_dim, h_dim, num_hidden_layers, L.ACT[nonlinearity])
        self._build_params(layout.cdae_spec(self._kind, input_dim, context_dim, h_dim, num_hidden_layers))
        self._no_grad_names = {"neglogprob.fc.bias"} if self._kind == "grad" else set()
        self._default_init()

    def _packed_weights(self):
        """MFMA-lane-linear image of the current weights (policy: FlatParamModule._pack_is_current; the fused engine keeps its own
        image and re-packs right after its own optimiser kernels)."""
        if self._pack_is_current():
            return self._packed
        lib = L.lib()
        if self._packed is None:
            self._packed = self._ws(lib.ardae_cdae_packed_floats(ctypes.byref(self._desc)))
        L.check(lib.ardae_cdae_pack(ctypes.byref(self._desc), L.ptr(self._flat), L.ptr(self._packed), L.stream_ptr()), "ardae_cdae_pack")
        self._note_packed()
        return self._packed

    def _prep(self, input, context, std):
        assert input.dim() == 3      # bsz x ssz x x_dim   (graddae/mlp.py:402)
        assert context.dim() == 3    # bsz x 1 x ctx_dim   (graddae/mlp.py:403)
        B, S = input.size(0), input.size(1)
        self._require_gpu(input, context)
        if std is None:
            std = input.new_zeros(B, S, 1)
        else:
            assert torch.is_tensor(std)
        x = _f32c(input).view(B * S, self.input_dim)
        c = _f32c(context).view(B, self.context_dim)
        s = _f32c(std).reshape(B * S)
        return B, S, x, c, s

    def forward(self, input, context, std=None, scale=None, eps=None):
        """-> (None, loss) like the reference.  `scale` is accepted and ignored (graddae/mlp.py:410-411).
        `eps` injects the Gaussian perturbation draw ([B*S, x_dim]); default: the library's Philox stream."""
        B, S, x, c, s = self._prep(input, context, std)
        if eps is None:
            eps = rng.normal((B * S, self.input_dim), x.device)
        eps = _f32c(eps).view(B * S, self.input_dim)
        xbar = torch.addcmul(x, s[:, None], eps)         # add_gaussian_noise (graddae/mlp.py:21-23)
        loss = _CdaeLossFn.apply(self, xbar, s, eps, c, B, S, *self.parameters())
        return None, loss

    def glogprob(self, input, context, std=None, scale=None):
        B, S, x, c, s = self._prep(input, context, std)
        lib = L.lib()
        ws = self._ws(lib.ardae_cdae_workspace_floats(ctypes.byref(self._desc), B, S, 0))
        out = torch.empty(B * S, self.input_dim, device=x.device)
        L.check(lib.ardae_cdae_score(ctypes.byref(self._desc), L.ptr(self._flat), L.ptr(self._packed_weights()), L.ptr(x), L.ptr(s),
                                     L.ptr(c), B, S, L.ptr(ws), ws.numel(), L.ptr(out), L.stream_ptr()), "ardae_cdae_score")
        return out.view(B, S, self.input_dim)


class MLPGradCARDAE(ConditionalARDAE):
    """models/graddae/mlp.py::ConditionalARDAE (`--cdae mlp-grad`)."""
    _kind = "grad"


class MLPResCARDAE(ConditionalARDAE):
    """models/resdae/mlp.py::ConditionalARDAE (`--cdae mlp-res`)."""
    _kind = "res"


# ---------------------------------------------------------------------------------------------------------------
# implicit-posterior VAE
# ---------------------------------------------------------------------------------------------------------------
def normal_energy_func(x, mu=0., logvar=0.):
    """utils/energy.py:74-77 (the only prior energy the HIP engine implements)."""
    x = x.view(x.size(0), -1)
    return torch.sum(0.5 * (logvar + (x - mu) ** 2 / math.exp(logvar) + math.log(2. * math.pi)), dim=1)


class _VaeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, noise, beta, nz, *params):
        lib = L.lib()
        d = mod._desc
        B = x.size(0)
        ws = mod._ws(lib.ardae_model_workspace_floats(ctypes.byref(d), B, nz, 1))
        z = torch.empty(B * nz, mod.z_dim, device=x.device)
        losses = torch.empty(3, device=x.device)
        L.check(lib.ardae_model_vae_forward(ctypes.byref(d), L.ptr(mod._flat), L.ptr(mod._packed_weights()), L.ptr(x), L.ptr(noise), B, nz,
                                            float(beta), L.ptr(ws), ws.numel(), L.ptr(z), L.ptr(losses), L.stream_ptr()),
                "ardae_model_vae_forward")
        ctx.mod, ctx.ws, ctx.x, ctx.noise, ctx.beta, ctx.nz = mod, ws, x, noise, float(beta), nz
        ctx.mark_non_differentiable(losses)
        return z.view(B, nz, mod.z_dim), losses[0].clone(), losses

    @staticmethod
    def backward(ctx, dz, dloss, _dlosses):
        mod, lib = ctx.mod, L.lib()
        B = ctx.x.size(0)
        grads = torch.empty_like(mod._flat)
        dl = float(dloss) if dloss is not None else 0.0
        dzc = _f32c(dz).view(B * ctx.nz, mod.z_dim) if dz is not None else None
        L.check(lib.ardae_model_vae_backward(ctypes.byref(mod._desc), L.ptr(mod._flat), L.ptr(mod._packed_weights()), L.ptr(ctx.x),
                                             L.ptr(ctx.noise), B, ctx.nz, ctx.beta, dl, L.ptr(dzc), L.ptr(ctx.ws), ctx.ws.numel(),
                                             L.ptr(grads), 0.0, L.stream_ptr()), "ardae_model_vae_backward")
        out = []
        for name, _ in mod.named_parameters():
            off, n, shape = mod._offs[name]
            out.append(grads[off:off + n].view(shape))
        return (None,) * 5 + tuple(out)


class ImplicitPosteriorVAE(FlatParamModule):
    _kind = None
    _enc_types = ("concat",)
    return_samples = True        # forward() also returns the decoder sample / mean the reference's visualisation code reads

    def __init__(self, energy_func=normal_energy_func, input_dim=784, noise_dim=100, h_dim=300, z_dim=32, nonlinearity="softplus",
                 num_hidden_layers=1, init="gaussian", enc_type="concat"):
        super().__init__()
        assert enc_type in self._enc_types               # ivae/mnist.py:224; the other toy encoders are out of scope
        if energy_func is not normal_energy_func:
            raise NotImplementedError("only utils.normal_energy_func is implemented on the HIP path")
        if nonlinearity not in (("elu",) if self._kind in ("resconv", "auxresconv") else tuple(k for k in L.ACT if k not in ("none", None))):
            raise NotImplementedError(f"nonlinearity {nonlinearity!r}")
        self.energy_func = energy_func
        self.input_dim, self.noise_dim, self.h_dim, self.z_dim = input_dim, noise_dim, h_dim, z_dim
        self.latent_dim = z_dim
        self.nonlinearity, self.num_hidden_layers, self.init, self.enc_type = nonlinearity, num_hidden_layers, init, enc_type
        flags = L.MODEL_NO_CENTER if self._kind in ("resconv", "auxresconv") and not getattr(self, "do_center", True) else 0
        flags |= getattr(self, "_flag_extra", 0)
        self._desc = L.ModelDesc(KIND_IDS[self._kind], input_dim, noise_dim, h_dim, z_dim, num_hidden_layers, L.ACT[nonlinearity], flags)
        # floats per row of a sampler draw: the aux models take two draws per call, laid out side by side [eps0 | eps]
        self._noise_width = noise_dim + z_dim if self._kind in AUX_KINDS else noise_dim
        # columns of the `hidden1a` cDAE context (ivae_ardae.py:572-580): cat(h0, h) for the MLP / conv aux models, h alone for auxresconv
        self.hidden_dim = {"auxmnist": 2 * h_dim, "auxconv": 2 * h_dim, "auxresconv": h_dim, "auxtoy": 2 * h_dim}.get(self._kind, 0)
        self._build_params(layout.model_spec(self._kind, input_dim, noise_dim, h_dim, z_dim, num_hidden_layers, **getattr(self, "_spec_extra", {})),
                           {"encode": _EncodeBox})
        object.__setattr__(self.encode, "_owner_ref", weakref.ref(self))   # `model.encode(x, std=0)` (ivae_ardae.py:735)
        self.reset_parameters()

    def reset_parameters(self):
        self._default_init()
        with torch.no_grad():
            p = dict(self.named_parameters())
            if self._kind in ("resconv", "auxresconv"):
                # WNlinear / WNconv2d / WeightNormalizedLinear.reset_parameters (layers2.py:66-71,184-190, layers.py:40-45): direction and bias
                # U(+-1/sqrt(in_features | in_channels)), scale 1; the plain nn.Linear layers of the aux sampler keep torch's default
                spec = dict(self._spec)
                for name, t in p.items():
                    if name.endswith(".scale"):
                        t.fill_(1.0)
                    elif name.endswith(".direction") or (name[:-len("bias")] + "direction") in spec:
                        d = spec[name if name.endswith(".direction") else name[:-len("bias")] + "direction"]
                        t.uniform_(-1.0 / math.sqrt(d[1]), 1.0 / math.sqrt(d[1]))
                return
            if self._kind == "auxmnist":                  # self.apply(weight_init) on the whole model (ivae/auxmnist.py:172-174)
                if self.do_xavier:
                    for t in p.values():
                        nn.init.xavier_uniform_(t) if t.dim() == 2 else t.zero_()
                return
            if self._kind in ("conv", "auxconv"):         # self.apply(weight_init): xavier-uniform on Conv2d / Linear, zero biases;
                if self.do_xavier:                        # ConvTranspose2d keeps torch's default init (vae/auxconv.py:18-23)
                    for name, t in p.items():
                        if "deconv" in name or "logit_fn" in name:
                            continue
                        nn.init.xavier_uniform_(t) if t.dim() >= 2 else t.zero_()
                return
            if self._kind == "auxtoy":                    # init='gaussian' reaches the toy Decoder only (ivae/auxtoy.py:165, vae/toy.py)
                if self.init == "gaussian":
                    p["decode.reparam.mean_fn.weight"].normal_()
                return
            if self._kind == "mnist":                     # decode.apply(weight_init): xavier-uniform W, zero b (ivae/mnist.py:20-25,235)
                for name, t in p.items():
                    if name.startswith("decode."):
                        nn.init.xavier_uniform_(t) if t.dim() == 2 else t.zero_()
            else:                                         # ivae/toy.py:719-720
                if self.init == "gaussian":
                    p["decode.reparam.mean_fn.weight"].normal_()
            if self.init == "gaussian":                   # ivae/mnist.py:158-159
                p["encode.fc.fc.weight"].normal_()

    def _packed_weights(self):
        """See ConditionalARDAE._packed_weights."""
        if self._pack_is_current():
            return self._packed
        lib = L.lib()
        if self._packed is None:
            self._packed = self._ws(lib.ardae_model_packed_floats(ctypes.byref(self._desc)))
        L.check(lib.ardae_model_pack(ctypes.byref(self._desc), L.ptr(self._flat), L.ptr(self._packed), L.stream_ptr()), "ardae_model_pack")
        self._note_packed()
        return self._packed

    def _set_logvar_clips(self, clip_z0_logvar, clip_z_logvar):
        """clip_z0_logvar / clip_z_logvar of the hierarchical MLP models (ivae/auxmnist.py:144-161: 'none' -> None; the choices are
        NormalDistribution.clip_logvar's, models/reparam.py:17-41; any other name leaves the log-variance as it is there, and is refused here)."""
        for c in (clip_z0_logvar, clip_z_logvar):
            if c not in L.LOGVAR_CLIP:
                raise NotImplementedError(f"clip_logvar {c!r}: models/reparam.py:17-41 knows {sorted(k for k in L.LOGVAR_CLIP if k)}")
        self.clip_z0_logvar = None if clip_z0_logvar == "none" else clip_z0_logvar
        self.clip_z_logvar = None if clip_z_logvar == "none" else clip_z_logvar
        self._flag_extra = (L.LOGVAR_CLIP[clip_z0_logvar] << L.MODEL_CLIP_Z0_SHIFT) | (L.LOGVAR_CLIP[clip_z_logvar] << L.MODEL_CLIP_Z_SHIFT)

    def _x(self, input):
        self._require_gpu(input)
        return _f32c(input).view(input.size(0), self.input_dim)

    def _sample(self, x, nz, std, noise):
        """z = f(x, std*eps) without autograd (the reference loop detaches every use: ivae_ardae.py:735,748-750)."""
        B = x.size(0)
        if noise is None and std is not None and float(std) == 0.0:
            nptr = None                                   # encode(x, std=0): the draw is multiplied by zero
        else:
            if noise is None:
                noise = rng.normal((self._noise_numel(B, nz),), x.device)
                if std is not None:
                    noise = noise * float(std)
            noise = self._noise_rows(noise, B * nz)
            nptr = L.ptr(noise)
        lib = L.lib()
        ws = self._ws(lib.ardae_model_workspace_floats(ctypes.byref(self._desc), B, nz, 0))
        z = torch.empty(B * nz, self.z_dim, device=x.device)
        L.check(lib.ardae_model_encode(ctypes.byref(self._desc), L.ptr(self._flat), L.ptr(self._packed_weights()), L.ptr(x), nptr, B, nz,
                                       L.ptr(ws), ws.numel(), L.ptr(z), L.stream_ptr()), "ardae_model_encode")
        return z.view(B, nz, self.z_dim)

    def _noise_numel(self, B, nz):
        """Floats of one sampler call's draws for B images x nz rows."""
        return B * nz * self._noise_width

    def _noise_rows(self, noise, rows):
        """[rows, noise width] fp32 contiguous; the aux models also take the pair (eps0 [rows, noise_dim], eps [rows, z_dim])."""
        if isinstance(noise, (tuple, list)):
            noise = torch.cat([_f32c(n).view(rows, -1) for n in noise], 1)
        return _f32c(noise).view(rows, self._noise_width)

    def forward_hidden(self, input, std=None, nz=1, noise=None):
        return self._sample(self._x(input), nz, std, noise)

    def _hidden(self, input):
        """cat(h0, h) [B, 2 h] of the std = 0 pass (aux models; the `hidden1a` cDAE context)."""
        if self._kind not in AUX_KINDS:
            raise NotImplementedError("hidden contexts exist for the aux models only")
        x = self._x(input)
        B, lib = x.size(0), L.lib()
        ws = self._ws(lib.ardae_model_workspace_floats(ctypes.byref(self._desc), B, 1, 0))
        hid = torch.empty(B, self.hidden_dim, device=x.device)
        L.check(lib.ardae_model_encode_hidden(ctypes.byref(self._desc), L.ptr(self._flat), L.ptr(self._packed_weights()), L.ptr(x), B, L.ptr(ws),
                                              ws.numel(), None, L.ptr(hid), L.stream_ptr()), "ardae_model_encode_hidden")
        return hid

    def _decoder_sample(self, z_rows, dec_noise=None):
        """(x_sample, decoder mean) of the reference's Decoder.forward for z_rows [R, z]: Bernoulli decoders return the relaxed
        (logistic-sigmoid, T = 1) sample and sigmoid(logit) (reparam.py:111-158, ivae/mnist.py:188-199,300); the toy model's
        Gaussian decoder returns mu + exp(logvar/2) eps and mu (reparam.py:42-51, ivae/toy.py:725-737,858).  dec_noise injects
        the draw (uniform [R, D] resp. normal [R, D]); default: the library's Philox stream."""
        out = self.decode_params(z_rows)
        R, lib = out[0].size(0), L.lib()
        sample = torch.empty_like(out[0])
        if self._kind in GAUSSIAN_DECODERS:
            e = _f32c(dec_noise).view(R, self.input_dim) if dec_noise is not None else rng.normal((R, self.input_dim), out[0].device)
            L.check(lib.ardae_gaussian_sample(L.ptr(out[0]), L.ptr(out[1]), L.ptr(e), out[0].numel(), L.ptr(sample), L.stream_ptr()),
                    "ardae_gaussian_sample")
            return sample, out[0]
        u = _f32c(dec_noise).view(R, self.input_dim) if dec_noise is not None else rng.uniform((R, self.input_dim), out[0].device)
        mean = torch.empty_like(out[0])
        L.check(lib.ardae_relaxed_bernoulli(L.ptr(out[0]), L.ptr(u), out[0].numel(), 1.0, L.ptr(sample), L.ptr(mean), L.stream_ptr()),
                "ardae_relaxed_bernoulli")
        return sample, mean

    def forward(self, input, beta=1.0, eta=0.0, lmbd=0.0, std=None, nz=1, noise=None, dec_noise=None):
        """-> (x_sample, decoder mean, z, loss, recon.detach(), prior.detach()) like the reference (ivae/mnist.py:267-301).
        The first two entries feed only the reference's visualisation code; they cost one extra decoder pass on the B*nz rows
        and are skipped (None, None) when `self.return_samples` is False - the fused engine never produces them."""
        if lmbd > 0:
            raise NotImplementedError                     # ivae/mnist.py:288-290
        x = self._x(input)
        B = x.size(0)
        if noise is None:
            noise = rng.normal((self._noise_numel(B, nz),), x.device)
            if std is not None:
                noise = noise * float(std)
        noise = self._noise_rows(noise, B * nz)
        z, loss, losses = _VaeFn.apply(self, x, noise, beta, nz, *self.parameters())
        xs, xm = (None, None)
        if self.return_samples:
            with torch.no_grad():
                xs, xm = self._decoder_sample(z.detach().reshape(B * nz, self.z_dim), dec_noise)
        return xs, xm, z, loss, losses[1].detach(), losses[2].detach()


    # ---- evaluation --------------------------------------------------------------------------------
    def decode_params(self, z):
        """Decoder heads for z [R, z_dim] -> (logits,) or (mean, logvar)."""
        self._require_gpu(z)
        z = _f32c(z).view(-1, self.z_dim)
        R, lib = z.size(0), L.lib()
        ws = self._ws(lib.ardae_model_workspace_floats(ctypes.byref(self._desc), R, 1, 2))
        o0 = torch.empty(R, self.input_dim, device=z.device)
        o1 = torch.empty(R, self.input_dim, device=z.device) if self._kind in GAUSSIAN_DECODERS else None
        L.check(lib.ardae_model_decode(ctypes.byref(self._desc), L.ptr(self._flat), L.ptr(self._packed_weights()), L.ptr(z), R, L.ptr(ws),
                                       ws.numel(), L.ptr(o0), L.ptr(o1), L.stream_ptr()), "ardae_model_decode")
        return (o0,) if o1 is None else (o0, o1)

    def generate(self, batch_size=1, z=None, dec_noise=None):
        """-> (x_sample, decoder mean, z) with z ~ N(0, I) (ivae/mnist.py:303-316, ivae/toy.py:862-873).  z / dec_noise inject the draws."""
        with torch.no_grad():
            z = rng.normal((batch_size, self.z_dim), self._flat.device) if z is None else _f32c(z).view(batch_size, self.z_dim)
            xs, xm = self._decoder_sample(z, dec_noise)
        return xs, xm, z

    def logprob(self, input, sample_size=128, z=None, std=None, enc_noise=None, prop_noise=None):
        """IWAE-k bound with a full-covariance Gaussian fitted to the encoder samples as proposal
        (logprob_w_cov_gaussian_posterior, ivae/mnist.py:378-437).  All images are processed at once: sampler and decoder
        are the HIP kernels, the k x z covariance / Cholesky / log-mean-exp glue is batched torch (off the timed path).
        enc_noise [B, k, noise_dim] (aux models: the pair ([B, k, noise_dim], [B, k, z])) / prop_noise [B, k, z] inject the draws."""
        x = self._x(input)
        B, k, zd = x.size(0), sample_size, self.z_dim
        assert sample_size >= 2 * self.z_dim                 # ivae/mnist.py:382
        with torch.no_grad():
            ke = k * k if self._kind == "auxtoy" else k       # ToyAuxIPVAE fits the proposal to Encoder._forward(nz=k): k z0's x k z's (ivae/auxtoy.py:313)
            if self._kind == "auxtoy":
                if enc_noise is not None and not isinstance(enc_noise, (tuple, list)):
                    raise ValueError("ToyAuxIPVAE.logprob: enc_noise is the pair (eps0 [B, k, noise_dim], eps [B, k k, z_dim])")
            elif isinstance(enc_noise, (tuple, list)):
                enc_noise = tuple(n.reshape(B * k, -1) for n in enc_noise)
            elif enc_noise is not None:
                enc_noise = enc_noise.reshape(B * k, self._noise_width)
            zs = self._sample(x, ke, std, enc_noise)          # [B,k,z]
            mu = zs.mean(1)
            zc = zs - mu.unsqueeze(1)
            cov = zc.transpose(1, 2) @ zc / (ke - 1)          # utils/stat.py:127-158
            if self._kind in AUX_KINDS:
                cov = cov + 1e-5 * torch.eye(zd, device=cov.device)      # ivae/auxmnist.py:321, ivae/auxconv.py, ivae/auxresconv.py:299
            cov = cov.contiguous()
            if zd <= 64:
                Lc = torch.empty_like(cov)                    # all B factorisations in one launch (MultivariateNormal's, ivae/mnist.py:397)
                L.check(L.lib().ardae_cholesky_batched(L.ptr(cov), B, zd, L.ptr(Lc), L.stream_ptr()), "ardae_cholesky_batched")
            else:
                # the LDS-resident kernel holds one z x z matrix per workgroup (z <= 64, every shipped recipe has z 32); larger latent
                # spaces factorise on the device through the library solver MultivariateNormal itself would use - evaluation only
                Lc, info = torch.linalg.cholesky_ex(cov)
                Lc = torch.where((info == 0).view(B, 1, 1), Lc, torch.full_like(Lc, float("nan")))
            if not bool(torch.isfinite(Lc).all()):
                raise ValueError("logprob: a sample covariance is not positive definite (torch.distributions would raise here too)")
            if prop_noise is None:
                prop_noise = rng.normal((B, k, zd), x.device)
            e = _f32c(prop_noise).view(B, k, zd)
            newz = (mu.unsqueeze(1) + e @ Lc.transpose(1, 2)).contiguous()
            logq = -0.5 * (e ** 2).sum(2) - torch.log(torch.diagonal(Lc, dim1=1, dim2=2)).sum(1, keepdim=True) - 0.5 * zd * math.log(2 * math.pi)
            out = self.decode_params(newz.view(B * k, zd))
            rec = torch.empty(B * k, device=x.device); pri = torch.empty(B * k, device=x.device)
            L.check(L.lib().ardae_model_loss_rows(ctypes.byref(self._desc), L.ptr(out[0]), L.ptr(out[1]) if len(out) > 1 else None, L.ptr(x),
                                                  L.ptr(newz), B * k, k, L.ptr(rec), L.ptr(pri), L.stream_ptr()), "ardae_model_loss_rows")
            lw = -rec.view(B, k) - pri.view(B, k) - logq
            m, _ = lw.max(1, keepdim=True)
            return (torch.log(torch.mean((lw - m).exp(), 1, keepdim=True) + 1e-10) + m).mean()


class MNISTIPVAE(ImplicitPosteriorVAE):
    """models/ivae/mnist.py::ImplicitPosteriorVAE (`--model mnist-concat`)."""
    _kind = "mnist"


class ConvIPVAE(ImplicitPosteriorVAE):
    """models/ivae/conv.py::ImplicitPosteriorVAE (`--model mnist-conv`, BASELINE config #4).  The reference's decoder only
    produces 28x28 outputs (cfg #5 note), so input_height=28 / input_channels=1 are required."""
    _kind = "conv"

    def __init__(self, energy_func=normal_energy_func, input_height=28, input_channels=1, z_dim=32, noise_dim=100,
                 nonlinearity="softplus", do_xavier=True):
        if input_height != 28 or input_channels != 1:
            raise NotImplementedError("ConvIPVAE: the reference decoder (models/vae/conv.py:79-136) is hard-wired to 28x28x1")
        self.input_height, self.input_channels, self.do_xavier = input_height, input_channels, do_xavier
        super().__init__(energy_func, input_height * input_height * input_channels, noise_dim, 800, z_dim, nonlinearity, 1, "none", "concat")


class MNISTAuxIPVAE(ImplicitPosteriorVAE):
    """models/ivae/auxmnist.py::ImplicitPosteriorVAE (`--model auxmnist`, the shipped "hierarchical mlp" recipe): AuxEncoder -> z0 ->
    SimpleEncoder -> z, both Gaussian reparameterisations, `enc_type='simple'`; `clip_z0_logvar` / `clip_z_logvar`: the `clip_logvar` choices of
    models/reparam.py:17-41 ('hard', 'softplus', 'spm10' .. 'spm2', 'tanh', '2tanh'; 'none' in the shipped recipe).  A sampler call takes two draws;
    `noise=` accepts the pair (eps0 [rows, noise_dim], eps [rows, z_dim]) or one [rows, noise_dim + z_dim] tensor."""
    _kind = "auxmnist"
    _enc_types = ("simple",)

    def __init__(self, energy_func=normal_energy_func, input_dim=784, noise_dim=100, h_dim=300, z_dim=32, nonlinearity="softplus",
                 num_hidden_layers=2, enc_type="simple", clip_z0_logvar=None, clip_z_logvar=None, do_xavier=True):
        if enc_type != "simple":
            raise NotImplementedError                     # ivae/auxmnist.py:72-73
        self.do_xavier = do_xavier
        self._set_logvar_clips(clip_z0_logvar, clip_z_logvar)
        super().__init__(energy_func, input_dim, noise_dim, h_dim, z_dim, nonlinearity, num_hidden_layers, "none", enc_type)


class ToyAuxIPVAE(ImplicitPosteriorVAE):
    """models/ivae/auxtoy.py::ImplicitPosteriorVAE (`--model auxmlp`, ivae_ardae.py:443-454): MNISTAuxIPVAE's networks without the 2x - 1 rescale,
    the toy problem's Gaussian decoder, and a SQUARE sampling scheme - a call with nz rows per image draws q = int(sqrt(nz)) z0's per image and
    q z's per z0 (`forward_hidden` / `forward`, :215,230; nz must be a square), `logprob(sample_size=k)` fits its proposal to k x k encoder
    samples (:313).  `noise=` of a sampler call: the pair (eps0 [B q, noise_dim], eps [B q q, z_dim]) or ONE flat tensor [eps0 block | eps block]."""
    _kind = "auxtoy"
    _enc_types = ("simple",)

    def __init__(self, energy_func=normal_energy_func, input_dim=2, noise_dim=2, h_dim=64, z_dim=2, nonlinearity="tanh", num_hidden_layers=1,
                 init="gaussian", enc_type="simple", clip_z0_logvar=None, clip_z_logvar=None):
        if enc_type != "simple":
            raise NotImplementedError                     # ivae/auxtoy.py:70-71
        self._set_logvar_clips(clip_z0_logvar, clip_z_logvar)
        super().__init__(energy_func, input_dim, noise_dim, h_dim, z_dim, nonlinearity, num_hidden_layers, init, enc_type)

    @staticmethod
    def _q(nz):
        q = math.isqrt(int(nz))
        if q * q != nz:
            raise ValueError(f"ToyAuxIPVAE draws q z0's x q z's per image: nz = {nz} is not a square (ivae/auxtoy.py:215)")
        return q

    def _noise_numel(self, B, nz):
        return B * self._q(nz) * self.noise_dim + B * nz * self.z_dim

    def _noise_rows(self, noise, rows):
        if isinstance(noise, (tuple, list)):
            noise = torch.cat([_f32c(n).reshape(-1) for n in noise])
        return _f32c(noise).reshape(-1)


class MNISTConvAuxIPVAE(ImplicitPosteriorVAE):
    """models/ivae/auxconv.py::ImplicitPosteriorVAE (`--model auxconv`, the shipped "hierarchical conv" recipe): the hierarchical
    sampler with two conv trunks and ConvIPVAE's decoder; 28x28x1 only; the hidden1a context is [B, 1600]."""
    _kind = "auxconv"

    def __init__(self, energy_func=normal_energy_func, input_height=28, input_channels=1, z0_dim=100, h_dim=300, z_dim=32,
                 nonlinearity="softplus", do_xavier=True):
        if input_height != 28 or input_channels != 1:
            raise NotImplementedError("MNISTConvAuxIPVAE: the reference decoder (models/vae/conv.py:79-136) is hard-wired to 28x28x1")
        self.input_height, self.input_channels, self.z0_dim, self.do_xavier = input_height, input_channels, z0_dim, do_xavier
        super().__init__(energy_func, 784, z0_dim, 800, z_dim, nonlinearity, 1, "none", "concat")


class ResConvIPVAE(ImplicitPosteriorVAE):
    """models/ivae/resconv.py::ImplicitPosteriorVAE as the ten `--model resconv*` choices build it (ivae_ardae.py:323-442): weight-normalised
    residual-conv trunk and decoder, ELU, 28x28x1, c_dim 512, do_center either way, and every sampler head of `enc_type`:
    'mlp' (resconv / resconvct), 'res-wn-mlp' (-res: the shipped "implicit resconv" recipe), 'res-mlp' (-res2), 'res-wn-mlp-lin' (-res3),
    'res-mlp-lin' (-res4), with 1 .. 4 hidden layers (`--model-n-layers`)."""
    _kind = "resconv"

    def __init__(self, energy_func=normal_energy_func, input_height=28, input_channels=1, z_dim=32, noise_dim=100, c_dim=512, h_dim=800,
                 num_hidden_layers=1, nonlinearity="elu", do_center=False, do_m5bias=False, enc_noise=False, enc_type="mlp"):
        if input_height != 28 or input_channels != 1:
            raise AssertionError("input_height == 28 and input_channels == 1")           # ivae/resconv.py:218-219
        assert enc_type in layout.RESCONV_HEADS                                          # ivae/resconv.py:77
        assert num_hidden_layers > 0                                                     # ivae/resconv.py:73
        if enc_noise or do_m5bias or c_dim != 512 or num_hidden_layers > 4:
            raise NotImplementedError("the HIP engine builds ResConvIPVAE as ivae_ardae.py's --model resconv* choices do: c_dim=512, no enc_noise / "
                                      "do_m5bias, at most 4 hidden layers (any enc_type, do_center either way)")
        if enc_type != "mlp" and c_dim + noise_dim == h_dim:
            raise NotImplementedError("c_dim + noise_dim == h_dim makes the first ResLinear's skip the concatenated input itself; not built")
        self._flag_extra = layout.RESCONV_HEADS[enc_type] << L.MODEL_HEAD_SHIFT
        self._spec_extra = {"enc_type": enc_type}
        self.input_height, self.input_channels, self.c_dim = input_height, input_channels, c_dim
        self.do_center, self.do_m5bias, self.enc_noise = do_center, do_m5bias, enc_noise
        super().__init__(energy_func, 784, noise_dim, h_dim, z_dim, nonlinearity, num_hidden_layers, "none", "concat")
        self.enc_type = enc_type


class MNISTResConvAuxIPVAE(ImplicitPosteriorVAE):
    """models/ivae/auxresconv.py::ImplicitPosteriorVAE (`--model auxresconvct`, the shipped "hierarchical resconv" recipe): the
    residual-conv trunk shared by a Gaussian z0 head and the z head (log-variances clipped 'spm4'), the residual-conv decoder; its
    hidden1a cDAE context is h [B, c_dim].  A sampler call takes two draws: noise = (eps0 [rows, z0_dim], eps [rows, z_dim])."""
    _kind = "auxresconv"

    def __init__(self, energy_func=normal_energy_func, input_height=28, input_channels=1, z0_dim=100, z_dim=32, c_dim=450, nonlinearity="elu",
                 do_center=False, do_m5bias=False):
        assert input_height == 28 and input_channels == 1 and nonlinearity == "elu"     # ivae/auxresconv.py:67-69
        if do_m5bias:
            raise NotImplementedError("the HIP engine builds MNISTResConvAuxIPVAE as --model auxresconvct / auxresconv do (no do_m5bias; do_center either way)")
        self.input_height, self.input_channels, self.z0_dim, self.c_dim, self.do_center, self.do_m5bias = input_height, input_channels, z0_dim, c_dim, bool(do_center), False
        super().__init__(energy_func, 784, z0_dim, c_dim, z_dim, nonlinearity, 1, "none", "concat")


class MNISTResConvAuxIPVAEClipped(MNISTResConvAuxIPVAE):
    """models/ivae/auxresconv2.py::ImplicitPosteriorVAE (`--model auxresconv-clip` / `auxresconvct-clip`, ivae_ardae.py:507-534): the
    same networks with the two log-variance heads built WITHOUT the 'spm4' clip (:71-72) and z0 = mu0 + (std exp(lv0 / 2) + 1) eps0
    (`sample_gaussian(..., min_std=1.)`, :29-36,91).  Consequence the training loop inherits: `encode(x, std=0)` /
    `forward_hidden(x, std=0)` are RANDOM draws here (z0 = mu0 + eps0).  `noise` of a std = 0 call is that unscaled eps0 [B, z0_dim]
    (default: a fresh Philox draw); other calls take unscaled draws (std None / 1: the only values the reference's loop uses)."""
    _clipped = True

    def __init__(self, *a, **k):
        self._flag_extra = L.MODEL_CLIPPED
        super().__init__(*a, **k)

    def _std0(self, x, raw0, want_hidden):
        B, lib = x.size(0), L.lib()
        raw = _f32c(raw0).view(B, -1)[:, :self.z0_dim].contiguous() if raw0 is not None else rng.normal((B, self.z0_dim), x.device)
        ws = self._ws(lib.ardae_model_workspace_floats(ctypes.byref(self._desc), B, 1, 0))
        out = torch.empty(B, self.hidden_dim if want_hidden else self.z_dim, device=x.device)
        L.check(lib.ardae_model_encode_hidden_raw(ctypes.byref(self._desc), L.ptr(self._flat), L.ptr(self._packed_weights()), L.ptr(x), L.ptr(raw), B,
                                                  L.ptr(ws), ws.numel(), None if want_hidden else L.ptr(out), L.ptr(out) if want_hidden else None,
                                                  L.stream_ptr()), "ardae_model_encode_hidden_raw")
        return out

    def _sample(self, x, nz, std, noise):
        if std is not None and float(std) == 0.0:
            if nz != 1:
                raise NotImplementedError("the clipped class at std = 0 is built for nz = 1 (what ivae_ardae.py:735-748,815-826 calls)")
            if isinstance(noise, (tuple, list)):
                noise = noise[0]
            return self._std0(x, noise, False).view(x.size(0), 1, self.z_dim)
        if std is not None and float(std) != 1.0:
            raise NotImplementedError("MNISTResConvAuxIPVAEClipped: std must be None, 1 or 0 (the unscaled eps0 term needs the draw and std apart)")
        return super()._sample(x, nz, None, noise)

    def _hidden(self, input, raw0=None):
        return self._std0(self._x(input), raw0, True)


class ToyIPVAE(ImplicitPosteriorVAE):
    """models/ivae/toy.py::ImplicitPosteriorVAE with enc_type='concat' (`--model mlp-concat`)."""
    _kind = "toy"

    def __init__(self, energy_func=normal_energy_func, input_dim=2, noise_dim=2, h_dim=64, z_dim=2, nonlinearity="tanh",
                 num_hidden_layers=1, init="gaussian", enc_type="scale"):
        if enc_type != "concat":
            raise NotImplementedError(f"enc_type {enc_type!r}: only 'concat' is on the BASELINE path (SURVEY 2 #5)")
        super().__init__(energy_func, input_dim, noise_dim, h_dim, z_dim, nonlinearity, num_hidden_layers, init, enc_type)
