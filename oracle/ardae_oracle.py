"""CPU oracle for the AR-DAE-VAE inner training loop  --  TEST INFRASTRUCTURE ONLY.

This module is a plain-PyTorch (CPU, autograd) restatement of the reference's
hot path.  It is the *checker* for the HIP kernels and the `cpu_baseline` leg of
bench.py; nothing under `pytorch-ardae-vae_amd/` may import it (the product path
fails loudly when the HIP library is missing).

Parity status: PINNED.  `oracle/gen_golden.py` imports the reference's own classes
from /root/reference (in the build container only), loads identical parameters and
noise, and checks every function here against them before writing the fixtures in
`tests/golden/` (the reference ships no tests or golden vectors of its own).

Everything is functional: networks are `dict[str, Tensor]` keyed by the reference's
`state_dict()` names, weights are `[out, in]` like `nn.Linear`.

Reference map (file:line in /root/reference):
  mlp()                 models/layers.py:477-515      (MLP)
  act()                 utils/models.py:14-32         (get_nonlinear_func)
  mnist_encode()        models/ivae/mnist.py:76-121,123-165
  mnist_decode()        models/ivae/mnist.py:167-199
  toy_encode()          models/ivae/toy.py:60-103,154-194 + models/layers.py:681-724
  toy_decode()          models/ivae/toy.py:694-737
  vae_forward()         models/ivae/mnist.py:240-249,267-301 / toy.py:777-873
  cdae_grad_forward()   models/graddae/mlp.py:400-444
  cdae_grad_glogprob()  models/graddae/mlp.py:446-483
  cdae_res_forward()    models/resdae/mlp.py:344-381
  cdae_res_glogprob()   models/resdae/mlp.py:383-413
  latent_stats()        ivae_ardae.py:753-761
  adam_ref_step()       utils/optim.py:49-108
  rmsprop_step()        torch.optim.RMSprop as constructed at ivae_ardae.py:625-626
  train_step()          ivae_ardae.py:707-846
  iwae_logprob()        models/ivae/mnist.py:378-437, utils/stat.py:65-85,127-158
  res_conv() / res_linear_wn() / res_linear()   models/layers2.py:237-265,305-352, models/layers.py:25-85,559-622
  resconv_trunk() / resconv_decode()            models/ivae/resconv.py:53-180, models/vae/resconv.py:79-136,
                                                models/vae/auxresconv.py:26-185, models/ivae/auxresconv.py:48-134
"""
import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

LOG2PI = math.log(2.0 * math.pi)
AUX_KINDS = ("auxmnist", "auxconv", "auxresconv", "auxtoy")     # hierarchical samplers: two draws per call


# --------------------------------------------------------------------------- #
# configuration
# --------------------------------------------------------------------------- #
@dataclass
class ModelCfg:
    """Implicit-posterior VAE hyper-parameters (ivae_ardae.py:295-314)."""
    kind: str = "mnist"          # "mnist" (MNISTIPVAE) | "toy" (ToyIPVAE, enc_type='concat') | "conv" (ConvIPVAE, 28x28x1) | "auxmnist" | "auxconv"
                                 # | "resconv" (ResConvIPVAE do_center, enc_type='res-wn-mlp': --model resconvct-res; c_dim 512, h_dim = fc width)
                                 # | "auxresconv" (MNISTResConvAuxIPVAE do_center: --model auxresconvct; noise_dim = z0_dim, h_dim = c_dim 450)
    input_dim: int = 784
    noise_dim: int = 100
    h_dim: int = 256
    z_dim: int = 32
    n_layers: int = 2            # --model-n-layers
    nonlin: str = "softplus"
    do_center: bool = True       # residual-conv kinds: the trunk sees 2x - 1 (--model resconvct-res / auxresconvct) or x (resconv-res / auxresconv)
    clipped: bool = False        # "auxresconv" only: MNISTResConvAuxIPVAEClipped (--model auxresconv-clip / auxresconvct-clip, ivae/auxresconv2.py):
                                 # NO 'spm4' clip of the two log-variances and z0 = mu0 + (std exp(lv0 / 2) + 1) eps0 (`min_std=1.`, :91)
    clip_z0: str = "none"        # "auxmnist" / "auxtoy": clip_z0_logvar / clip_z_logvar (ivae/auxmnist.py:144-161), the choices of
    clip_z: str = "none"         # NormalDistribution.clip_logvar (models/reparam.py:17-41)
    enc_type: str = "res-wn-mlp"  # "resconv" only: the sampler head (models/ivae/resconv.py:101-116): 'mlp' (--model resconv / resconvct), 'res-wn-mlp'
                                 # (-res), 'res-mlp' (-res2), 'res-wn-mlp-lin' (-res3), 'res-mlp-lin' (-res4); n_layers = --model-n-layers


@dataclass
class CdaeCfg:
    """Conditional AR-DAE hyper-parameters (ivae_ardae.py:583-606)."""
    kind: str = "grad"           # "grad" (MLPGradCARDAE) | "res" (MLPResCARDAE)
    input_dim: int = 32
    context_dim: int = 32
    h_dim: int = 256
    n_layers: int = 3            # --cdae-n-layers
    nonlin: str = "softplus"


@dataclass
class TrainCfg:
    """Loop constants (run_vae_dbmnist.sh / run_vae_25gaussians.sh)."""
    delta: float = 0.1
    std_scale: float = 1e4
    nz_cdae: int = 256
    nz_model: int = 1
    nstd: int = 1
    num_cdae_updates: int = 1
    beta: float = 1.0
    m_lr: float = 1e-4
    m_beta1: float = 0.5
    d_lr: float = 1e-4
    d_momentum: float = 0.5
    m_optimizer: str = "adam"    # --m-optimizer / --d-optimizer: sgd | adam | amsgrad | rmsprop (ivae_ardae.py:545-556,612-622); the shipped
    d_optimizer: str = "rmsprop" # recipes pass adam / rmsprop (argparse's own default for --d-optimizer is adam)
    d_beta1: float = 0.5         # --d-beta1 (cDAE Adam / amsgrad)
    ctx_type: str = "lt0"        # --cdae-ctx-type: "lt0" (context = encode(x, std=0)) | "hidden1a" (aux models: encoder hiddens) | "data"
    ctx_center: bool = True      # ctx_type "data": 2x - 1 when 'mnist' is in the dataset's name (ivae_ardae.py:731-734), x itself otherwise


# --------------------------------------------------------------------------- #
# parameter specs (names/shapes exactly as the reference's named_parameters())
# --------------------------------------------------------------------------- #
def _mlp_spec(prefix, din, dh, dout, n_hidden):
    spec = []
    for i in range(n_hidden):
        spec += [(f"{prefix}layers.{i}.weight", (dh, din if i == 0 else dh)),
                 (f"{prefix}layers.{i}.bias", (dh,))]
    spec += [(f"{prefix}fc.weight", (dout, din if n_hidden == 0 else dh)),
             (f"{prefix}fc.bias", (dout,))]
    return spec


def _ctxcat_mlp_spec(prefix, din, dctx, dh, dout, n_hidden):
    # ContextConcatMLP (models/layers.py:681-724): every layer eats cat([hidden, ctx])
    spec = []
    for i in range(n_hidden):
        spec += [(f"{prefix}layers.{i}.weight", (dh, (din if i == 0 else dh) + dctx)),
                 (f"{prefix}layers.{i}.bias", (dh,))]
    spec += [(f"{prefix}fc.weight", (dout, (din if n_hidden == 0 else dh) + dctx)),
             (f"{prefix}fc.bias", (dout,))]
    return spec


def model_param_spec(c: ModelCfg):
    if c.kind == "mnist":
        s = _mlp_spec("encode.inp_encode.", c.input_dim, c.h_dim, c.h_dim, c.n_layers + 1)
        s += _mlp_spec("encode.fc.", c.h_dim + c.noise_dim, c.h_dim, c.z_dim, 1)
        s += _mlp_spec("decode.main.", c.z_dim, c.h_dim, c.h_dim, c.n_layers)
        s += [("decode.reparam.logit_fn.weight", (c.input_dim, c.h_dim)),
              ("decode.reparam.logit_fn.bias", (c.input_dim,))]
        return s
    if c.kind == "conv":   # models/ivae/conv.py:44-136 + models/vae/conv.py:79-136 (28x28x1 only: the decoder is hard-wired to it)
        s = [("encode.conv1.weight", (16, 1, 5, 5)), ("encode.conv1.bias", (16,)),
             ("encode.conv2.weight", (32, 16, 5, 5)), ("encode.conv2.bias", (32,)),
             ("encode.conv3.weight", (32, 32, 5, 5)), ("encode.conv3.bias", (32,)),
             ("encode.fc4.weight", (800, 512 + c.noise_dim)), ("encode.fc4.bias", (800,)),
             ("encode.fc5.weight", (c.z_dim, 800)), ("encode.fc5.bias", (c.z_dim,))]
        s += _mlp_spec("decode.fc.", c.z_dim, 300, 512, 1)
        s += [("decode.deconv1.weight", (32, 32, 5, 5)), ("decode.deconv1.bias", (32,)),
              ("decode.deconv2.weight", (32, 16, 5, 5)), ("decode.deconv2.bias", (16,)),
              ("decode.reparam.logit_fn.weight", (16, 1, 5, 5)), ("decode.reparam.logit_fn.bias", (1,))]
        return s
    if c.kind == "auxtoy":
        # models/ivae/auxtoy.py:44-130 (Encoder = AuxEncoder + SimpleEncoder of models/vae/auxtoy.py, the auxmnist networks without the 2x - 1
        # rescale) + models/vae/toy.py Decoder (Gaussian: mean_fn / logvar_fn)
        s = _mlp_spec("encode.aux_encode.main.", c.input_dim, c.h_dim, c.h_dim, c.n_layers - 1)
        s += [("encode.aux_encode.reparam.mean_fn.weight", (c.noise_dim, c.h_dim)), ("encode.aux_encode.reparam.mean_fn.bias", (c.noise_dim,)),
              ("encode.aux_encode.reparam.logvar_fn.weight", (c.noise_dim, c.h_dim)), ("encode.aux_encode.reparam.logvar_fn.bias", (c.noise_dim,))]
        s += _mlp_spec("encode.encode.fc.", c.input_dim + c.noise_dim, c.h_dim, c.h_dim, c.n_layers - 1)
        s += [("encode.encode.reparam.mean_fn.weight", (c.z_dim, c.h_dim)), ("encode.encode.reparam.mean_fn.bias", (c.z_dim,)),
              ("encode.encode.reparam.logvar_fn.weight", (c.z_dim, c.h_dim)), ("encode.encode.reparam.logvar_fn.bias", (c.z_dim,))]
        s += _mlp_spec("decode.main.", c.z_dim, c.h_dim, c.h_dim, c.n_layers - 1)
        s += [("decode.reparam.mean_fn.weight", (c.input_dim, c.h_dim)), ("decode.reparam.mean_fn.bias", (c.input_dim,)),
              ("decode.reparam.logvar_fn.weight", (c.input_dim, c.h_dim)), ("decode.reparam.logvar_fn.bias", (c.input_dim,))]
        return s
    if c.kind == "auxmnist":
        # models/ivae/auxmnist.py:47-132 (Encoder = AuxEncoder + SimpleEncoder of models/vae/auxmnist.py:31-190, enc_input = enc_noise
        # = False: the images and z0 enter encode.fc directly) + models/vae/mnist.py Decoder (n_layers - 1 hidden layers)
        s = _mlp_spec("encode.aux_encode.main.", c.input_dim, c.h_dim, c.h_dim, c.n_layers - 1)
        s += [("encode.aux_encode.reparam.mean_fn.weight", (c.noise_dim, c.h_dim)), ("encode.aux_encode.reparam.mean_fn.bias", (c.noise_dim,)),
              ("encode.aux_encode.reparam.logvar_fn.weight", (c.noise_dim, c.h_dim)), ("encode.aux_encode.reparam.logvar_fn.bias", (c.noise_dim,))]
        s += _mlp_spec("encode.encode.fc.", c.input_dim + c.noise_dim, c.h_dim, c.h_dim, c.n_layers - 1)
        s += [("encode.encode.reparam.mean_fn.weight", (c.z_dim, c.h_dim)), ("encode.encode.reparam.mean_fn.bias", (c.z_dim,)),
              ("encode.encode.reparam.logvar_fn.weight", (c.z_dim, c.h_dim)), ("encode.encode.reparam.logvar_fn.bias", (c.z_dim,))]
        s += _mlp_spec("decode.main.", c.z_dim, c.h_dim, c.h_dim, c.n_layers - 1)
        s += [("decode.reparam.logit_fn.weight", (c.input_dim, c.h_dim)),
              ("decode.reparam.logit_fn.bias", (c.input_dim,))]
        return s
    if c.kind == "auxconv":
        # models/ivae/auxconv.py:48-126 (AuxEncoder + Encoder of models/vae/auxconv.py:32-140: two conv trunks 1->16->32->32, k5 s2 p2) +
        # models/vae/conv.py Decoder; 28x28x1 only; noise_dim = z0_dim, h_dim = 800 (the fc width)
        def trunk(prefix, fc_in):
            return [(prefix + "conv1.weight", (16, 1, 5, 5)), (prefix + "conv1.bias", (16,)),
                    (prefix + "conv2.weight", (32, 16, 5, 5)), (prefix + "conv2.bias", (32,)),
                    (prefix + "conv3.weight", (32, 32, 5, 5)), (prefix + "conv3.bias", (32,)),
                    (prefix + "fc.weight", (800, fc_in)), (prefix + "fc.bias", (800,))]
        def heads(prefix, out):
            return [(prefix + "reparam.mean_fn.weight", (out, 800)), (prefix + "reparam.mean_fn.bias", (out,)),
                    (prefix + "reparam.logvar_fn.weight", (out, 800)), (prefix + "reparam.logvar_fn.bias", (out,))]
        s = trunk("encode.aux_encode.", 512) + heads("encode.aux_encode.", c.noise_dim)
        s += trunk("encode.encode.", 512 + c.noise_dim) + heads("encode.encode.", c.z_dim)
        s += _mlp_spec("decode.fc.", c.z_dim, 300, 512, 1)
        s += [("decode.deconv1.weight", (32, 32, 5, 5)), ("decode.deconv1.bias", (32,)),
              ("decode.deconv2.weight", (32, 16, 5, 5)), ("decode.deconv2.bias", (16,)),
              ("decode.reparam.logit_fn.weight", (16, 1, 5, 5)), ("decode.reparam.logit_fn.bias", (1,))]
        return s
    if c.kind in ("resconv", "auxresconv"):
        # weight-normalised residual blocks (models/layers2.py:50-93,237-265,305-352; models/layers.py:25-85 for encode.fc);
        # every block owns three weight-normalised operators, parameters in the order direction, scale, bias
        def wn(prefix, out, inn, conv):
            return [(prefix + "direction", (out, inn, 3, 3) if conv else (out, inn)), (prefix + "scale", (out,)), (prefix + "bias", (out,))]
        def block(prefix, out, inn, conv):
            a, b_, c_ = ("conv_0h.", "conv_h1.", "conv_01.") if conv else ("dot_0h.", "dot_h1.", "dot_01.")
            return wn(prefix + a, out, inn, conv) + wn(prefix + b_, out, out, conv) + wn(prefix + c_, out, inn, conv)
        cdim = 512 if c.kind == "resconv" else c.h_dim
        tp = "encode.inp_encode." if c.kind == "resconv" else "encode.inp_encode.enc."
        s = []
        for i, (o, inn) in zip((0, 2, 4, 6, 8), ((16, 1), (16, 16), (32, 16), (32, 32), (32, 32))):
            s += block(f"{tp}{i}.", o, inn, True)
        s += block(f"{tp}11.", cdim, 512, False)
        if c.kind == "resconv":     # encode.fc (models/ivae/resconv.py:101-116)
            s += _resconv_head_spec(c.enc_type, c.n_layers, cdim + c.noise_dim, c.h_dim, c.z_dim)
        else:
            s += [("encode.aux_encode.reparam.mean_fn.weight", (c.noise_dim, cdim)), ("encode.aux_encode.reparam.mean_fn.bias", (c.noise_dim,)),
                  ("encode.aux_encode.reparam.logvar_fn.weight", (c.noise_dim, cdim)), ("encode.aux_encode.reparam.logvar_fn.bias", (c.noise_dim,)),
                  ("encode.encode.fc.0.weight", (cdim, cdim + c.noise_dim)), ("encode.encode.fc.0.bias", (cdim,)),
                  ("encode.encode.reparam.mean_fn.weight", (c.z_dim, cdim)), ("encode.encode.reparam.mean_fn.bias", (c.z_dim,)),
                  ("encode.encode.reparam.logvar_fn.weight", (c.z_dim, cdim)), ("encode.encode.reparam.logvar_fn.bias", (c.z_dim,))]
        s += block("decode.dec.0.", cdim, c.z_dim, False) + block("decode.dec.2.", 512, cdim, False)
        for i, (o, inn) in zip((6, 8, 12, 14, 17), ((32, 32), (32, 32), (16, 32), (16, 16), (1, 16))):
            s += block(f"decode.dec.{i}.", o, inn, True)
        return s
    if c.kind == "toy":
        s = _mlp_spec("encode.inp_encode.", c.input_dim, c.h_dim, c.h_dim, c.n_layers - 1)
        s += _ctxcat_mlp_spec("encode.fc.", c.h_dim, c.noise_dim, c.h_dim, c.z_dim, c.n_layers)
        s += _mlp_spec("decode.main.", c.z_dim, c.h_dim, c.h_dim, c.n_layers - 1)
        s += [("decode.reparam.mean_fn.weight", (c.input_dim, c.h_dim)),
              ("decode.reparam.mean_fn.bias", (c.input_dim,)),
              ("decode.reparam.logvar_fn.weight", (c.input_dim, c.h_dim)),
              ("decode.reparam.logvar_fn.bias", (c.input_dim,))]
        return s
    raise NotImplementedError(c.kind)


def cdae_param_spec(c: CdaeCfg):
    s = _mlp_spec("ctx_encode.", c.context_dim, c.h_dim, c.h_dim, c.n_layers - 1)
    s += _mlp_spec("inp_encode.", c.input_dim, c.h_dim, c.h_dim, c.n_layers - 1)
    if c.kind == "grad":
        s += _mlp_spec("neglogprob.", 2 * c.h_dim + 1, c.h_dim, 1, c.n_layers)
    elif c.kind == "res":
        s += _mlp_spec("dae.", 2 * c.h_dim + 1, c.h_dim, c.input_dim, c.n_layers)
    else:
        raise NotImplementedError(c.kind)
    return s


def init_params(spec, seed, special=None, dtype=torch.float32):
    """Deterministic, platform-independent initialiser (numpy PCG64).

    Distribution families follow the reference (nn.Linear default: U(+-1/sqrt(fan_in)) for
    both weight (kaiming_uniform a=sqrt5) and bias); `special` maps a name to
    ("normal",) | ("xavier",) | ("zeros",) for the overrides listed in SURVEY App. B.
    It is NOT bit-identical to torch's own init RNG - parity tests load the same numbers
    into the reference, so only the family matters.
    """
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    special = special or {}
    out = {}
    for name, shape in spec:
        fan_in = shape[1] if len(shape) == 2 else (shape[1] * shape[2] * shape[3] if len(shape) == 4 else None)
        kind = special.get(name, ("default",))[0]
        if name.endswith(".scale"):          # weight-norm gains (layers2.py:66-71: 1 at construction); spread here so that gradients tell them apart
            a = rng.uniform(0.7, 1.3, shape)
        elif name.endswith(".bias") and (name[:-len("bias")] + "direction") in dict(spec):
            wshape = dict(spec)[name[:-len("bias")] + "direction"]
            bound = 1.0 / math.sqrt(wshape[1])                             # layers2.py:66 / 184-190: stdv = 1/sqrt(in_features | in_channels)
            a = rng.uniform(-bound, bound, shape)
        elif name.endswith(".direction"):
            bound = 1.0 / math.sqrt(shape[1])
            a = rng.uniform(-bound, bound, shape)
        elif kind == "normal":
            a = rng.standard_normal(shape)
        elif kind == "zeros":
            a = np.zeros(shape)
        elif kind == "xavier":
            rf = shape[2] * shape[3] if len(shape) == 4 else 1
            bound = math.sqrt(6.0 / ((shape[0] + shape[1]) * rf))
            a = rng.uniform(-bound, bound, shape)
        else:
            if fan_in is None:  # bias: bound uses the fan_in of the matching weight
                wshape = dict(spec)[name[:-len("bias")] + "weight"]
                fan_in = wshape[1] if len(wshape) == 2 else wshape[1] * wshape[2] * wshape[3]
            bound = 1.0 / math.sqrt(fan_in)
            a = rng.uniform(-bound, bound, shape)
        out[name] = torch.tensor(a, dtype=dtype)
    return out


def model_init_special(c: ModelCfg):
    """Init overrides of the reference constructors."""
    sp = {}
    spec = model_param_spec(c)
    if c.kind in ("conv", "auxconv"):   # self.apply(weight_init): xavier-uniform on every Conv2d/Linear (NOT ConvTranspose2d), zero biases
        for name, shape in spec:
            if "deconv" in name or "logit_fn" in name:
                continue
            sp[name] = ("xavier",) if name.endswith("weight") else ("zeros",)
        return sp
    if c.kind in ("resconv", "auxresconv"):   # no init override (the WN operators' and nn.Linear's own reset_parameters)
        return {}
    if c.kind == "auxtoy":     # init='gaussian' reaches the toy Decoder only (ivae/auxtoy.py:165, models/vae/toy.py)
        return {"decode.reparam.mean_fn.weight": ("normal",)}
    if c.kind == "auxmnist":   # do_xavier=True: self.apply(weight_init) on the whole model (ivae/auxmnist.py:172-174)
        return {name: (("xavier",) if name.endswith("weight") else ("zeros",)) for name, _ in spec}
    for name, _ in spec:
        if name.startswith("decode.") and c.kind == "mnist":
            sp[name] = ("xavier",) if name.endswith("weight") else ("zeros",)
    sp["encode.fc.fc.weight"] = ("normal",)          # init='gaussian' (ivae/mnist.py:158-159)
    if c.kind == "toy":
        sp["decode.reparam.mean_fn.weight"] = ("normal",)   # ivae/toy.py:719-720
    return sp


# --------------------------------------------------------------------------- #
# building blocks
# --------------------------------------------------------------------------- #
def act(name):
    if name == "softplus":
        return F.softplus            # beta=1, threshold=20
    if name == "relu":
        return F.relu
    if name == "tanh":
        return torch.tanh
    if name == "elu":
        return F.elu
    if name == "csoftplus":
        return lambda x: torch.log(torch.exp(x) + 1)
    if name == "leaky_relu":
        return lambda x: F.leaky_relu(x, negative_slope=0.2)
    if name == "swish":
        return lambda x: x * torch.sigmoid(x)
    raise NotImplementedError(name)


def mlp(p, prefix, x, n_hidden, nonlin, act_out):
    f = act(nonlin)
    h = x
    for i in range(n_hidden):
        h = f(F.linear(h, p[f"{prefix}layers.{i}.weight"], p[f"{prefix}layers.{i}.bias"]))
    y = F.linear(h, p[f"{prefix}fc.weight"], p[f"{prefix}fc.bias"])
    return f(y) if act_out else y


def ctxcat_mlp(p, prefix, x, ctx, n_hidden, nonlin):
    f = act(nonlin)
    h = x
    for i in range(n_hidden):
        h = f(F.linear(torch.cat([h, ctx], 1), p[f"{prefix}layers.{i}.weight"], p[f"{prefix}layers.{i}.bias"]))
    return F.linear(torch.cat([h, ctx], 1), p[f"{prefix}fc.weight"], p[f"{prefix}fc.bias"])


def _resconv_head_spec(enc_type, n_layers, cin, h_dim, z_dim):
    """`encode.fc` of ResConvIPVAE in registration order (models/ivae/resconv.py:101-116): MLP (models/layers.py:477-515), ResMLP of ResLinear
    blocks with WeightNormalizedLinear(norm=False) or nn.Linear operators (:25-85,559-622; no dot_01 when a block keeps its width), or
    Sequential(ResMLP(..., n_layers - 1 hidden, output h_dim, activated), Linear(h_dim, z_dim))."""
    assert enc_type in ("mlp", "res-wn-mlp", "res-mlp", "res-wn-mlp-lin", "res-mlp-lin")
    lin = lambda pre, out, inn: [(pre + "weight", (out, inn)), (pre + "bias", (out,))]
    oper = lambda pre, out, inn, wn: ([(pre + "direction", (out, inn)), (pre + "scale", (out,)), (pre + "bias", (out,))] if wn else lin(pre, out, inn))
    res = lambda pre, out, inn, wn: (oper(pre + "dot_0h.", out, inn, wn) + oper(pre + "dot_h1.", out, out, wn)
                                     + ([] if inn == out else oper(pre + "dot_01.", out, inn, wn)))
    s = []
    if enc_type == "mlp":
        for i in range(n_layers):
            s += lin(f"encode.fc.layers.{i}.", h_dim, cin if i == 0 else h_dim)
        return s + lin("encode.fc.fc.", z_dim, h_dim)
    wn = enc_type.startswith("res-wn")
    if enc_type in ("res-wn-mlp", "res-mlp"):
        for i in range(n_layers):
            s += res(f"encode.fc.layers.{i}.", h_dim, cin if i == 0 else h_dim, wn)
        return s + res("encode.fc.fc.", z_dim, h_dim, wn)
    for i in range(n_layers - 1):
        s += res(f"encode.fc.0.layers.{i}.", h_dim, cin if i == 0 else h_dim, wn)
    return s + res("encode.fc.0.fc.", h_dim, cin if n_layers == 1 else h_dim, wn) + lin("encode.fc.1.", z_dim, h_dim)


def _head_oper(p, pre, x, wn):
    """An operator of ResMLP: WeightNormalizedLinear(norm=False) (weight = scale * direction, models/layers.py:47-53) or nn.Linear."""
    return F.linear(x, _wn_weight(p, pre, False) if wn else p[pre + "weight"], p[pre + "bias"])


def _head_res_linear(p, pre, x, wn):
    """ResLinear (models/layers.py:66-85): dot_h1(relu(dot_0h(x))) + (x if same_dim else dot_01(x))."""
    skip = _head_oper(p, pre + "dot_01.", x, wn) if (pre + "dot_01.bias") in p else x
    return _head_oper(p, pre + "dot_h1.", F.relu(_head_oper(p, pre + "dot_0h.", x, wn)), wn) + skip


def resconv_head(c, p, hin):
    """ResConvIPVAE's `encode.fc` on hin = [trunk output | noise] (models/ivae/resconv.py:101-116,150-157), every enc_type; ELU between
    the operators (get_nonlinear_func(nonlinearity), models/layers.py:510,617)."""
    et, nl = c.enc_type, c.n_layers
    if et == "mlp":
        h = hin
        for i in range(nl):
            h = F.elu(F.linear(h, p[f"encode.fc.layers.{i}.weight"], p[f"encode.fc.layers.{i}.bias"]))
        return F.linear(h, p["encode.fc.fc.weight"], p["encode.fc.fc.bias"])
    wn = et.startswith("res-wn")
    if et in ("res-wn-mlp", "res-mlp"):
        h = hin
        for i in range(nl):
            h = F.elu(_head_res_linear(p, f"encode.fc.layers.{i}.", h, wn))
        return _head_res_linear(p, "encode.fc.fc.", h, wn)
    h = hin
    for i in range(nl - 1):
        h = F.elu(_head_res_linear(p, f"encode.fc.0.layers.{i}.", h, wn))
    h = F.elu(_head_res_linear(p, "encode.fc.0.fc.", h, wn))
    return F.linear(h, p["encode.fc.1.weight"], p["encode.fc.1.bias"])


def _wn_weight(p, pre, norm):
    """Effective weight of a weight-normalised operator: scale * direction / ||direction||  (norm over everything but the output
    index; layers2.py:73-83,255-265) or scale * direction (models/layers.py:47-53 with norm=False, as ResMLP builds them)."""
    d, sc = p[pre + "direction"], p[pre + "scale"]
    view = (-1,) + (1,) * (d.dim() - 1)
    if norm:
        d = d / d.pow(2).flatten(1).sum(1).sqrt().view(view)
    return sc.view(view) * d


def res_conv(p, pre, x, stride, f):
    """layers2.ResConv2d (3x3, padding 1; conv_h1 is 3x3 stride 1): conv_h1(f(conv_0h(x))) + conv_01(x)   (layers2.py:305-328)."""
    h = f(F.conv2d(x, _wn_weight(p, pre + "conv_0h.", True), p[pre + "conv_0h.bias"], stride, 1))
    return (F.conv2d(h, _wn_weight(p, pre + "conv_h1.", True), p[pre + "conv_h1.bias"], 1, 1)
            + F.conv2d(x, _wn_weight(p, pre + "conv_01.", True), p[pre + "conv_01.bias"], stride, 1))


def res_linear(p, pre, x, norm):
    """ResLinear with its DEFAULT inner activation ReLU (layers2.py:331-352 with WNlinear, norm=True; models/layers.py:66-85 with
    WeightNormalizedLinear norm=False inside ResMLP) and a projected skip (same_dim is False in every block these models build)."""
    h = F.relu(F.linear(x, _wn_weight(p, pre + "dot_0h.", norm), p[pre + "dot_0h.bias"]))
    return (F.linear(h, _wn_weight(p, pre + "dot_h1.", norm), p[pre + "dot_h1.bias"])
            + F.linear(x, _wn_weight(p, pre + "dot_01.", norm), p[pre + "dot_01.bias"]))


def resconv_trunk(c, p, x):
    """The per-image trunk both families share (ivae/resconv.py:81-96, vae/auxresconv.py:36-63; do_center=True):
    28 -> 14 -> 14 -> 7 -> 7 -> 4, flatten (NCHW), ResLinear 512 -> c_dim, ELU after every block."""
    tp = "encode.inp_encode." if c.kind == "resconv" else "encode.inp_encode.enc."
    h = x.reshape(x.size(0), 784)
    h = (2 * h - 1 if c.do_center else h).view(-1, 1, 28, 28)
    for i, st in zip((0, 2, 4, 6, 8), (2, 1, 2, 1, 2)):
        h = F.elu(res_conv(p, f"{tp}{i}.", h, st, F.elu))
    return F.elu(res_linear(p, f"{tp}11.", h.reshape(h.size(0), 512), True))


def clip_logvar(name, lv):
    """NormalDistribution.clip_logvar (models/reparam.py:17-41; MIN_LOGVAR = -4, MAX_LOGVAR = 2 :7-8)."""
    if name in (None, "none"):
        return lv
    if name == "hard":
        return torch.min(torch.max(lv, -4. * torch.ones_like(lv)), 2. * torch.ones_like(lv))
    if name == "softplus":
        return F.softplus(lv)
    if name.startswith("spm"):
        k = float(name[3:])
        return F.softplus(lv + k) - k
    if name == "tanh":
        return torch.tanh(lv)
    if name == "2tanh":
        return 2.0 * torch.tanh(lv)
    raise ValueError(name)


def spm4(lv):
    """NormalDistribution.clip_logvar with nonlinearity='spm4' (models/reparam.py:30-31)."""
    return F.softplus(lv + 4.) - 4.


def auxres_encode(c, p, x, noise, nz):
    """MNISTResConvAuxIPVAE's sampler (ivae/auxresconv.py:73-106): heads on the shared trunk, both log-variances clipped 'spm4';
    noise = (eps0 [B*nz, z0_dim], eps [B*nz, z_dim]) already scaled by std.
    c.clipped (MNISTResConvAuxIPVAEClipped, ivae/auxresconv2.py:71-72,91): plain log-variances and z0 = mu0 + (std exp(lv0 / 2) + 1) eps0 - the
    `+ 1 eps0` term needs the UNSCALED draw: noise = (std eps0, std eps, eps0); with two entries the draws are taken as unscaled (std = 1)."""
    eps0, eps = noise[0], noise[1]
    clip = (lambda t: t) if c.clipped else spm4
    inp = resconv_trunk(c, p, x)
    mu0 = F.linear(inp, p["encode.aux_encode.reparam.mean_fn.weight"], p["encode.aux_encode.reparam.mean_fn.bias"])
    lv0 = clip(F.linear(inp, p["encode.aux_encode.reparam.logvar_fn.weight"], p["encode.aux_encode.reparam.logvar_fn.bias"]))
    z0 = expand_rows(mu0, nz) + torch.exp(0.5 * expand_rows(lv0, nz)) * eps0
    if c.clipped:
        z0 = z0 + (noise[2] if len(noise) > 2 else eps0)
    h = F.elu(F.linear(torch.cat([expand_rows(inp, nz), z0], 1), p["encode.encode.fc.0.weight"], p["encode.encode.fc.0.bias"]))
    mu = F.linear(h, p["encode.encode.reparam.mean_fn.weight"], p["encode.encode.reparam.mean_fn.bias"])
    lv = clip(F.linear(h, p["encode.encode.reparam.logvar_fn.weight"], p["encode.encode.reparam.logvar_fn.bias"]))
    z = mu + torch.exp(0.5 * lv) * eps
    return {"z": z, "h": h, "z0": z0, "inp": inp}


def resconv_decode(c, p, z):
    """models/vae/resconv.py:79-136: two ResLinears, then bilinear x2 upsampling (align_corners=True) between ResConv2d blocks,
    4 -> 8 -> (slice) 7 -> 14 -> 28, logits [R, 784]."""
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)
    h = F.elu(res_linear(p, "decode.dec.0.", z, True))
    h = F.elu(res_linear(p, "decode.dec.2.", h, True)).view(-1, 32, 4, 4)
    h = up(h)
    h = F.elu(res_conv(p, "decode.dec.6.", h, 1, F.elu))
    h = F.elu(res_conv(p, "decode.dec.8.", h, 1, F.elu))[:, :, :-1, :-1]
    h = up(h)
    h = F.elu(res_conv(p, "decode.dec.12.", h, 1, F.elu))
    h = F.elu(res_conv(p, "decode.dec.14.", h, 1, F.elu))
    h = up(h)
    return res_conv(p, "decode.dec.17.", h, 1, F.elu).reshape(z.size(0), 784)


def expand_rows(t, nz):
    """[B, d] -> [B*nz, d], image-major (utils/msc.py:21-40)."""
    return t.unsqueeze(1).expand(-1, nz, -1).reshape(t.size(0) * nz, -1)


# --------------------------------------------------------------------------- #
# implicit-posterior VAE
# --------------------------------------------------------------------------- #
def encode(c: ModelCfg, p, x, noise, nz):
    """z = f(x, noise): x [B, input_dim], noise [B*nz, noise_dim] (already scaled by std)."""
    B = x.size(0)
    x = x.reshape(B, c.input_dim)
    if c.kind == "mnist":
        inp = mlp(p, "encode.inp_encode.", 2 * x - 1, c.n_layers + 1, c.nonlin, True)
        hin = torch.cat([expand_rows(inp, nz), noise], 1)
        z = mlp(p, "encode.fc.", hin, 1, c.nonlin, False)
    elif c.kind == "conv":
        f = act(c.nonlin)
        h = (2 * x - 1).view(B, 1, 28, 28)
        for i in (1, 2, 3):
            h = f(F.conv2d(h, p[f"encode.conv{i}.weight"], p[f"encode.conv{i}.bias"], stride=2, padding=2))
        hin = torch.cat([expand_rows(h.reshape(B, -1), nz), noise], 1)
        z = F.linear(f(F.linear(hin, p["encode.fc4.weight"], p["encode.fc4.bias"])), p["encode.fc5.weight"], p["encode.fc5.bias"])
    elif c.kind == "toy":
        inp = mlp(p, "encode.inp_encode.", x, c.n_layers - 1, c.nonlin, True)
        z = ctxcat_mlp(p, "encode.fc.", expand_rows(inp, nz), noise, c.n_layers, c.nonlin)
    elif c.kind in ("auxmnist", "auxconv"):
        z = aux_encode(c, p, x, noise, nz)["z"]
    elif c.kind == "auxtoy":      # the model-level entry points take nz = q^2 (ivae/auxtoy.py:215,230)
        q = math.isqrt(nz)
        assert q * q == nz, "ToyAuxIPVAE samples q z0's x q z's per image: nz must be a square"
        z = aux_encode(c, p, x, noise, q)["z"]
    elif c.kind == "resconv":      # ivae/resconv.py:141-159
        inp = resconv_trunk(c, p, x)
        z = resconv_head(c, p, torch.cat([expand_rows(inp, nz), noise], 1))
    elif c.kind == "auxresconv":
        z = auxres_encode(c, p, x, noise, nz)["z"]
    else:
        raise NotImplementedError
    return z.view(B, nz, c.z_dim)


def aux_encode(c: ModelCfg, p, x, noise, nz):
    """Hierarchical sampler of the aux models (ivae/auxmnist.py:75-108): noise = (eps0 [B*nz, noise_dim], eps [B*nz, z_dim]), both
    already scaled by std (std = 0 -> zeros: z0 = mu0(x), z = mu(x, z0)).
      h0 = MLP(2x-1);  (mu0, lv0) = heads(h0);  z0 = mu0[b] + exp(lv0[b]/2) eps0         (per image -> per sample)
      h  = MLP(cat[2x-1, z0]);  (mu, lv) = heads(h);  z = mu + exp(lv/2) eps
    (--model-clip-z0-logvar / --model-clip-z-logvar are 'none' in the shipped recipes - argparse offers nothing else, ivae_ardae.py:67-72 -;
    the classes' own clip_z0_logvar / clip_z_logvar arguments take every choice of NormalDistribution.clip_logvar: c.clip_z0 / c.clip_z.)"""
    eps0, eps = noise
    B = x.size(0)
    if c.kind == "auxconv":   # models/vae/auxconv.py:60-81,115-140: conv trunks in place of the MLPs, fc width 800
        f = act(c.nonlin)
        def trunk(prefix):
            hh = (2 * x.reshape(B, 784) - 1).view(B, 1, 28, 28)
            for i in (1, 2, 3):
                hh = f(F.conv2d(hh, p[f"{prefix}conv{i}.weight"], p[f"{prefix}conv{i}.bias"], stride=2, padding=2))
            return hh.reshape(B, -1)
        def heads(prefix, hh):
            return (F.linear(hh, p[prefix + "reparam.mean_fn.weight"], p[prefix + "reparam.mean_fn.bias"]),
                    F.linear(hh, p[prefix + "reparam.logvar_fn.weight"], p[prefix + "reparam.logvar_fn.bias"]))
        h0 = f(F.linear(trunk("encode.aux_encode."), p["encode.aux_encode.fc.weight"], p["encode.aux_encode.fc.bias"]))
        mu0, lv0 = heads("encode.aux_encode.", h0)
        z0 = expand_rows(mu0, nz) + torch.exp(0.5 * expand_rows(lv0, nz)) * eps0
        h3 = trunk("encode.encode.")
        h = f(F.linear(torch.cat([expand_rows(h3, nz), z0], 1), p["encode.encode.fc.weight"], p["encode.encode.fc.bias"]))
        mu, lv = heads("encode.encode.", h)
        z = mu + torch.exp(0.5 * lv) * eps
        return {"z": z, "h0": h0, "h": h, "z0": z0, "mu0": mu0, "lv0": lv0, "mu": mu, "lv": lv}
    if c.kind == "auxtoy":
        # ToyAuxIPVAE (ivae/auxtoy.py:74-103): NO rescale of x, and a SQUARE sampling scheme - `nz` here is Encoder._forward's nz = q:
        # q z0's per image (eps0 [B q, noise_dim]) and q z's per z0 (eps [B q q, z_dim]); the model-level calls pass q = int(sqrt(nz))
        # (:215,230), logprob passes q = sample_size (:313).  Returns z [B q q, z]; h0 [B, h], h / mu / lv [B q, .]
        q = nz
        xs = x.reshape(B, c.input_dim)
        h0 = mlp(p, "encode.aux_encode.main.", xs, c.n_layers - 1, c.nonlin, True)
        mu0 = F.linear(h0, p["encode.aux_encode.reparam.mean_fn.weight"], p["encode.aux_encode.reparam.mean_fn.bias"])
        lv0 = clip_logvar(c.clip_z0, F.linear(h0, p["encode.aux_encode.reparam.logvar_fn.weight"], p["encode.aux_encode.reparam.logvar_fn.bias"]))
        z0 = expand_rows(mu0, q) + torch.exp(0.5 * expand_rows(lv0, q)) * eps0
        h = mlp(p, "encode.encode.fc.", torch.cat([expand_rows(xs, q), z0], 1), c.n_layers - 1, c.nonlin, True)
        mu = F.linear(h, p["encode.encode.reparam.mean_fn.weight"], p["encode.encode.reparam.mean_fn.bias"])
        lv = clip_logvar(c.clip_z, F.linear(h, p["encode.encode.reparam.logvar_fn.weight"], p["encode.encode.reparam.logvar_fn.bias"]))
        z = expand_rows(mu, q) + torch.exp(0.5 * expand_rows(lv, q)) * eps
        return {"z": z, "h0": h0, "h": h, "z0": z0, "mu0": mu0, "lv0": lv0, "mu": mu, "lv": lv}
    xs = 2 * x.reshape(B, c.input_dim) - 1
    h0 = mlp(p, "encode.aux_encode.main.", xs, c.n_layers - 1, c.nonlin, True)
    mu0 = F.linear(h0, p["encode.aux_encode.reparam.mean_fn.weight"], p["encode.aux_encode.reparam.mean_fn.bias"])
    lv0 = clip_logvar(c.clip_z0, F.linear(h0, p["encode.aux_encode.reparam.logvar_fn.weight"], p["encode.aux_encode.reparam.logvar_fn.bias"]))
    z0 = expand_rows(mu0, nz) + torch.exp(0.5 * expand_rows(lv0, nz)) * eps0
    h = mlp(p, "encode.encode.fc.", torch.cat([expand_rows(xs, nz), z0], 1), c.n_layers - 1, c.nonlin, True)
    mu = F.linear(h, p["encode.encode.reparam.mean_fn.weight"], p["encode.encode.reparam.mean_fn.bias"])
    lv = clip_logvar(c.clip_z, F.linear(h, p["encode.encode.reparam.logvar_fn.weight"], p["encode.encode.reparam.logvar_fn.bias"]))
    z = mu + torch.exp(0.5 * lv) * eps
    return {"z": z, "h0": h0, "h": h, "z0": z0, "mu0": mu0, "lv0": lv0, "mu": mu, "lv": lv}


def zero_noise(c: ModelCfg, rows, like, raw0=None):
    """The draws of an encode(x, std=0) call, multiplied by 0.  raw0 [rows, noise_dim]: the clipped aux-resconv class keeps an unscaled
    eps0 at std = 0 (ivae/auxresconv2.py:91) - `encode(x, std=0)` is a RANDOM draw there."""
    if c.kind in AUX_KINDS:
        if c.clipped:
            return (like.new_zeros(rows, c.noise_dim), like.new_zeros(rows, c.z_dim), like.new_zeros(rows, c.noise_dim) if raw0 is None else raw0)
        return (like.new_zeros(rows, c.noise_dim), like.new_zeros(rows, c.z_dim))
    return like.new_zeros(rows, c.noise_dim)


def cdae_context(c: ModelCfg, tc, p, x, raw0=None):
    """--cdae-ctx-type (ivae_ardae.py:729-741): data -> the (centred) image [B, D]; lt0 -> encode(x, std=0) [B, z]; hidden1a -> cat(h0, h) of the
    std=0 pass [B, 2h]."""
    B = x.size(0)
    if tc.ctx_type == "data":           # the flattened image itself (ivae_ardae.py:730-734,809-813)
        xf = x.reshape(B, -1)
        return 2 * xf - 1 if tc.ctx_center else xf
    if tc.ctx_type == "lt0":
        return encode(c, p, x, zero_noise(c, B, x), 1).reshape(B, c.z_dim)
    if tc.ctx_type == "hidden1a":
        assert c.kind in AUX_KINDS, "hidden1a is the aux models' context"
        if c.kind == "auxresconv":      # Encoder.forward_hidden returns h alone: 450 columns (ivae/auxresconv.py:125-132, ivae_ardae.py:578-579)
            return auxres_encode(c, p, x, zero_noise(c, B, x, raw0), 1)["h"]
        a = aux_encode(c, p, x, zero_noise(c, B, x), 1)
        return torch.cat([a["h0"], a["h"]], 1)
    raise NotImplementedError(tc.ctx_type)


def decode(c: ModelCfg, p, z):
    """Returns the decoder's distribution parameters for z [R, z_dim]."""
    if c.kind in ("mnist", "auxmnist"):
        h = mlp(p, "decode.main.", z, c.n_layers if c.kind == "mnist" else c.n_layers - 1, c.nonlin, True)
        return (F.linear(h, p["decode.reparam.logit_fn.weight"], p["decode.reparam.logit_fn.bias"]),)
    if c.kind in ("conv", "auxconv"):
        f = act(c.nonlin)
        h = mlp(p, "decode.fc.", z, 1, c.nonlin, True).view(-1, 32, 4, 4)
        h = F.pad(f(F.conv_transpose2d(h, p["decode.deconv1.weight"], p["decode.deconv1.bias"], stride=2, padding=2)), (0, 1, 0, 1))
        h = f(F.conv_transpose2d(h, p["decode.deconv2.weight"], p["decode.deconv2.bias"], stride=2, padding=2))
        logit = F.conv_transpose2d(h, p["decode.reparam.logit_fn.weight"], p["decode.reparam.logit_fn.bias"], stride=2, padding=2)
        return (logit[:, :, :28, :28].reshape(z.size(0), 784),)
    if c.kind in ("resconv", "auxresconv"):
        return (resconv_decode(c, p, z),)
    h = mlp(p, "decode.main.", z, c.n_layers - 1, c.nonlin, True)
    mu = F.linear(h, p["decode.reparam.mean_fn.weight"], p["decode.reparam.mean_fn.bias"])
    logvar = F.linear(h, p["decode.reparam.logvar_fn.weight"], p["decode.reparam.logvar_fn.bias"])
    return mu, logvar


def recon_rows(c: ModelCfg, dist, target):
    if c.kind in ("mnist", "conv", "auxmnist", "auxconv", "resconv", "auxresconv"):
        (logit,) = dist
        return F.binary_cross_entropy_with_logits(logit, target, reduction="none").sum(1)
    mu, logvar = dist
    return 0.5 * (logvar + (target - mu) ** 2 / logvar.exp() + LOG2PI).sum(1)


def prior_rows(z):
    return (0.5 * (z ** 2 + LOG2PI)).sum(1)


def vae_forward(c: ModelCfg, p, x, noise, beta, nz=1):
    """-> (z [B,nz,z], loss, recon.mean, prior.mean); loss carries grad_fn."""
    B = x.size(0)
    x = x.reshape(B, c.input_dim)
    z = encode(c, p, x, noise, nz)
    zf = z.reshape(B * nz, c.z_dim)
    dist = decode(c, p, zf)
    rec = recon_rows(c, dist, expand_rows(x, nz))
    pri = prior_rows(zf)
    loss = (rec + beta * pri).mean()
    return z, loss, rec.mean().detach(), pri.mean().detach(), dist


# --------------------------------------------------------------------------- #
# conditional AR-DAE
# --------------------------------------------------------------------------- #
def _cdae_h(c: CdaeCfg, p, x_bar, ctx_rows, std_rows):
    ctx = mlp(p, "ctx_encode.", ctx_rows, c.n_layers - 1, c.nonlin, True)
    inp = mlp(p, "inp_encode.", x_bar, c.n_layers - 1, c.nonlin, True)
    return torch.cat([inp, ctx, std_rows], 1)


def cdae_score(c: CdaeCfg, p, x_bar, ctx_rows, std_rows, create_graph):
    """Score estimate at x_bar [N, z] (x_bar must require grad for kind='grad')."""
    h = _cdae_h(c, p, x_bar, ctx_rows, std_rows)
    if c.kind == "grad":
        logprob = (-mlp(p, "neglogprob.", h, c.n_layers, c.nonlin, False)).sum()
        return torch.autograd.grad(logprob, x_bar, retain_graph=True, create_graph=create_graph)[0]
    return mlp(p, "dae.", h, c.n_layers, c.nonlin, False)


def cdae_forward(c: CdaeCfg, p, inp, context, std, eps):
    """DAE loss.  inp [B,S,z], context [B,1,c], std [B,S,1], eps [B*S,z] -> scalar loss."""
    B, S, _ = inp.shape
    x = inp.reshape(B * S, c.input_dim)
    ctx_rows = expand_rows(context.reshape(B, c.context_dim), S)      # reference runs ctx_encode on N rows
    std_rows = std.reshape(B * S, 1)
    x_bar = (x + std_rows * eps).detach().requires_grad_(True)
    g = cdae_score(c, p, x_bar, ctx_rows, std_rows, create_graph=True)
    return F.mse_loss(std_rows * g, -eps)


def cdae_glogprob(c: CdaeCfg, p, inp, context, std):
    B, S, _ = inp.shape
    x = inp.reshape(B * S, c.input_dim).detach().requires_grad_(True)
    ctx_rows = expand_rows(context.reshape(B, c.context_dim), S)
    g = cdae_score(c, p, x, ctx_rows, std.reshape(B * S, 1), create_graph=False)
    return g.detach().view(B, S, c.input_dim)


def latent_stats(latent, latent_mean, std_scale, delta):
    """u = s(z - z0); std = delta * mean_d(std_nz(u)) (unbiased)  (ivae_ardae.py:753-755)."""
    u = std_scale * (latent - latent_mean)
    std_qz = torch.std(u, dim=1, keepdim=True)
    std = delta * torch.mean(std_qz, dim=2, keepdim=True)
    return u, std


# --------------------------------------------------------------------------- #
# optimisers
# --------------------------------------------------------------------------- #
def adam_ref_step(params, grads, state, lr, beta1, beta2=0.999, eps=1e-8, amsgrad=False):
    """Vendored old-style Adam: eps added BEFORE the bias correction (utils/optim.py:102-106); amsgrad: the running maximum of
    exp_avg_sq normalises (:96-100).  Tensors whose grad is None are skipped and keep no state."""
    for name, w in params.items():
        g = grads.get(name)
        if g is None:
            continue
        st = state.setdefault(name, {"step": 0, "exp_avg": torch.zeros_like(w), "exp_avg_sq": torch.zeros_like(w)})
        st["step"] += 1
        bc1 = 1 - beta1 ** st["step"]
        bc2 = 1 - beta2 ** st["step"]
        st["exp_avg"].mul_(beta1).add_(g, alpha=1 - beta1)
        st["exp_avg_sq"].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        v = st["exp_avg_sq"]
        if amsgrad:
            v = st["max_exp_avg_sq"] = torch.maximum(st.get("max_exp_avg_sq", torch.zeros_like(w)), st["exp_avg_sq"])
        denom = (v.sqrt() + eps) / math.sqrt(bc2)
        w.addcdiv_(st["exp_avg"], denom, value=-lr / bc1)


def rmsprop_step(params, grads, state, lr, momentum, alpha=0.99, eps=1e-8):
    """torch.optim.RMSprop (not centred) with momentum; grad None -> skipped."""
    for name, w in params.items():
        g = grads.get(name)
        if g is None:
            continue
        st = state.setdefault(name, {"step": 0, "square_avg": torch.zeros_like(w), "momentum_buffer": torch.zeros_like(w)})
        st["step"] += 1
        st["square_avg"].mul_(alpha).addcmul_(g, g, value=1 - alpha)
        avg = st["square_avg"].sqrt().add_(eps)
        if momentum > 0:
            st["momentum_buffer"].mul_(momentum).addcdiv_(g, avg)
            w.add_(st["momentum_buffer"], alpha=-lr)
        else:
            w.addcdiv_(g, avg, value=-lr)


def sgd_step(params, grads, lr):
    """torch.optim.SGD(lr) as ivae_ardae.py:546-547,613-614 construct it; grad None -> skipped."""
    for name, w in params.items():
        if grads.get(name) is not None:
            w.add_(grads[name], alpha=-lr)


def optimizer_step(kind, params, grads, state, lr, beta1, momentum):
    """--m-optimizer / --d-optimizer (ivae_ardae.py:545-556,612-622): sgd | adam | amsgrad | rmsprop."""
    if kind == "sgd":
        sgd_step(params, grads, lr)
    elif kind in ("adam", "amsgrad"):
        adam_ref_step(params, grads, state, lr, beta1, amsgrad=kind == "amsgrad")
    elif kind == "rmsprop":
        rmsprop_step(params, grads, state, lr, momentum)
    else:
        raise NotImplementedError(kind)


# --------------------------------------------------------------------------- #
# noise (reference draw order)
# --------------------------------------------------------------------------- #
def draw_step_noise(mc: ModelCfg, tc: TrainCfg, B, gen):
    """All Gaussian/uniform draws one train step consumes, in the reference's order.
    The std=0 encodes consume a draw that is multiplied by 0, so they are skipped here
    (they only advance the reference's RNG stream)."""
    N = B * tc.nz_cdae
    aux = mc.kind in AUX_KINDS       # a second draw per sampler call: eps of z = mu + exp(lv/2) eps (ivae/auxmnist.py:113-114)
    sq = (lambda rows_per_image: math.isqrt(rows_per_image)) if mc.kind == "auxtoy" else (lambda rows_per_image: rows_per_image)   # eps0 rows per image
    n = {"sampler": torch.randn(B * sq(tc.nz_cdae), mc.noise_dim, generator=gen)}          # forward_hidden
    if aux:
        n["sampler_z"] = torch.randn(N, mc.z_dim, generator=gen)
    if aux and mc.clipped:       # the two std = 0 calls that open the cDAE update keep an unscaled eps0 each (ivae/auxresconv2.py:91)
        n["ctx_raw"] = torch.randn(B, mc.noise_dim, generator=gen)
        n["z0_raw"] = torch.randn(B, mc.noise_dim, generator=gen)
    n["sigma"] = torch.randn(B, tc.nz_cdae * tc.nstd, 1, generator=gen)   # stdmat
    n["eps"] = torch.randn(N * tc.nstd, mc.z_dim, generator=gen)          # add_gaussian_noise
    n["vae"] = torch.randn(B * sq(tc.nz_model), mc.noise_dim, generator=gen)
    if aux:
        n["vae_z"] = torch.randn(B * tc.nz_model, mc.z_dim, generator=gen)
    if aux and mc.clipped:
        n["vctx_raw"] = torch.randn(B, mc.noise_dim, generator=gen)
        n["vz0_raw"] = torch.randn(B, mc.noise_dim, generator=gen)
    return n


def sampler_noise(mc: ModelCfg, noise, which):
    """The sampler's draws out of a step's noise dict: which = "sampler" (cDAE phase) | "vae"."""
    return (noise[which], noise[which + "_z"]) if mc.kind in AUX_KINDS else noise[which]


# --------------------------------------------------------------------------- #
# one train step
# --------------------------------------------------------------------------- #
def _grads_of(loss, p):
    names = list(p.keys())
    gs = torch.autograd.grad(loss, [p[n] for n in names], allow_unused=True)
    return {n: g for n, g in zip(names, gs)}


def cdae_update_grads(mc, cc, tc, pm, pc, x, noise):
    """cDAE phase up to (not including) the optimiser step  (ivae_ardae.py:713-771).
    Returns loss, grads dict (None for tensors the loss does not reach), std [B,1,1]."""
    B = x.size(0)
    with torch.no_grad():
        z0 = encode(mc, pm, x, zero_noise(mc, B, x, noise.get("z0_raw")), 1)          # latent_mean (and the context for lt0)
        if mc.clipped and tc.ctx_type == "lt0":      # the reference's two encode(x, std=0) calls draw separately (ivae_ardae.py:741,748)
            ctx = encode(mc, pm, x, zero_noise(mc, B, x, noise.get("ctx_raw")), 1)
        else:
            ctx = z0 if tc.ctx_type == "lt0" else cdae_context(mc, tc, pm, x, noise.get("ctx_raw")).unsqueeze(1)
        latent = encode(mc, pm, x, sampler_noise(mc, noise, "sampler"), tc.nz_cdae)
        u, std = latent_stats(latent, z0, tc.std_scale, tc.delta)
        stdmat = std * noise["sigma"]
        u_exp = u.unsqueeze(2).expand(B, tc.nz_cdae, tc.nstd, mc.z_dim).reshape(B, tc.nz_cdae * tc.nstd, mc.z_dim)
    pc_req = {k: v.detach().requires_grad_(True) for k, v in pc.items()}
    loss = cdae_forward(cc, pc_req, u_exp, ctx, stdmat, noise["eps"])
    grads = _grads_of(loss, pc_req)
    return loss.detach(), grads, std


def vae_update_grads(mc, cc, tc, pm, pc, x, noise, beta=None):
    """VAE phase up to the optimiser step (ivae_ardae.py:781-834)."""
    beta = tc.beta if beta is None else beta
    B = x.size(0)
    pm_req = {k: v.detach().requires_grad_(True) for k, v in pm.items()}
    z, loss, rec, pri, _ = vae_forward(mc, pm_req, x, sampler_noise(mc, noise, "vae"), beta, tc.nz_model)
    with torch.no_grad():
        z0 = encode(mc, pm, x, zero_noise(mc, B, x, noise.get("vz0_raw")), 1)
        if mc.clipped and tc.ctx_type == "lt0":
            ctx = encode(mc, pm, x, zero_noise(mc, B, x, noise.get("vctx_raw")), 1)
        else:
            ctx = z0 if tc.ctx_type == "lt0" else cdae_context(mc, tc, pm, x, noise.get("vctx_raw")).unsqueeze(1)
    u = (tc.std_scale * (z - z0)).detach()
    stdmat = x.new_zeros(B, tc.nz_model, 1)
    g = cdae_glogprob(cc, pc, u, ctx, stdmat)
    seed = beta * g / float(B * tc.nz_model)
    # loss.backward(); (s*(z - z0)).backward(seed)  ==  d/dp [loss + sum(s*z*seed)]
    total = loss + (tc.std_scale * (z - z0) * seed).sum()
    grads = _grads_of(total, pm_req)
    return loss.detach(), rec, pri, g, grads


def train_step(mc, cc, tc, pm, pc, st_m, st_c, x_cdae, x_vae, noise):
    """One iteration of ivae_ardae.py:707-846 with num_cdae_updates=1.  Mutates pm/pc/states."""
    closs, gc, std = cdae_update_grads(mc, cc, tc, pm, pc, x_cdae, noise)
    with torch.no_grad():
        optimizer_step(tc.d_optimizer, pc, gc, st_c, tc.d_lr, tc.d_beta1, tc.d_momentum)
    mloss, rec, pri, g, gm = vae_update_grads(mc, cc, tc, pm, pc, x_vae, noise)
    with torch.no_grad():      # the model's RMSprop takes --d-momentum too (ivae_ardae.py:553: momentum=opt.d_momentum)
        optimizer_step(tc.m_optimizer, pm, gm, st_m, tc.m_lr, tc.m_beta1, tc.d_momentum)
    return {"cdae_loss": closs, "std_mean": std.mean(), "std_max": std.max(), "std_min": std.min(),
            "model_loss": mloss, "recon": rec, "prior": pri}


# --------------------------------------------------------------------------- #
# IWAE evaluation (quality gate, not timed)
# --------------------------------------------------------------------------- #
def iwae_logprob(mc: ModelCfg, pm, x, sample_size, enc_noise, prop_noise):
    """IWAE-k with a full-covariance Gaussian fitted to encoder samples as proposal.
    enc_noise [B, k, noise_dim], prop_noise [B, k, z] (standard normal draws)."""
    assert sample_size >= 2 * mc.z_dim
    B = x.size(0)
    x = x.reshape(B, mc.input_dim)
    with torch.no_grad():
        if mc.kind == "auxtoy":       # Encoder._forward(nz = sample_size) (ivae/auxtoy.py:313): k z0's x k z's = k^2 encoder samples per image fit
            # the proposal; enc_noise = (eps0 [B, k, noise_dim], eps [B, k k, z])
            noise = (enc_noise[0].reshape(B * sample_size, mc.noise_dim), enc_noise[1].reshape(B * sample_size * sample_size, mc.z_dim))
            z = aux_encode(mc, pm, x, noise, sample_size)["z"].view(B, sample_size * sample_size, mc.z_dim)
        else:
            if mc.kind in AUX_KINDS:      # enc_noise = (eps0 [B,k,noise_dim], eps [B,k,z]); ivae/auxmnist.py:306-326
                noise = (enc_noise[0].reshape(B * sample_size, mc.noise_dim), enc_noise[1].reshape(B * sample_size, mc.z_dim))
            else:
                noise = enc_noise.reshape(B * sample_size, mc.noise_dim)
            z = encode(mc, pm, x, noise, sample_size)   # [B,k,z]
        mu = z.mean(1)
        zc = z - mu.unsqueeze(1)
        cov = zc.transpose(1, 2) @ zc / (z.size(1) - 1)
        if mc.kind in AUX_KINDS:
            cov = cov + 1e-5 * torch.eye(mc.z_dim, dtype=cov.dtype)      # ivae/auxmnist.py:321, ivae/auxresconv.py:299
        Lc = torch.linalg.cholesky(cov)
        newz = mu.unsqueeze(1) + prop_noise @ Lc.transpose(1, 2)
        half_logdet = torch.log(torch.diagonal(Lc, dim1=1, dim2=2)).sum(1, keepdim=True)
        logq = -0.5 * (prop_noise ** 2).sum(2) - half_logdet - 0.5 * mc.z_dim * LOG2PI
        logp = -prior_rows(newz.reshape(B * sample_size, mc.z_dim)).view(B, sample_size)
        dist = decode(mc, pm, newz.reshape(B * sample_size, mc.z_dim))
        loglik = -recon_rows(mc, dist, expand_rows(x, sample_size)).view(B, sample_size)
        lw = loglik + logp - logq
        m, _ = lw.max(1, keepdim=True)
        out = torch.log(torch.mean((lw - m).exp(), 1, keepdim=True) + 1e-10) + m
    return out.mean()
