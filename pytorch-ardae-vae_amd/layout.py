"""Parameter names / shapes / flat-buffer offsets, identical to the reference's `named_parameters()` order.

Reference: models/layers.py:477-515 (MLP: `layers.{i}.{weight,bias}`, `fc.{weight,bias}`), :681-724 (ContextConcatMLP),
models/ivae/mnist.py:123-199, models/ivae/toy.py:154-194,694-737, models/graddae/mlp.py:342-378,
models/resdae/mlp.py:287-326.  The C++ side (csrc/cdae.hip, csrc/model.hip) computes the same offsets; the ABI
test checks `ardae_*_param_floats` against these totals.
"""
import math


def _mlp(prefix, din, dh, dout, n_hidden, extra_in=0):
    spec = []
    for i in range(n_hidden):
        spec += [(f"{prefix}layers.{i}.weight", (dh, (din if i == 0 else dh) + extra_in)), (f"{prefix}layers.{i}.bias", (dh,))]
    spec += [(f"{prefix}fc.weight", (dout, (din if n_hidden == 0 else dh) + extra_in)), (f"{prefix}fc.bias", (dout,))]
    return spec


def model_spec(kind, input_dim, noise_dim, h_dim, z_dim, n_layers, enc_type="res-wn-mlp"):
    if kind == "mnist":
        s = _mlp("encode.inp_encode.", input_dim, h_dim, h_dim, n_layers + 1)
        s += _mlp("encode.fc.", h_dim + noise_dim, h_dim, z_dim, 1)
        s += _mlp("decode.main.", z_dim, h_dim, h_dim, n_layers)
        s += [("decode.reparam.logit_fn.weight", (input_dim, h_dim)), ("decode.reparam.logit_fn.bias", (input_dim,))]
    elif kind == "conv":   # models/ivae/conv.py:44-136 + models/vae/conv.py:79-136 (28x28x1, fixed architecture)
        s = [("encode.conv1.weight", (16, 1, 5, 5)), ("encode.conv1.bias", (16,)),
             ("encode.conv2.weight", (32, 16, 5, 5)), ("encode.conv2.bias", (32,)),
             ("encode.conv3.weight", (32, 32, 5, 5)), ("encode.conv3.bias", (32,)),
             ("encode.fc4.weight", (800, 512 + noise_dim)), ("encode.fc4.bias", (800,)),
             ("encode.fc5.weight", (z_dim, 800)), ("encode.fc5.bias", (z_dim,))]
        s += _mlp("decode.fc.", z_dim, 300, 512, 1)
        s += [("decode.deconv1.weight", (32, 32, 5, 5)), ("decode.deconv1.bias", (32,)),
              ("decode.deconv2.weight", (32, 16, 5, 5)), ("decode.deconv2.bias", (16,)),
              ("decode.reparam.logit_fn.weight", (16, 1, 5, 5)), ("decode.reparam.logit_fn.bias", (1,))]
    elif kind in ("resconv", "auxresconv"):
        # weight-normalised residual blocks (models/layers2.py:50-93,237-352; models/layers.py:25-85 inside ResMLP): three operators per
        # block, each with direction / scale / bias; ivae/resconv.py:81-125, vae/auxresconv.py:36-63,94,149-153, vae/resconv.py:89-109
        def wn(prefix, out, inn, conv):
            return [(prefix + "direction", (out, inn, 3, 3) if conv else (out, inn)), (prefix + "scale", (out,)), (prefix + "bias", (out,))]

        def block(prefix, out, inn, conv):
            a, b, c = ("conv_0h.", "conv_h1.", "conv_01.") if conv else ("dot_0h.", "dot_h1.", "dot_01.")
            return wn(prefix + a, out, inn, conv) + wn(prefix + b, out, out, conv) + wn(prefix + c, out, inn, conv)
        cdim = 512 if kind == "resconv" else h_dim
        tp = "encode.inp_encode." if kind == "resconv" else "encode.inp_encode.enc."
        s = []
        for i, (o, inn) in zip((0, 2, 4, 6, 8), ((16, 1), (16, 16), (32, 16), (32, 32), (32, 32))):
            s += block(f"{tp}{i}.", o, inn, True)
        s += block(f"{tp}11.", cdim, 512, False)
        if kind == "resconv":
            s += resconv_head_spec(enc_type, n_layers, cdim + noise_dim, h_dim, z_dim)
        else:
            s += [("encode.aux_encode.reparam.mean_fn.weight", (noise_dim, cdim)), ("encode.aux_encode.reparam.mean_fn.bias", (noise_dim,)),
                  ("encode.aux_encode.reparam.logvar_fn.weight", (noise_dim, cdim)), ("encode.aux_encode.reparam.logvar_fn.bias", (noise_dim,)),
                  ("encode.encode.fc.0.weight", (cdim, cdim + noise_dim)), ("encode.encode.fc.0.bias", (cdim,)),
                  ("encode.encode.reparam.mean_fn.weight", (z_dim, cdim)), ("encode.encode.reparam.mean_fn.bias", (z_dim,)),
                  ("encode.encode.reparam.logvar_fn.weight", (z_dim, cdim)), ("encode.encode.reparam.logvar_fn.bias", (z_dim,))]
        s += block("decode.dec.0.", cdim, z_dim, False) + block("decode.dec.2.", 512, cdim, False)
        for i, (o, inn) in zip((6, 8, 12, 14, 17), ((32, 32), (32, 32), (16, 32), (16, 16), (1, 16))):
            s += block(f"decode.dec.{i}.", o, inn, True)
    elif kind == "toy":
        s = _mlp("encode.inp_encode.", input_dim, h_dim, h_dim, n_layers - 1)
        s += _mlp("encode.fc.", h_dim, h_dim, z_dim, n_layers, extra_in=noise_dim)
        s += _mlp("decode.main.", z_dim, h_dim, h_dim, n_layers - 1)
        s += [("decode.reparam.mean_fn.weight", (input_dim, h_dim)), ("decode.reparam.mean_fn.bias", (input_dim,)),
              ("decode.reparam.logvar_fn.weight", (input_dim, h_dim)), ("decode.reparam.logvar_fn.bias", (input_dim,))]
    elif kind == "auxtoy":     # models/ivae/auxtoy.py:44-130 (AuxEncoder + SimpleEncoder of models/vae/auxtoy.py) + models/vae/toy.py Decoder (Gaussian)
        s = _mlp("encode.aux_encode.main.", input_dim, h_dim, h_dim, n_layers - 1)
        s += [("encode.aux_encode.reparam.mean_fn.weight", (noise_dim, h_dim)), ("encode.aux_encode.reparam.mean_fn.bias", (noise_dim,)),
              ("encode.aux_encode.reparam.logvar_fn.weight", (noise_dim, h_dim)), ("encode.aux_encode.reparam.logvar_fn.bias", (noise_dim,))]
        s += _mlp("encode.encode.fc.", input_dim + noise_dim, h_dim, h_dim, n_layers - 1)
        s += [("encode.encode.reparam.mean_fn.weight", (z_dim, h_dim)), ("encode.encode.reparam.mean_fn.bias", (z_dim,)),
              ("encode.encode.reparam.logvar_fn.weight", (z_dim, h_dim)), ("encode.encode.reparam.logvar_fn.bias", (z_dim,))]
        s += _mlp("decode.main.", z_dim, h_dim, h_dim, n_layers - 1)
        s += [("decode.reparam.mean_fn.weight", (input_dim, h_dim)), ("decode.reparam.mean_fn.bias", (input_dim,)),
              ("decode.reparam.logvar_fn.weight", (input_dim, h_dim)), ("decode.reparam.logvar_fn.bias", (input_dim,))]
    elif kind == "auxmnist":   # models/ivae/auxmnist.py:47-132 (AuxEncoder + SimpleEncoder of models/vae/auxmnist.py) + models/vae/mnist.py Decoder
        s = _mlp("encode.aux_encode.main.", input_dim, h_dim, h_dim, n_layers - 1)
        s += [("encode.aux_encode.reparam.mean_fn.weight", (noise_dim, h_dim)), ("encode.aux_encode.reparam.mean_fn.bias", (noise_dim,)),
              ("encode.aux_encode.reparam.logvar_fn.weight", (noise_dim, h_dim)), ("encode.aux_encode.reparam.logvar_fn.bias", (noise_dim,))]
        s += _mlp("encode.encode.fc.", input_dim + noise_dim, h_dim, h_dim, n_layers - 1)
        s += [("encode.encode.reparam.mean_fn.weight", (z_dim, h_dim)), ("encode.encode.reparam.mean_fn.bias", (z_dim,)),
              ("encode.encode.reparam.logvar_fn.weight", (z_dim, h_dim)), ("encode.encode.reparam.logvar_fn.bias", (z_dim,))]
        s += _mlp("decode.main.", z_dim, h_dim, h_dim, n_layers - 1)
        s += [("decode.reparam.logit_fn.weight", (input_dim, h_dim)), ("decode.reparam.logit_fn.bias", (input_dim,))]
    elif kind == "auxconv":    # models/ivae/auxconv.py:48-126 (AuxEncoder + Encoder of models/vae/auxconv.py) + models/vae/conv.py Decoder
        def trunk(prefix, fc_in):
            return [(prefix + "conv1.weight", (16, 1, 5, 5)), (prefix + "conv1.bias", (16,)),
                    (prefix + "conv2.weight", (32, 16, 5, 5)), (prefix + "conv2.bias", (32,)),
                    (prefix + "conv3.weight", (32, 32, 5, 5)), (prefix + "conv3.bias", (32,)),
                    (prefix + "fc.weight", (800, fc_in)), (prefix + "fc.bias", (800,))]
        def heads(prefix, out):
            return [(prefix + "reparam.mean_fn.weight", (out, 800)), (prefix + "reparam.mean_fn.bias", (out,)),
                    (prefix + "reparam.logvar_fn.weight", (out, 800)), (prefix + "reparam.logvar_fn.bias", (out,))]
        s = trunk("encode.aux_encode.", 512) + heads("encode.aux_encode.", noise_dim)
        s += trunk("encode.encode.", 512 + noise_dim) + heads("encode.encode.", z_dim)
        s += _mlp("decode.fc.", z_dim, 300, 512, 1)
        s += [("decode.deconv1.weight", (32, 32, 5, 5)), ("decode.deconv1.bias", (32,)),
              ("decode.deconv2.weight", (32, 16, 5, 5)), ("decode.deconv2.bias", (16,)),
              ("decode.reparam.logit_fn.weight", (16, 1, 5, 5)), ("decode.reparam.logit_fn.bias", (1,))]
    else:
        raise NotImplementedError(kind)
    return s


def cdae_spec(kind, input_dim, context_dim, h_dim, n_layers):
    s = _mlp("ctx_encode.", context_dim, h_dim, h_dim, n_layers - 1)
    s += _mlp("inp_encode.", input_dim, h_dim, h_dim, n_layers - 1)
    if kind == "grad":
        s += _mlp("neglogprob.", 2 * h_dim + 1, h_dim, 1, n_layers)
    elif kind == "res":
        s += _mlp("dae.", 2 * h_dim + 1, h_dim, input_dim, n_layers)
    else:
        raise NotImplementedError(kind)
    return s


def offsets(spec):
    out, off = {}, 0
    for name, shape in spec:
        n = math.prod(shape)
        out[name] = (off, n, shape)
        off += n
    return out, off


RESCONV_HEADS = {"res-wn-mlp": 0, "mlp": 1, "res-mlp": 2, "res-wn-mlp-lin": 3, "res-mlp-lin": 4}     # ardae_model_desc.flags bits 1-3


def resconv_head_spec(enc_type, n_layers, cin, h_dim, z_dim):
    """Parameters of ResConvIPVAE's sampler head `encode.fc` in the reference's order (models/ivae/resconv.py:101-116): an MLP
    ('mlp'), a ResMLP of ResLinear blocks ('res-wn-mlp': un-normalised WeightNormalizedLinear operators direction / scale / bias;
    'res-mlp': nn.Linear operators weight / bias; models/layers.py:25-85,477-515,559-622) or `Sequential(ResMLP(... -> h_dim, output
    activated), Linear(h_dim, z_dim))` ('-lin').  A ResLinear whose input and output widths coincide has no skip operator (same_dim)."""
    if enc_type not in RESCONV_HEADS:
        raise AssertionError(enc_type)                      # ivae/resconv.py:77
    lin = lambda pre, out, inn: [(pre + "weight", (out, inn)), (pre + "bias", (out,))]

    def oper(pre, out, inn, wn):
        return [(pre + "direction", (out, inn)), (pre + "scale", (out,)), (pre + "bias", (out,))] if wn else lin(pre, out, inn)

    def res(pre, out, inn, wn):
        return oper(pre + "dot_0h.", out, inn, wn) + oper(pre + "dot_h1.", out, out, wn) + ([] if inn == out else oper(pre + "dot_01.", out, inn, wn))
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
