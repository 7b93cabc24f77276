"""Conditioner networks as pure functions of a state dict, CPU oracle (test
infrastructure).  The conditioners are the *callees* of the hot path: dense
layers whose outputs parameterise the bijector.

* ``residual_net`` <- ``normflow/nets/resnet.py`` ResidualNet :60-106,
  ResidualBlock :8-57 (no batch norm, dropout p=0 -> identity)
* ``conv_residual_net`` <- ``normflow/nets/resnet.py`` ConvResidualNet :163-212, ConvResidualBlock :109-160
* ``made``         <- ``normflow/nets/made.py`` MADE :215-300, MaskedResidualBlock :138-212, MaskedLinear :79-80
* ``mlp``          <- ``normflow/nets/mlp.py`` MLP :7-58 (Linear + LeakyReLU)
"""
import torch
import torch.nn.functional as F


def _lin(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def residual_net(sd, prefix, x, context=None, activation=F.relu):
    """resnet.py:92-106.  With a context the first layer sees cat(x, context)
    (:97-102) and every block gates its branch with
    glu(cat(branch, context_layer(context))) (:49-56)."""
    h = _lin(sd, prefix + "initial_layer", x if context is None else torch.cat((x, context), dim=1))
    i = 0
    while (prefix + "blocks.%d.linear_layers.0.weight" % i) in sd:
        bp = prefix + "blocks.%d." % i
        t = activation(h)                                   # :42
        t = _lin(sd, bp + "linear_layers.0", t)             # :43
        t = activation(t)                                   # :46
        t = _lin(sd, bp + "linear_layers.1", t)             # :48
        if context is not None:                             # :49-56
            t = F.glu(torch.cat((t, _lin(sd, bp + "context_layer", context)), dim=1), dim=1)
        h = h + t                                           # :57
        i += 1
    return _lin(sd, prefix + "final_layer", h)              # :105


def conv_residual_net(sd, prefix, x, context=None, activation=F.relu):
    """resnet.py:199-212 (ConvResidualNet) with ConvResidualBlock :139-160: 1x1 conv on
    cat(x, context), blocks of two 3x3 convs with an optional GLU gate from a 1x1 conv of the
    context image, 1x1 conv out (no batch norm, dropout p=0)."""
    conv = lambda key, t, pad: F.conv2d(t, sd[key + ".weight"], sd[key + ".bias"], padding=pad)
    h = conv(prefix + "initial_layer", x if context is None else torch.cat((x, context), dim=1), 0)
    i = 0
    while (prefix + "blocks.%d.conv_layers.0.weight" % i) in sd:
        bp = prefix + "blocks.%d." % i
        t = activation(h)                                   # :143
        t = conv(bp + "conv_layers.0", t, 1)                # :144
        t = activation(t)                                   # :147
        t = conv(bp + "conv_layers.1", t, 1)                # :149
        if context is not None:                             # :150-158
            t = F.glu(torch.cat((t, conv(bp + "context_layer", context, 0)), dim=1), dim=1)
        h = h + t                                           # :159
        i += 1
    return conv(prefix + "final_layer", h, 0)               # :211


def periodic_features(sd, prefix, x, scale):
    """utils/nn.py:111-118: columns ``ind`` -> w0 sin(scale x) + w1 cos(scale x), others unchanged."""
    ind, rest, inv_perm = sd[prefix + "ind"], sd[prefix + "ind_"], sd[prefix + "inv_perm"]
    w = sd[prefix + "weights"]
    a = scale * x[..., ind]
    a = w[:, 0] * torch.sin(a) + w[:, 1] * torch.cos(a)
    return torch.cat((a, x[..., rest]), -1)[..., inv_perm]


def made(sd, prefix, x, context=None, activation=F.relu, preprocess=None):
    """nets/made.py:292-300 (MADE.forward) with residual blocks :196-212: every Linear uses
    weight * mask (:79-80); the masks are buffers of the state dict."""
    def masked(key, t):
        return F.linear(t, sd[key + ".weight"] * sd[key + ".mask"], sd[key + ".bias"])
    h = x if preprocess is None else preprocess(x)
    h = masked(prefix + "initial_layer", h)
    if context is not None:
        h = h + _lin(sd, prefix + "context_layer", context)
    i = 0
    while (prefix + "blocks.%d.linear_layers.0.weight" % i) in sd:
        bp = prefix + "blocks.%d." % i
        t = activation(h)
        t = masked(bp + "linear_layers.0", t)
        t = activation(t)
        t = masked(bp + "linear_layers.1", t)
        if context is not None:
            t = F.glu(torch.cat((t, _lin(sd, bp + "context_layer", context)), dim=1), dim=1)
        h = h + t
        i += 1
    return masked(prefix + "final_layer", h)


def mlp(sd, prefix, x, leaky=0.0):
    """mlp.py:30-35,55-58 without output_fn: Linear, LeakyReLU, ..., Linear.
    Linear layers sit at even positions of ``net`` (keys ``net.{2k}``)."""
    idx = sorted({int(k[len(prefix) + 4:].split(".")[0]) for k in sd
                  if k.startswith(prefix + "net.") and k.endswith(".weight")})
    h = x
    for n, i in enumerate(idx):
        h = _lin(sd, prefix + "net.%d" % i, h)
        if n + 1 < len(idx):
            h = F.leaky_relu(h, leaky)
    return h


def conv_net(sd, prefix, x, leaky=0.0):
    """nets/cnn.py:27-46 without actnorm: Conv2d ('same' padding), LeakyReLU, ..., Conv2d;
    convolutions sit at even positions of ``net``."""
    idx = sorted({int(k[len(prefix) + 4:].split(".")[0]) for k in sd
                  if k.startswith(prefix + "net.") and k.endswith(".weight")})
    h = x
    for n, i in enumerate(idx):
        w, b = sd[prefix + "net.%d.weight" % i], sd[prefix + "net.%d.bias" % i]
        h = F.conv2d(h, w, b, padding=w.shape[-1] // 2)
        if n + 1 < len(idx):
            h = F.leaky_relu(h, leaky)
    return h
