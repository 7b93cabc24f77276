"""Host side of csrc/fused_final.hip: the last conditioner layer + splines of an RQS coupling in one
kernel for any number of transformed features (hidden width 128, linear tails, 8, 10 or 16 bins).

Packed buffer (floats): for feature group g (features 4 g .. 4 g + 3), row block b, k-step s the two
fragments hi | lo of 64 lanes x 8 halves, lane = 16 q' + i holding
    W[(4 g + (i >> 2)) * P + 4 b + (i & 3)][32 s + 8 q' + j],  j = 0..7      (natural k order: the
operand comes from memory, not from a previous layer's accumulators), rows with 4 b + (i & 3) >= P or
feature >= d_t zero; then the bias rows [g][q][4 P4] in accumulator order.  Weights and biases carry the logit
scale and the log2(e) factors of the spline's exponentials (see pack).
"""
import torch

from . import _lib
from .fused import _split_halves, _as_floats


def eligible(coupling, inputs, context):
    from .nets.resnet import ResidualNet
    net = coupling.transform_net
    if type(net) is not ResidualNet or net.preprocessing is not None or inputs.dim() != 2:
        return False
    if any(b.use_batch_norm or (b.dropout.p > 0 and b.training) for b in net.blocks):
        return False
    if (context is None) != (not net.context_features):
        return False
    from . import fused
    if fused.precision_of(coupling) != fused.PREC_F16X3:        # fused_precision='fp32' asks for exact fp32 products
        return False
    code = {'linear': _lib.TAILS_LINEAR, None: _lib.TAILS_NONE, 'circular': _lib.TAILS_CIRCULAR}.get(coupling.tails, -1)
    return bool(_lib.lib().vcnf_rqs_final_fused_supported(coupling.num_transform_features, net.hidden_features,
                                                          coupling.num_bins, code))


def pack(weight, bias, d_t, k, wh_scale):
    """Scaled like fused._pack_final6 (csrc/rqs_lean.hpp evaluates 2^x on pre-scaled logits): width / height rows by
    wh_scale * log2(e) (coupling.py:314-316), derivative rows by log2(e)."""
    from .fused import LOG2E
    p = 3 * k - 1
    scale = torch.where(torch.arange(p, device=weight.device) < 2 * k, wh_scale * LOG2E, LOG2E).double().repeat(d_t)
    weight = (weight.double() * scale.view(-1, 1)).float()
    bias = (bias.double() * scale).float()
    p4 = (p + 3) // 4
    ng = (d_t + 3) // 4
    hdim = weight.shape[1]
    dev = weight.device
    g = torch.arange(ng, device=dev).view(-1, 1, 1, 1, 1)
    b = torch.arange(p4, device=dev).view(1, -1, 1, 1, 1)
    s = torch.arange(hdim // 32, device=dev).view(1, 1, -1, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 1, 1, -1, 1)
    j = torch.arange(8, device=dev).view(1, 1, 1, 1, -1)
    i = lane & 15
    t = 4 * b + (i & 3)
    feat = 4 * g + (i >> 2)
    shape = (ng, p4, hdim // 32, 64, 8)
    ok = ((t < p) & (feat < d_t)).expand(shape)
    rows = torch.where((t < p) & (feat < d_t), feat * p + t, torch.zeros_like(feat + t)).expand(shape)
    cols = (32 * s + 8 * (lane >> 4) + j).expand(shape)
    w = torch.where(ok, weight[rows, cols], torch.zeros((), device=dev, dtype=weight.dtype))
    hi, lo = _split_halves(w)                                   # [ng, p4, ns, 64, 8] each
    frag = torch.stack([hi, lo], dim=3)                         # [ng, p4, ns, 2, 64, 8]
    g2 = torch.arange(ng, device=dev).view(-1, 1, 1)
    q2 = torch.arange(4, device=dev).view(1, -1, 1)
    t2 = torch.arange(4 * p4, device=dev).view(1, 1, -1)
    okb = ((t2 < p) & (4 * g2 + q2 < d_t)).expand(ng, 4, 4 * p4)
    idx = torch.where(okb, ((4 * g2 + q2) * p + t2).expand(ng, 4, 4 * p4), torch.zeros((), device=dev, dtype=torch.long))
    bf = torch.where(okb, bias[idx], torch.zeros((), device=dev, dtype=bias.dtype))
    return torch.cat([_as_floats(frag), bf.reshape(-1).float()]).contiguous()


def packed_weights(coupling):
    lin = coupling.transform_net.final_layer
    key = tuple((t.data_ptr(), t._version, str(t.device)) for t in (lin.weight, lin.bias))
    cache = coupling.__dict__.setdefault('_fused_final_pack', {})
    if cache.get('key') != key:
        cache['key'] = key
        with torch.no_grad():
            buf = pack(lin.weight.detach(), lin.bias.detach(), coupling.num_transform_features, coupling.num_bins,
                       float(coupling._cfg(True).wh_scale))
        old = cache.get('buf')
        if old is not None and old.shape == buf.shape and old.device == buf.device:
            old.copy_(buf)
        else:
            cache['buf'] = buf
    want = int(_lib.lib().vcnf_rqs_final_fused_pack_floats(coupling.num_transform_features,
                                                           coupling.transform_net.hidden_features, coupling.num_bins))
    assert cache['buf'].numel() == want, (cache['buf'].numel(), want)
    return cache['buf']


def trunk_eligible(net, first_in, context):
    """ResidualNet trunks the one-launch kernel (csrc/resnet_trunk.hip) covers: hidden 128, ReLU, 1-3 blocks, no context
    gate, inference."""
    from .fused import _is_relu
    from . import autograd
    if context is not None or net.context_features:
        return False
    if any(not _is_relu(b.activation) for b in net.blocks):
        return False
    if autograd.needs_grad(first_in, *net.parameters()):
        return False
    return bool(_lib.lib().vcnf_resnet_trunk_supported(first_in.shape[1], net.hidden_features, len(net.blocks)))


def pack_trunk(net):
    from .fused_affine import _pack_chained
    hb = net.hidden_features // 16
    w0 = net.initial_layer.weight.detach()
    parts = [_pack_chained(w0, hb, w0.shape[1] // 16), net.initial_layer.bias.detach()]
    for blk in net.blocks:
        for lin in blk.linear_layers:
            parts += [_pack_chained(lin.weight.detach(), hb, hb), lin.bias.detach()]
    return torch.cat([p.reshape(-1).float() for p in parts]).contiguous()


def packed_trunk(coupling):
    net = coupling.transform_net
    params = [net.initial_layer.weight, net.initial_layer.bias] + [p for b in net.blocks for l in b.linear_layers
                                                                   for p in (l.weight, l.bias)]
    key = tuple((t.data_ptr(), t._version, str(t.device)) for t in params)
    cache = coupling.__dict__.setdefault('_fused_trunk_pack', {})
    if cache.get('key') != key:
        cache['key'] = key
        with torch.no_grad():
            buf = pack_trunk(net)
        old = cache.get('buf')
        if old is not None and old.shape == buf.shape and old.device == buf.device:
            old.copy_(buf)
        else:
            cache['buf'] = buf
    return cache['buf']


def _identity_kernel_ok(coupling):
    """The identity features go through csrc/rqs_kernels.hip::rqs_identity_half_kernel: no unconditional spline, or one
    with a single (tails, bound) for all features in a bin count that kernel is built for."""
    uncond = coupling.unconditional_transform
    if uncond is None:
        return True
    if uncond.per_feature or uncond.tails not in ('linear', None):
        return False
    code = _lib.TAILS_LINEAR if uncond.tails == 'linear' else _lib.TAILS_NONE
    return bool(_lib.lib().vcnf_rqs_identity_half_supported(uncond.num_bins, code))


def run(coupling, inputs, context, sampling, log_q=None, sign=1.0):
    """Coupling layer in three launches: identity half (gather, unconditional spline, conditioner input), conditioner
    trunk (csrc/resnet_trunk.hip, or PyTorch-ROCm for shapes it does not cover), last layer + splines
    (csrc/fused_final.hip); same (out, log_det) contract as PiecewiseRationalQuadraticCoupling._run."""
    net = coupling.transform_net
    uncond = coupling.unconditional_transform
    d_t, k = coupling.num_transform_features, coupling.num_bins
    out = torch.empty_like(inputs)
    lad_i = None
    if _identity_kernel_ok(coupling):
        shared = uncond.logits() if uncond is not None else None
        rows_f = int(_lib.lib().vcnf_rqs_final_fused_partial_rows(d_t, k))
        rows_i = _lib.identity_half_rows(coupling.num_identity_features, shared)
        partial = torch.empty(rows_f + rows_i, inputs.shape[0], dtype=torch.float32, device=inputs.device)
        xi = _lib.rqs_identity_half(inputs, out, coupling._index32('id'), coupling.num_identity_features, shared,
                                    uncond._cfg() if uncond is not None else None, sampling,
                                    partial=partial[rows_f:] if rows_i else None)
    else:
        partial = None
        xi = inputs[:, coupling.identity_features]
        if sampling:                                    # coupling.py:110-114: the conditioner sees S^-1(x_id)
            xi, lad_i = uncond.inverse(xi)
            out[:, coupling.identity_features] = xi
        else:
            out[:, coupling.identity_features], lad_i = uncond.forward(xi)
    first_in = xi if context is None else torch.cat((xi, context), dim=1)
    presplit = False
    if coupling.fused_trunk and trunk_eligible(net, first_in, context):
        # the trunk kernel hands its output over already split into fp16 halves (bit-identical to splitting in the
        # last-layer kernel, done once per sample instead of once per sample and feature-group workgroup)
        presplit = True
        h = _lib.resnet_trunk(first_in, packed_trunk(coupling), net.hidden_features, len(net.blocks), split=True)
    else:
        h = net.hidden(first_in, context)
    partial = _lib.rqs_final_fused(inputs, h, out, coupling._index32('tf'), d_t, net.hidden_features,
                                   packed_weights(coupling), coupling._cfg(True), sampling, partial=partial,
                                   presplit=presplit)
    lad = partial.sum(0) if partial.shape[0] > 1 else partial[0]
    if lad_i is not None:
        lad = lad + lad_i
    if log_q is not None:
        return out, log_q.add_(lad, alpha=sign)
    return out, (lad if sign == 1.0 else sign * lad)
