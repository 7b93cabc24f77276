"""Host side of csrc/masked_affine_stack.hip: runs of ``MaskedAffineFlow`` layers (with their MLP conditioners) and of the
``AffineConstFlow`` / ``ActNorm`` layers between them in ONE launch - the models of the reference's own drivers
(/root/reference/run.py:58-68, runadultvdeq.py:101-108: K x [MaskedAffineFlow(b, t, s), ActNorm], s, t = MLP([D, H, D]),
fp32 or .double(), 1024 - 2048 samples per call), whose layer-by-layer evaluation is ~9 launches of a microsecond of work
per (coupling, ActNorm) pair.

``plan`` finds the longest run a launch covers, ``run`` builds (and caches) the device table of parameter addresses the
kernel walks.  Anything outside the kernel's family - conditioners that are not Linear-LeakyReLU-Linear, more than 16
features or 64 hidden units, 4-D inputs, a gradient being required, an ActNorm that has not seen its first batch yet -
ends the run and takes the per-layer path.  Training goes through the same launch: ``MaskedStackFn`` is the run as one
autograd node whose backward is ONE launch of the VJP kernel (it keeps only the run's output and rebuilds every layer's
input from its output - coupling layers are invertible).
"""
import struct

import torch
from torch import nn

from . import _lib

MAX_LAYERS = 4096


def _mlp_parts(net, d):
    """(W1, b1, W2, b2, slope) of an ``MLP([d, H, d])`` conditioner, None for anything else; ``net is None`` -> ()."""
    if net is None:
        return ()
    from .nets.mlp import MLP
    if type(net) is not MLP:
        return None
    mods = list(net.net)
    if len(mods) != 3 or type(mods[0]) is not nn.Linear or type(mods[1]) is not nn.LeakyReLU or type(mods[2]) is not nn.Linear:
        return None
    l1, act, l2 = mods
    if l1.in_features != d or l2.out_features != d or l1.out_features != l2.in_features:
        return None
    if l1.bias is None or l2.bias is None:
        return None
    return l1.weight, l1.bias, l2.weight, l2.bias, float(act.negative_slope)


def _layer(flow, z):
    """Descriptor (kind, hidden, slope, tensors...) of one flow on inputs like ``z``, or None."""
    from .flows.affine.coupling import MaskedAffineFlow, AffineConstFlow
    from .flows.normalization import ActNorm
    d = z.shape[1]
    if type(flow) is MaskedAffineFlow:
        if flow.b.numel() != d:
            return None
        ps, pt = _mlp_parts(flow.s, d), _mlp_parts(flow.t, d)
        if ps is None or pt is None:
            return None
        hs = ps[0].shape[0] if ps else 0
        ht = pt[0].shape[0] if pt else 0
        if ps and pt and (hs != ht or ps[4] != pt[4]):
            return None
        h = hs or ht
        if not _lib.lib().vcnf_masked_affine_stack_supported(d, h):
            return None
        slope = (ps or pt or (0, 0, 0, 0, 0.0))[4]
        tens = [flow.b] + list(ps[:4] if ps else (None,) * 4) + list(pt[:4] if pt else (None,) * 4)
        return 0, h, slope, tens
    if type(flow) in (AffineConstFlow, ActNorm):
        if flow.s.numel() != d or flow.t.numel() != d or not _lib.lib().vcnf_masked_affine_stack_supported(d, 0):
            return None
        if type(flow) is ActNorm and not flow._initialised():
            return None                                       # the first batch initialises the layer (normalization.py:22-37): per-layer path until then
        return 1, 0, 0.0, [None, flow.s, flow.t] + [None] * 6
    return None


def _layer_tensors(flow, kind):
    """The nine table slots of a layer, read from the module as it is now: [b, s.W1, s.b1, s.W2, s.b2, t.W1, ...] or
    [None, s, t, None ...] for a per-feature layer."""
    if kind == 1:
        return [None, flow.s, flow.t] + [None] * 6
    out = [flow.b]
    for net in (flow.s, flow.t):
        if net is None:
            out += [None] * 4
        else:
            l1, l2 = net.net[0], net.net[2]
            out += [l1.weight, l1.bias, l2.weight, l2.bias]
    return out


def plan(order, start, z):
    """Longest run order[start:end] of at least two layers one launch evaluates, or None: [B, D] inputs on the device
    in fp32 or fp64, D <= 16, every layer in the kernel's family and of the inputs' dtype.  With autograd recording the
    run is one differentiable node (MaskedStackFn: one more launch for the whole backward pass)."""
    if z.dim() != 2 or not z.is_cuda or z.dtype not in (torch.float32, torch.float64) or z.shape[1] > 16:
        return None
    run = []
    for flow in order[start:start + MAX_LAYERS]:
        desc = _layer(flow, z)
        if desc is None:
            break
        tens = [t for t in desc[3] if t is not None]
        if any(t.dtype != z.dtype or t.device != z.device for t in tens):
            break
        run.append((flow, desc))
    if len(run) < 2:
        return None
    return start + len(run), run


def _stamp(flow):
    # ActNorm: whether it has seen its first batch decides whether it can be part of a run (asked of the layer itself: the
    # answer is cached on the host once it is yes, so this synchronises only while the layer is still uninitialised)
    init = flow._initialised() if hasattr(flow, '_initialised') else None
    return (id(getattr(flow, 's', None)), id(getattr(flow, 't', None)), init)


def cached_plan(owner, order, start, z):
    """``plan`` memoised on the calling model (the plan does not depend on whether autograd records): 64 layers of
    checks cost more than the launch they save.  A cached plan is reused while the run's modules (and their conditioner
    sub-modules) are the same objects; shape, dtype and device of the inputs are part of the key."""
    if z.dim() != 2:
        return None
    key = (start, len(order), id(order[start]), z.shape[1], z.dtype, str(z.device))
    plans = owner.__dict__.setdefault('_masked_stack_plans', {})
    hit = plans.get(key)
    if hit is not None:
        p, mods, stamp = hit
        if all(a is b for a, b in zip(order[start:start + len(mods)], mods)) and stamp == tuple(_stamp(f) for f in mods):
            return p
    p = plan(order, start, z)
    n = (p[0] - start if p is not None else 0) + 1              # the run and the flow that ended it
    mods = list(order[start:start + n])
    if len(plans) > 64:
        plans.clear()
    plans[key] = (p, mods, tuple(_stamp(f) for f in mods))
    return p


def run(steps, z, inverse, log_q, sign):
    """Execute a planned run in one launch.  The table of parameter addresses lives on the first layer of the run and is
    rebuilt when an address changes (parameters are read in place: in-place updates need nothing)."""
    first = steps[0][0]
    cache = first.__dict__.setdefault('_masked_stack', {})
    ids = tuple(id(f) for f, _ in steps)
    if cache.get('ids') != ids:
        cache.clear()
        cache['ids'] = ids
    # the tensors behind the table (parameters and the mask buffers; re-read from the modules: a re-assigned Parameter is seen)
    tens = []
    for flow, (kind, h, slope, ts) in steps:
        tens += [t for t in _layer_tensors(flow, kind) if t is not None]
    key = (tuple(t.data_ptr() for t in tens), str(z.device))
    if cache.get('key') != key:
        rows, keep = [], []
        for flow, (kind, h, slope, _) in steps:
            ts = [None if t is None else t.detach() for t in _layer_tensors(flow, kind)]
            if any(t is not None and not t.is_contiguous() for t in ts):
                raise _lib.VcnfError("masked affine stack: non-contiguous parameter")
            keep.append(ts)
            rows.append([kind, h, struct.unpack('q', struct.pack('d', slope))[0]] + [0 if t is None else t.data_ptr() for t in ts])
        cache['key'] = key
        cache['table'] = torch.tensor(rows, dtype=torch.int64, device=z.device)
        cache['n'] = len(rows)
    if torch.is_grad_enabled() and (z.requires_grad or any(t.requires_grad for t in tens)):
        if 'goff' not in cache or cache.get('goff_key') != key:
            offs, n = [], 0
            for flow, (kind, h, slope, _) in steps:
                offs.append(n)
                n += sum(t.numel() for t in _layer_tensors(flow, kind)[1:] if t is not None)
            cache['goff'] = torch.tensor(offs, dtype=torch.int64, device=z.device)
            cache['n_grad'] = n
            cache['goff_key'] = key
        params = [t for flow, (kind, h, slope, _) in steps for t in _layer_tensors(flow, kind)[1:] if t is not None]
        out, ld = MaskedStackFn.apply(z, cache['table'], cache['goff'], cache['n'], cache['n_grad'], bool(inverse), *params)
        if log_q is not None:
            return out, log_q + sign * ld
        return out, (ld if sign == 1.0 else sign * ld)
    return _lib.masked_affine_stack(z, cache['table'], cache['n'], inverse, logdet=log_q, sign=sign)


class MaskedStackFn(torch.autograd.Function):
    """(out, log_det [B]) of a planned run as ONE autograd node: forward = vcnf_masked_affine_stack_*, backward =
    vcnf_masked_affine_stack_bwd_* (keeps only ``out``).  ``params``: the run's parameters in table order (the mask
    buffers are not differentiated); their gradients come back as views of one flat buffer."""

    @staticmethod
    def forward(ctx, z, table, goff, n_layers, n_grad, inverse, *params):
        with torch.no_grad():
            out, ld = _lib.masked_affine_stack(z.detach(), table, n_layers, inverse)
        ctx.save_for_backward(out, table, goff)
        ctx.meta = (n_layers, n_grad, inverse, [tuple(p.shape) for p in params], z.requires_grad)
        return out, ld

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out, g_ld):
        out, table, goff = ctx.saved_tensors
        n_layers, n_grad, inverse, shapes, need_z = ctx.meta
        g_in, flat = _lib.masked_affine_stack_bwd(out, g_out, g_ld, table, goff, n_layers, n_grad, inverse)
        grads, at = [], 0
        for shp in shapes:
            n = 1
            for v in shp:
                n *= v
            grads.append(flat[at:at + n].view(shp))
            at += n
        return (g_in if need_z else None, None, None, None, None, None) + tuple(grads)
