"""Differentiable spline op: forward = vcnf_rqs_elementwise_f32, backward =
vcnf_rqs_elementwise_bwd_f32 (csrc/rqs_backward.hip).  This is the training path of the
RQS couplings: when gradients are required the coupling layer composes
gather / conditioner / scatter in PyTorch (all differentiable) around this op, exactly the
structure of the reference (flows/neural_spline/coupling.py:70-125), with the spline
arithmetic and its gradient on the HIP kernels."""
import torch

from . import _lib


class RqsSplineFn(torch.autograd.Function):
    """(y, logabsdet) = spline(x; uw, uh, ud), elementwise.  ``wh_scale`` is already folded
    into ``cfg`` (it multiplies the width / height logits inside the kernels)."""

    @staticmethod
    def forward(ctx, x, uw, uh, ud, cfg, inverse):
        with torch.no_grad():
            y, lad = _lib.rqs_elementwise(x, uw, uh, ud, cfg, inverse, allow_grad=True)
        ctx.save_for_backward(x, uw, uh, ud)
        ctx.cfg, ctx.inverse = cfg, inverse
        return y, lad

    @staticmethod
    def backward(ctx, gy, glad):
        x, uw, uh, ud = ctx.saved_tensors
        gx, gw, gh, gd = _lib.rqs_elementwise_bwd(x, uw, uh, ud, gy.contiguous(), glad.contiguous(),
                                                  ctx.cfg, ctx.inverse)
        return gx, gw, gh, gd, None, None


def rqs_spline(x, uw, uh, ud, cfg, inverse=False):
    return RqsSplineFn.apply(x, uw.expand(x.shape + uw.shape[-1:]), uh.expand(x.shape + uh.shape[-1:]),
                             ud.expand(x.shape + ud.shape[-1:]), cfg, inverse)


class _HipForwardTorchBackward(torch.autograd.Function):
    """Forward on a HIP kernel, backward by re-evaluating a PyTorch restatement of the same
    (cheap, elementwise) op on the device and differentiating that.  Used for the end caps
    and the affine layers, whose VJPs are two or three elementwise ops; the spline VJP has
    its own kernel (RqsSplineFn)."""

    @staticmethod
    def forward(ctx, hip_fn, torch_fn, *tensors):
        with torch.no_grad():
            outs = hip_fn(*[t.detach() for t in tensors])
        ctx.torch_fn = torch_fn
        ctx.save_for_backward(*tensors)
        ctx.single = torch.is_tensor(outs)
        return outs if ctx.single else tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        needs = ctx.needs_input_grad[2:]
        leaves = [t.detach().requires_grad_(n) for t, n in zip(ctx.saved_tensors, needs)]
        with torch.enable_grad():
            outs = ctx.torch_fn(*leaves)
        outs = [outs] if torch.is_tensor(outs) else list(outs)
        pairs = [(o, g) for o, g in zip(outs, gouts) if g is not None and o.requires_grad]
        want = [l for l, n in zip(leaves, needs) if n]
        grads = iter(torch.autograd.grad([o for o, _ in pairs], want, [g for _, g in pairs], allow_unused=True)
                     if pairs and want else [None] * len(want))
        return (None, None) + tuple(next(grads) if n else None for n in needs)


def hip_forward(hip_fn, torch_fn, *tensors):
    """``hip_fn(*tensors)`` with gradients defined by ``torch_fn(*tensors)``."""
    return _HipForwardTorchBackward.apply(hip_fn, torch_fn, *tensors)


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
