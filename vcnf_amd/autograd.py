"""Differentiable spline op: forward = vcnf_rqs_elementwise_f32, backward =
vcnf_rqs_elementwise_bwd_f32 (csrc/rqs_backward.hip).  This is the training path of the
RQS couplings: when gradients are required the coupling layer composes
gather / conditioner / scatter in PyTorch (all differentiable) around this op, exactly the
structure of the reference (flows/neural_spline/coupling.py:70-125), with the spline
arithmetic and its gradient on the HIP kernels."""
import math

import torch
from torch.nn import functional as F

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


_ARANGE32 = {}


def _arange32(n, device):
    key = (n, str(device))
    hit = _ARANGE32.get(key)
    if hit is None:
        hit = torch.arange(n, dtype=torch.int32, device=device)
        _ARANGE32[key] = hit
    return hit


class RqsPackedFn(torch.autograd.Function):
    """Transform half of a coupling: (y, logabsdet[B]) = spline(x; params) with x [B, C, *inner]
    and the conditioner output params [B, C*P, *inner] used in place, forward and backward: the
    gradient comes back in the conditioner's own output layout (no slicing, no permute, no
    per-tensor copies)."""

    @staticmethod
    def forward(ctx, x, params, cfg, inverse):
        ctx.save_for_backward(x, params)
        ctx.cfg, ctx.inverse = cfg, inverse
        with torch.no_grad():
            if x.dim() == 2 and params.dim() == 2 and cfg.tails == _lib.TAILS_LINEAR:
                # [B, C] inputs: the coupling kernel of the inference path with all C features transformed and no
                # identity half - rows staged through LDS with 16-byte transfers (88 us at 131 072 x 32 against 186 us for
                # the generally-addressed elementwise kernel); it returns the per-sample sum of the log-derivatives
                idx = _arange32(x.shape[1], x.device)
                return _lib.rqs_coupling(x.detach(), params.detach(), idx, idx[:0], None, cfg, inverse)
            y, lad = _lib.rqs_elementwise_image(x, params, cfg, inverse, allow_grad=True)
        return y, lad.reshape(lad.shape[0], -1).sum(1)

    @staticmethod
    def backward(ctx, gy, glad):
        x, params = ctx.saved_tensors
        gx, gp = _lib.rqs_packed_bwd(x, params, gy, glad, ctx.cfg, ctx.inverse)
        return gx, gp, None, None


class RqsSharedFn(torch.autograd.Function):
    """Identity half of a coupling: per-position spline whose logits are shared by the batch.
    Backward reduces the logit gradient over the batch inside the kernel (per-thread knot
    adjoints, one Jacobian application per thread) instead of materialising [B, ..., 3K-1]."""

    @staticmethod
    def forward(ctx, x, uw, uh, ud, cfg, inverse):
        with torch.no_grad():
            y, lad = _lib.rqs_elementwise_shared(x, uw, uh, ud, cfg, inverse, allow_grad=True)
        ctx.save_for_backward(x, uw, uh, ud)
        ctx.cfg, ctx.inverse = cfg, inverse
        return y, lad.reshape(lad.shape[0], -1).sum(1)

    @staticmethod
    def backward(ctx, gy, glad):
        x, uw, uh, ud = ctx.saved_tensors
        gx, gw, gh, gd = _lib.rqs_shared_bwd(x, uw, uh, ud, gy, glad, ctx.cfg, ctx.inverse)
        return gx, gw, gh, gd, None, None


def rqs_shared(x, uw, uh, ud, cfg, inverse=False):
    """(y, logabsdet[B]) of the batch-shared spline, differentiable."""
    if cfg.num_bins in _lib.SHARED_BWD_BINS:
        return RqsSharedFn.apply(x, uw, uh, ud, cfg, inverse)
    y, lad = rqs_spline(x, uw, uh, ud, cfg, inverse)           # any K: dense expansion
    return y, lad.reshape(lad.shape[0], -1).sum(1)


def rqs_packed(x, params, cfg, inverse=False):
    """(y, logabsdet[B]) of the conditional spline on packed conditioner output, differentiable."""
    if cfg.num_bins <= 64:
        return RqsPackedFn.apply(x, params, cfg, inverse)
    raise NotImplementedError("spline VJP kernel covers up to 64 bins")


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _bc(g, like):
    """[B] upstream log-det gradient broadcast over the feature dims of ``like``."""
    return g.reshape((-1,) + (1,) * (like.dim() - 1))


def _sum_to_channels(t):
    """Sum over batch and inner dims -> [C] (per-channel parameters)."""
    return t.transpose(0, 1).reshape(t.shape[1], -1).sum(1)


# The remaining layers are one or two elementwise ops; their forward values come from the HIP
# kernels and their VJPs are the closed forms below, evaluated with device tensor ops.

class AffineCouplingFn(torch.autograd.Function):
    """vcnf_affine_coupling_f32.  With m the multiplier (exp(s) | 1/sigmoid(s+2) | sigmoid(s+2))
    and q = d log m / ds:  forward  out = z m + shift, ld = sum log m;  inverse
    out = (z - shift) / m, ld = -sum log m.  Channels outside [t_off, t_off + d_t) pass through."""

    @staticmethod
    def forward(ctx, z, param, t_off, d_t, code, inverse):
        with torch.no_grad():
            out, ld = _lib.affine_coupling(z, param, t_off, d_t, code, inverse)
        ctx.save_for_backward(z if not inverse else out, param)
        ctx.cfg = (t_off, d_t, code, inverse)
        if ld is None:
            ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        return out, ld

    @staticmethod
    def backward(ctx, g_out, g_ld):
        keep, param = ctx.saved_tensors
        t_off, d_t, code, inverse = ctx.cfg
        sl = slice(t_off, t_off + d_t)
        g_t = g_out[:, sl]
        g_z = g_out.clone()
        if code == _lib.SCALE_NONE:
            g_param = -g_t if inverse else g_t
            return g_z, g_param.contiguous(), None, None, None, None
        s = param[:, 1::2]
        if code == _lib.SCALE_EXP:
            m, q = torch.exp(s), None
        else:
            sig = torch.sigmoid(s + 2)
            m, q = (1 / sig, sig - 1) if code == _lib.SCALE_SIGMOID else (sig, 1 - sig)
        gl = _bc(g_ld, g_t)
        if not inverse:
            g_z[:, sl] = g_t * m
            g_shift = g_t
            g_s = g_t * keep[:, sl] * m + gl            # keep = z
        else:
            g_z[:, sl] = g_t / m
            g_shift = -g_z[:, sl]
            g_s = -(g_t * keep[:, sl]) - gl             # keep = out
        if q is not None:
            g_s = g_s * q
        g_param = torch.empty_like(param)
        g_param[:, 0::2] = g_shift
        g_param[:, 1::2] = g_s
        return g_z, g_param, None, None, None, None


class MaskedAffineFn(torch.autograd.Function):
    """vcnf_masked_affine_f32: out = b z + (1-b)(z e^s + t) | b z + (1-b)(z - t) e^-s, [B, D]."""

    @staticmethod
    def forward(ctx, z, s, t, bmask, inverse):
        with torch.no_grad():
            out, ld = _lib.masked_affine(z, s, t, bmask, inverse)
        ctx.save_for_backward(z, s, t, bmask, out)
        ctx.inverse = inverse
        return out, ld

    @staticmethod
    def backward(ctx, g_out, g_ld):
        z, s, t, b, out = ctx.saved_tensors
        nb = 1 - b
        gl = g_ld[:, None]
        e = torch.exp(s if not ctx.inverse else -s) if s is not None else None
        if not ctx.inverse:
            g_z = g_out * (b + nb * e) if s is not None else g_out
            g_s = nb * (g_out * z * e + gl) if s is not None else None
            g_t = nb * g_out if t is not None else None
        else:
            g_z = g_out * (b + nb * e) if s is not None else g_out
            # (1-b) out = (1-b)(z - t) e^-s
            g_s = nb * (-(g_out * out) - gl) if s is not None else None
            g_t = (-(nb * g_out * e) if s is not None else -(nb * g_out)) if t is not None else None
        return g_z, g_s, g_t, None, None


class AffineConstFn(torch.autograd.Function):
    """vcnf_affine_const_f32: out = z e^s + t | (z - t) e^-s with per-channel s, t [C]."""

    @staticmethod
    def forward(ctx, z, s, t, inverse):
        with torch.no_grad():
            out = _lib.affine_const(z, s, t, inverse)
        ctx.save_for_backward(z if not inverse else out, s)
        ctx.inverse = inverse
        return out

    @staticmethod
    def backward(ctx, g):
        keep, s = ctx.saved_tensors
        shape = (1, -1) + (1,) * (g.dim() - 2)
        if not ctx.inverse:
            e = torch.exp(s).reshape(shape)
            return g * e, _sum_to_channels(g * keep * e), _sum_to_channels(g), None
        e = torch.exp(-s).reshape(shape)
        g_z = g * e
        return g_z, -_sum_to_channels(g * keep), -_sum_to_channels(g_z), None


class PermuteFn(torch.autograd.Function):
    """vcnf_permute_f32; the VJP is the same kernel with the opposite index vector."""

    @staticmethod
    def forward(ctx, z, idx32, back32):
        ctx.back = back32
        with torch.no_grad():
            return _lib.permute(z, idx32)

    @staticmethod
    def backward(ctx, g):
        return _lib.permute(g.contiguous(), ctx.back), None, None


class SplitColumnsFn(torch.autograd.Function):
    """(z[:, gather[:first]], z[:, gather[first:]]) as two contiguous tensors in one pass (vcnf_split_columns); the VJP
    is MergeColumnsFn's forward.  Replaces a full column permutation, two slice copies and, in the backward pass, two
    zero fills, two slice writes and an add."""

    @staticmethod
    def forward(ctx, z, gather32, scatter32, first):
        ctx.scatter = scatter32
        with torch.no_grad():
            return _lib.split_columns(z, gather32, first)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ga, gb):
        return _lib.merge_columns(ga, gb, ctx.scatter), None, None, None


class MergeColumnsFn(torch.autograd.Function):
    """cat([a, b], 1)[:, scatter] in one pass (vcnf_merge_columns); the VJP is SplitColumnsFn's forward."""

    @staticmethod
    def forward(ctx, a, b, gather32, scatter32):
        ctx.gather, ctx.first = gather32, a.shape[1]
        with torch.no_grad():
            return _lib.merge_columns(a, b, scatter32)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        ga, gb = _lib.split_columns(g, ctx.gather, ctx.first)
        return ga, gb, None, None


class DiagGaussianLogProbFn(torch.autograd.Function):
    """vcnf_diag_gaussian_log_prob_f32: lp = const - sum(ls + ((z - loc) / e^ls)^2 / 2)."""

    @staticmethod
    def forward(ctx, z, loc, ls, temperature):
        with torch.no_grad():
            lp = _lib.diag_gaussian_log_prob(z, loc, ls, temperature)
        ctx.save_for_backward(z, loc, ls)
        ctx.lt = 0.0 if temperature is None else math.log(temperature)
        return lp

    @staticmethod
    def backward(ctx, g):
        z, loc, ls = ctx.saved_tensors
        v = z.reshape(len(z), -1)
        inv = torch.exp(-(ls + ctx.lt))
        u = (v - loc) * inv
        gu = g[:, None] * u
        return (-(gu * inv)).reshape(z.shape), (gu * inv).sum(0), (gu * u - g[:, None]).sum(0), None


class DiagGaussianSampleFn(torch.autograd.Function):
    """vcnf_diag_gaussian_sample_f32: z = loc + e^ls eps, lp = const - sum(ls + eps^2 / 2)."""

    @staticmethod
    def forward(ctx, eps, loc, ls, temperature):
        with torch.no_grad():
            z, lp = _lib.diag_gaussian_sample(eps, loc, ls, temperature)
        ctx.save_for_backward(eps, ls)
        ctx.lt = 0.0 if temperature is None else math.log(temperature)
        return z, lp

    @staticmethod
    def backward(ctx, g_z, g_lp):
        eps, ls = ctx.saved_tensors
        e = eps.reshape(len(eps), -1)
        gz = g_z.reshape(len(eps), -1)
        sig = torch.exp(ls + ctx.lt)
        g_eps = gz * sig - g_lp[:, None] * e
        return g_eps.reshape(eps.shape), gz.sum(0), (gz * sig * e - g_lp[:, None]).sum(0), None

# Matrix path of the conditioner's dense layers on the training path at large batches: 'fp16x3' - forward products, the
# 128 -> 128 layers' input gradients (csrc/linear_f16x3.hip) and the weight gradients (csrc/linear_wgrad.hip, split-half
# form) on fp16 split-half operands with fp32 accumulation (error against fp64 below the library's fp32 GEMM on every
# layer shape; values beyond +-65504 are clamped and counted: nf.check_saturation()) - or 'fp32': the library's fp32
# GEMMs and the exact-fp32 weight-gradient kernel.
TRAIN_MATRIX_PATH = 'fp16x3'


def _f16x3_ok(x, n_in, n_out):
    return (TRAIN_MATRIX_PATH == 'fp16x3' and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
            and x.shape[0] >= WGRAD_MIN_BATCH and bool(_lib.lib().vcnf_linear_f16x3_supported(n_in, n_out)))


def _fwd(x, weight, bias):
    """x W^T + b: split-half kernel where it wins (every forward shape of the conditioner), library otherwise."""
    if _f16x3_ok(x, weight.shape[1], weight.shape[0]):
        return _lib.linear_f16x3(x, weight, bias)
    return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()


def _dgrad(gy, weight):
    """gy W: split-half kernel for the square hidden layers (40 against 52 us at 131 072 x 128 x 128), library for the
    narrow / very deep ones, where its tiling is as fast."""
    if weight.shape[0] == weight.shape[1] and _f16x3_ok(gy, weight.shape[0], weight.shape[1]):
        return _lib.linear_f16x3(gy, weight, None, input_grad=True)
    return gy @ weight


class LinearFn(torch.autograd.Function):
    """y = x W^T + b whose weight / bias gradients come from csrc/linear_wgrad.hip (batch reduction split over the
    chip) instead of the library's output-tiled GEMM + column-sum kernel; forward and input gradient stay library GEMMs.
    Used by the conditioner's dense layers at training batch sizes (nets/resnet.py)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _fwd(x, weight, bias)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = _dgrad(gy.contiguous(), weight)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = _lib.linear_wgrad(x, gy, want_bias=ctx.has_bias and ctx.needs_input_grad[2],
                                       f16x3=TRAIN_MATRIX_PATH == 'fp16x3')
            if not ctx.needs_input_grad[1]:
                gw = None
        return gx, gw, gb


WGRAD_MIN_BATCH = 32768          # below this the library's output-tiled GEMM is as fast (16 384: 18.1 vs 19.4 ms per C3 step)


def linear(mod, x):
    """``mod(x)`` for an nn.Linear: through LinearFn when a weight gradient will be needed, the batch is large and the
    layer shape is one the weight-gradient kernel covers; the plain module call otherwise."""
    if (torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and mod.weight.requires_grad
            and x.shape[0] >= WGRAD_MIN_BATCH and type(mod) is torch.nn.Linear
            and _lib.lib().vcnf_linear_wgrad_supported(mod.in_features, mod.out_features)):
        return LinearFn.apply(x, mod.weight, mod.bias)
    return mod(x)

def _wgrad_or_torch(x, gy, want_bias=True):
    if x.shape[0] >= WGRAD_MIN_BATCH and _lib.lib().vcnf_linear_wgrad_supported(x.shape[1], gy.shape[1]):
        return _lib.linear_wgrad(x, gy, want_bias=want_bias, f16x3=TRAIN_MATRIX_PATH == 'fp16x3')
    return gy.t() @ x, (gy.sum(0) if want_bias else None)


class ResBlockFn(torch.autograd.Function):
    """One residual block of the conditioner, out = h + W1 relu(W0 relu(h) + b0) + b1 (times sigmoid(gate) when the
    block is context-gated; nets/resnet.py:38-57), as ONE autograd node: the GEMMs stay library calls (weight gradients
    through csrc/linear_wgrad.hip), every elementwise piece of the forward and of the backward is a fused map of
    csrc/resblock_ops.hip, and only relu(h), relu(a), the second Linear's output and the gate logits are kept."""

    @staticmethod
    def forward(ctx, h, gate, w0, b0, w1, b1):
        fused_relu = (_f16x3_ok(h, w0.shape[1], w0.shape[0]) and _f16x3_ok(h, w1.shape[1], w1.shape[0])
                      and bool(_lib.lib().vcnf_linear_wgrad_supported(w0.shape[1], w0.shape[0])))
        ctx.fused_relu = fused_relu
        if fused_relu:
            # both ReLUs ride on the split-half kernels: relu(h) is applied while h is read, relu(a) before a is stored -
            # neither activation gets a pass over memory of its own; h itself is kept for the backward pass (its sign
            # is relu(h)'s mask, the weight-gradient kernel applies the ReLU on load)
            t0 = h
            t1 = _lib.linear_f16x3(h, w0, b0, relu_in=True, relu_out=True)
        else:
            t0 = torch.relu(h)
            t1 = torch.relu_(_fwd(t0, w0, b0))
        c = _fwd(t1, w1, b1)
        if gate is not None:
            out = _lib.resblock_op(0, h, c, gate)
            ctx.save_for_backward(t0, t1, w0, w1, c, gate)
        else:
            out = h + c
            ctx.save_for_backward(t0, t1, w0, w1)
        ctx.gated = gate is not None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if ctx.gated:
            t0, t1, w0, w1, c, gate = ctx.saved_tensors
            g_c, g_gate = _lib.resblock_op(1, g, c, gate, two_outputs=True)
        else:
            t0, t1, w0, w1 = ctx.saved_tensors
            g_c, g_gate = g.contiguous(), None
        need = ctx.needs_input_grad
        gw1, gb1 = _wgrad_or_torch(t1, g_c) if (need[4] or need[5]) else (None, None)
        sq1 = w1.shape[0] == w1.shape[1] and _f16x3_ok(g_c, w1.shape[0], w1.shape[1])
        # square layers: the ReLU's backward (and, below, the skip connection's gradient) applied as the split-half kernel
        # stores the input gradient - no elementwise pass of their own
        g_a = _lib.linear_f16x3(g_c, w1, None, input_grad=True, mask=t1) if sq1 else _lib.resblock_op(2, _dgrad(g_c, w1), t1)
        if ctx.fused_relu:          # t0 is the block input h: relu on load
            gw0, gb0 = _lib.linear_wgrad(t0, g_a, f16x3=TRAIN_MATRIX_PATH == 'fp16x3', relu_x=True) if (need[2] or need[3]) else (None, None)
        else:
            gw0, gb0 = _wgrad_or_torch(t0, g_a) if (need[2] or need[3]) else (None, None)
        sq0 = w0.shape[0] == w0.shape[1] and _f16x3_ok(g_a, w0.shape[0], w0.shape[1])
        if not need[0]:
            g_h = None
        elif sq0:
            g_h = _lib.linear_f16x3(g_a, w0, None, input_grad=True, mask=t0, addend=g)
        else:
            g_h = _lib.resblock_op(3, _dgrad(g_a, w0), t0, g)
        return g_h, (g_gate if ctx.gated and need[1] else None), gw0, gb0, gw1, gb1


def residual_block(block, inputs, context):
    """``block(inputs, context)`` through ResBlockFn when the block is the plain training configuration this node
    covers (ReLU, no batch norm, inactive dropout, fp32 on the GPU, a batch at which the fused maps pay); else None."""
    from .fused import _is_relu
    if not (torch.is_grad_enabled() and inputs.is_cuda and inputs.dim() == 2 and inputs.dtype == torch.float32
            and inputs.shape[0] >= WGRAD_MIN_BATCH and not block.use_batch_norm and _is_relu(block.activation)
            and not (block.dropout.p > 0 and block.training)):
        return None
    l0, l1 = block.linear_layers
    if not (type(l0) is torch.nn.Linear and type(l1) is torch.nn.Linear and l0.bias is not None and l1.bias is not None):
        return None
    if not (inputs.requires_grad or l0.weight.requires_grad or l1.weight.requires_grad):
        return None
    gate = linear(block.context_layer, context) if context is not None else None
    return ResBlockFn.apply(inputs, gate, l0.weight, l0.bias, l1.weight, l1.bias)
