"""Rational-quadratic-spline coupling on the HIP kernels of csrc/rqs_kernels.hip.

Reference: normflow/flows/neural_spline/coupling.py - Coupling :18-137,
PiecewiseCoupling :140-162, PiecewiseRationalQuadraticCDF :165-246,
PiecewiseRationalQuadraticCoupling :249-343.  Direction names follow that file
(nsf convention): ``forward`` is the density direction, ``inverse`` the
sampling direction; the user-facing wrapper flips them.

Per call, two kernels bracket the conditioner network:
  1. vcnf_rqs_conditioner_input_f32 - gathers the identity features (through the
     inverse unconditional spline when sampling) and appends the context;
  2. vcnf_rqs_coupling_f32 - splines on both halves, scatter through the feature
     index maps, per-sample log|det| reduction, optional accumulation into log_q.
"""
import warnings

import numpy as np
import torch
from torch import nn

from ..base import Flow
from ... import _lib, fused, fused_final, autograd
from ...utils import splines


class PiecewiseRationalQuadraticCDF(Flow):
    """Unconditional per-feature spline; its logits are shared by the whole batch
    and never leave L2/LDS inside the coupling kernel.  Stand-alone use goes
    through the elementwise kernel with the logits broadcast by stride 0."""

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1., identity_init=True,
                 min_bin_width=splines.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=splines.DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=splines.DEFAULT_MIN_DERIVATIVE):
        super().__init__()
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        if torch.is_tensor(tail_bound):
            self.register_buffer('tail_bound', tail_bound)        # coupling.py:180-181
        else:
            self.tail_bound = tail_bound
        self.tails = tails
        self.num_bins = num_bins
        self.per_feature = isinstance(tails, (list, tuple)) or torch.is_tensor(tail_bound)
        if self.per_feature and len(list(shape)) != 1:
            raise NotImplementedError("per-feature tails / tail bounds cover [B, features] inputs")
        if tails == 'linear':
            n_deriv = num_bins - 1
        elif tails == 'circular':
            n_deriv = num_bins
        else:
            n_deriv = num_bins + 1
        shape = list(shape)
        if identity_init:                              # coupling.py:194-200
            edge = float(np.log(np.exp(1 - min_derivative) - 1))
            self.unnormalized_widths = nn.Parameter(torch.zeros(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.zeros(*shape, num_bins))
            self.unnormalized_derivatives = nn.Parameter(torch.full((*shape, n_deriv), edge))
        else:                                          # coupling.py:201-205
            self.unnormalized_widths = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_derivatives = nn.Parameter(torch.rand(*shape, n_deriv))

    def logits(self):
        return (self.unnormalized_widths, self.unnormalized_heights, self.unnormalized_derivatives)

    def _cfg(self):
        return _lib.make_cfg(self.num_bins, self.tails, tail_bound=self.tail_bound,
                             min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                             min_derivative=self.min_derivative)

    def _spline_per_feature(self, inputs, inverse):
        """Tails given per feature and / or per-feature bounds (splines.py:50-66): one call of the
        batch-shared spline per group of features with the same (kind, bound)."""
        k = self.num_bins
        listed = isinstance(self.tails, (list, tuple))
        out = torch.empty_like(inputs)
        lad = 0.0
        grad = autograd.needs_grad(inputs, *self.logits())
        for kind, bound, idx in splines.feature_groups(self.tails, self.tail_bound, inputs.shape[1]):
            ix = torch.as_tensor(idx, device=inputs.device)
            cfg = _lib.make_cfg(k, kind, tail_bound=bound, min_bin_width=self.min_bin_width,
                                min_bin_height=self.min_bin_height, min_derivative=self.min_derivative)
            sl = splines.derivative_slice(kind, k) if listed else slice(None)
            args = (inputs.index_select(1, ix).contiguous(), self.unnormalized_widths[ix],
                    self.unnormalized_heights[ix], self.unnormalized_derivatives[ix][:, sl])
            if grad:
                y, ld = autograd.rqs_shared(*args, cfg, inverse=inverse)
                out = out.index_copy(1, ix, y)
            else:
                y, le = _lib.rqs_elementwise_shared(*args, cfg, inverse)
                ld = le.sum(1)
                out.index_copy_(1, ix, y)
            lad = lad + ld
        return out, lad

    def _spline(self, inputs, inverse):
        if self.per_feature:
            return self._spline_per_feature(inputs, inverse)
        if tuple(inputs.shape[1:]) != tuple(self.unnormalized_widths.shape[:-1]):
            raise ValueError('Expected inputs of shape [B, {}], got {}.'.format(
                tuple(self.unnormalized_widths.shape[:-1]), tuple(inputs.shape)))
        splines._check_bins(self.num_bins, self.min_bin_width, self.min_bin_height)
        if inputs.dtype == torch.float64:
            # fp64 models (the reference's drivers call .double()): the elementwise fp64 spline on the logit rows
            # expanded over the batch (no gradient path in fp64; csrc/rqs_f64.hip)
            uw, uh, ud = (t.unsqueeze(0).expand((inputs.shape[0],) + tuple(t.shape)) for t in self.logits())
            out, lad = _lib.rqs_elementwise(inputs, uw, uh, ud, self._cfg(), inverse)
        elif autograd.needs_grad(inputs, *self.logits()):
            return autograd.rqs_shared(inputs, *self.logits(), self._cfg(), inverse=inverse)
        else:
            # one logit row per position, shared by the batch: read in place (no [B, ...] expansion)
            out, lad = _lib.rqs_elementwise_shared(inputs, *self.logits(), self._cfg(), inverse)
        return out, torch.sum(lad, dim=list(range(1, lad.dim())))

    def forward(self, inputs, context=None):
        return self._spline(inputs, False)

    def inverse(self, inputs, context=None):
        return self._spline(inputs, True)


class PiecewiseRationalQuadraticCoupling(Flow):
    """``mask[i] > 0``: feature i is transformed by a spline whose logits come
    from ``transform_net(identity_features[, context])``; other features pass
    through (or through the unconditional spline)."""
    takes_context = True

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.,
                 apply_unconditional_transform=False, img_shape=None,
                 min_bin_width=splines.DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=splines.DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=splines.DEFAULT_MIN_DERIVATIVE):
        mask = torch.as_tensor(mask)
        if mask.dim() != 1:
            raise ValueError('Mask must be a 1-dim tensor.')
        if mask.numel() <= 0:
            raise ValueError('Mask can\'t be empty.')
        per_feature = isinstance(tails, (list, tuple)) or torch.is_tensor(tail_bound)
        if not per_feature and tails not in (None, 'linear', 'circular'):
            raise RuntimeError('{} tails are not implemented.'.format(tails))
        if per_feature and img_shape:
            raise NotImplementedError("per-feature tails / tail bounds cover [B, features] inputs")
        super().__init__()
        self.per_feature = per_feature
        self.img_shape = list(img_shape) if img_shape else None
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.features = len(mask)
        positions = torch.arange(self.features)
        self.register_buffer('identity_features', positions.masked_select(mask <= 0))
        self.register_buffer('transform_features', positions.masked_select(mask > 0))
        # coupling.py:264-278, :297-298: per-feature tails / bounds are split over the two halves
        id_tails, id_bound = tails, tail_bound
        if isinstance(tails, (list, tuple)):
            id_tails = [tails[int(i)] for i in self.identity_features]
            tails = [tails[int(i)] for i in self.transform_features]
        if torch.is_tensor(tail_bound):
            id_bound = tail_bound[self.identity_features]
            self.register_buffer('tail_bound', tail_bound[self.transform_features])
        else:
            self.tail_bound = tail_bound
        self.tails = tails
        if self.num_transform_features == 0:
            raise ValueError('Mask selects no feature to transform.')
        self.transform_net = transform_net_create_fn(
            self.num_identity_features, self.num_transform_features * self._transform_dim_multiplier())
        if apply_unconditional_transform:
            self.unconditional_transform = PiecewiseRationalQuadraticCDF(
                shape=[self.num_identity_features] + (self.img_shape or []), num_bins=num_bins, tails=id_tails,
                tail_bound=id_bound, min_bin_width=min_bin_width,
                min_bin_height=min_bin_height, min_derivative=min_derivative)
        else:
            self.unconditional_transform = None
        self._i32 = {}
        # use the single-kernel layer when shape and conditioner qualify (fused.eligible);
        # set False to force the three-step path (gather kernel, torch GEMMs, spline kernel)
        self.fused = True
        # matrix path of the fused kernel: 'fp32' (exact fp32 fma chains) or 'fp16x3'
        # (hi/lo split halves on the fp16 matrix cores, 22 significant bits, ~5x the rate)
        self.fused_precision = None      # None: vcnf_amd.fused.DEFAULT_PRECISION
        self.fused_trunk = True          # conditioner trunk in one launch where csrc/resnet_trunk.hip covers it

    @property
    def num_identity_features(self):
        return len(self.identity_features)

    @property
    def num_transform_features(self):
        return len(self.transform_features)

    def _transform_dim_multiplier(self):
        if self.tails == 'linear':
            return self.num_bins * 3 - 1
        if self.tails == 'circular':
            return self.num_bins * 3
        return self.num_bins * 3 + 1        # no tails, or a per-feature list (K+1 derivative logits each)

    # ------------------------------------------------------------ helpers
    def _index32(self, which):
        src = self.identity_features if which == 'id' else self.transform_features
        key = (which, src.device, src.data_ptr(), src._version)
        hit = self._i32.get(key)
        if hit is None:
            if len(self._i32) > 8:
                self._i32.clear()
            hit = src.to(torch.int32).contiguous()
            self._i32[key] = hit
        return hit

    def _logit_scale(self):
        # coupling.py:314-321: width/height logits are divided by sqrt(hidden width)
        for attr in ('hidden_features', 'hidden_channels'):
            if hasattr(self.transform_net, attr):
                return float(1.0 / np.sqrt(getattr(self.transform_net, attr)))
        warnings.warn('Inputs to the softmax are not scaled down: initialization might be bad.')
        return 1.0

    def _cfg(self, scaled):
        return _lib.make_cfg(self.num_bins, self.tails, tail_bound=self.tail_bound,
                             min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                             min_derivative=self.min_derivative,
                             wh_scale=self._logit_scale() if scaled else 1.0)

    def _check(self, inputs):
        if inputs.dim() not in [2, 4]:
            raise ValueError('Inputs must be a 2D or a 4D tensor.')
        if inputs.shape[1] != self.features:
            raise ValueError('Expected features = {}, got {}.'.format(self.features, inputs.shape[1]))

    def _params(self, inputs, context, sampling):
        """Conditioner call.  Sampling direction: the identity half first goes
        through the inverse unconditional spline (coupling.py:110-114)."""
        shared = self.unconditional_transform.logits() if self.unconditional_transform is not None else None
        net = self.transform_net
        fused_concat = context is not None and hasattr(net, 'trunk') and getattr(net, 'preprocessing', None) is None
        first = _lib.rqs_conditioner_input(inputs, self._index32('id'), context if fused_concat else None,
                                           shared, self._cfg(False), sampling and shared is not None)
        if fused_concat:
            return net.trunk(first, context)
        return net(first, context) if context is not None else net(first)

    def _split_index(self, device):
        """(gather, scatter) int32 index vectors of the channel partition: gathered = inputs[:, g]
        lists the identity features first, then the transform features; scattered = parts[:, s]
        puts them back.  Both directions are the HIP column-gather kernel (and each is the
        other's VJP), instead of two advanced-indexing gathers and two index_put scatters."""
        key = ('split', str(device), self.identity_features.data_ptr(), self.identity_features._version)
        hit = self._i32.get(key)
        if hit is None:
            g = torch.cat([self.identity_features, self.transform_features]).to(device)
            hit = (g.to(torch.int32).contiguous(), torch.argsort(g).to(torch.int32).contiguous())
            self._i32[key] = hit
        return hit

    def _needs_grad(self, inputs, context):
        if not torch.is_grad_enabled():
            return False
        if inputs.requires_grad or (context is not None and context.requires_grad):
            return True
        return any(p.requires_grad for p in self.parameters())

    def _run_differentiable(self, inputs, context, sampling):
        """Training path (coupling.py:70-125 as written there): the channel partition and its
        inverse are one pass each (autograd.SplitColumnsFn / MergeColumnsFn, each the other's VJP),
        the conditioner is PyTorch ops over the training GEMM kernels; the two spline families and
        their gradients are HIP kernels that read the conditioner output and the shared logits in
        place (vcnf_amd.autograd)."""
        gather, scatter = self._split_index(inputs.device)
        xi, xt = autograd.SplitColumnsFn.apply(inputs, gather, scatter, self.num_identity_features)
        lad = 0.0
        uncond = self.unconditional_transform
        if sampling and uncond is not None:
            xi, lad = uncond.inverse(xi)
        params = self.transform_net(xi, context) if context is not None else self.transform_net(xi)
        yt, lad_t = autograd.rqs_packed(xt, params, self._cfg(True), inverse=sampling)
        lad = lad + lad_t
        if (not sampling) and uncond is not None:
            xi, lad_i = uncond.forward(xi)
            lad = lad + lad_i
        out = autograd.MergeColumnsFn.apply(xi, yt, gather, scatter)
        return out, lad

    def _run_image(self, inputs, context, sampling):
        """4-D inputs [B, C, H, W], mask over channels (coupling.py:70-125 with :148-151): channel
        gather / scatter and the convolutional conditioner are PyTorch-ROCm ops; the splines read
        the conditioner output [B, C_t*P, H, W] in place through the strided elementwise kernel
        (the reference reshapes and permutes it to [B, C_t, H, W, P] first)."""
        k = self.num_bins
        uncond = self.unconditional_transform
        grad = self._needs_grad(inputs, context)
        xi = inputs[:, self.identity_features]
        xt = inputs[:, self.transform_features]
        lad = 0.0
        if sampling and uncond is not None:
            xi, lad = uncond.inverse(xi)
        params = self.transform_net(xi, context) if context is not None else self.transform_net(xi)
        if grad:
            yt, le = autograd.rqs_packed(xt.contiguous(), params, self._cfg(True), inverse=sampling)
            lad = lad + le
        else:
            yt, le = _lib.rqs_elementwise_image(xt, params, self._cfg(True), sampling)
            lad = lad + le.sum(dim=(1, 2, 3))
        if (not sampling) and uncond is not None:
            xi, l2 = uncond.forward(xi)
            lad = lad + l2
        out = torch.empty_like(inputs)
        out[:, self.identity_features] = xi
        out[:, self.transform_features] = yt
        return out, lad

    def _run_per_feature(self, inputs, context, sampling):
        """Per-feature tails / bounds (the circular NSF layers, wrapper.py:90-187): the coupling of
        coupling.py:70-125 with both spline families evaluated group by group
        (utils.splines.feature_groups)."""
        if inputs.dim() != 2:
            raise NotImplementedError("per-feature tails / tail bounds cover [B, features] inputs")
        k = self.num_bins
        uncond = self.unconditional_transform
        xi = inputs[:, self.identity_features]
        xt = inputs[:, self.transform_features]
        lad = 0.0
        if sampling and uncond is not None:
            xi, lad = uncond.inverse(xi)
        params = self.transform_net(xi, context) if context is not None else self.transform_net(xi)
        p = params.reshape(inputs.shape[0], self.num_transform_features, -1)
        scale = self._logit_scale()
        yt, le = splines.unconstrained_rational_quadratic_spline(
            xt, p[..., :k] * scale, p[..., k:2 * k] * scale, p[..., 2 * k:], inverse=sampling, tails=self.tails,
            tail_bound=self.tail_bound, min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
            min_derivative=self.min_derivative)
        lad = lad + le.sum(1)
        if (not sampling) and uncond is not None:
            xi, l2 = uncond.forward(xi)
            lad = lad + l2
        out = torch.empty_like(inputs)
        out[:, self.identity_features] = xi
        out[:, self.transform_features] = yt
        return out, lad

    def _run(self, inputs, context, sampling, log_q=None, sign=1.0):
        self._check(inputs)
        if self.per_feature or (inputs.dtype == torch.float64 and inputs.dim() == 2):
            # fp64: the plain structure of coupling.py:70-125 over the fp64 elementwise spline (csrc/rqs_f64.hip)
            out, lad = self._run_per_feature(inputs, context, sampling)
            if log_q is not None:
                return out, log_q.add_(lad, alpha=sign)
            return out, (lad if sign == 1.0 else sign * lad)
        if inputs.dim() == 4:
            out, lad = self._run_image(inputs, context, sampling)
            if log_q is not None:
                return out, log_q.add_(lad, alpha=sign)
            return out, (lad if sign == 1.0 else sign * lad)
        if self._needs_grad(inputs, context):
            out, lad = self._run_differentiable(inputs, context, sampling)
            if log_q is not None:
                return out, log_q.add_(lad, alpha=sign)
            return out, (lad if sign == 1.0 else sign * lad)
        if self.fused and fused.eligible(self, context):
            # conditioner + splines in one kernel (csrc/fused_layer.hip)
            return fused.run(self, inputs, context, sampling, log_q, sign)
        if self.fused and fused_final.eligible(self, inputs, context):
            # conditioner trunk on PyTorch-ROCm, last layer + splines in one kernel (csrc/fused_final.hip)
            return fused_final.run(self, inputs, context, sampling, log_q, sign)
        params = self._params(inputs, context, sampling)
        expect = self.num_transform_features * self._transform_dim_multiplier()
        if params.dim() != 2 or params.shape[1] != expect:
            raise ValueError('transform_net returned %s, expected [B, %d]' % (tuple(params.shape), expect))
        shared = self.unconditional_transform.logits() if self.unconditional_transform is not None else None
        return _lib.rqs_coupling(inputs, params, self._index32('tf'), self._index32('id'), shared,
                                 self._cfg(True), sampling, logdet=log_q, sign=sign)

    # ------------------------------------------------------------ nsf-convention API
    def forward(self, inputs, context=None):
        """Density direction (coupling.py:70-96)."""
        return self._run(inputs, context, False)

    def inverse(self, inputs, context=None):
        """Sampling direction (coupling.py:98-125)."""
        return self._run(inputs, context, True)
