"""Autoregressive rational-quadratic spline flow: every variable gets its own spline whose logits a
MADE network computes from the variables of lower degree.  The density direction is one MADE pass +
one launch of the packed spline kernel (the MADE output [B, D*P] is read in place); the sampling
direction needs D sequential passes (variable i can only be inverted once its predecessors are), as
in the reference.  Per-feature tails / tensor bounds go through utils.splines' grouped evaluation.
Reference: normflow/flows/neural_spline/autoregressive.py:18-136, flows/affine/autoregressive.py:11-45."""
import numpy as np
import torch
from torch.nn import functional as F

from ..base import Flow
from ... import _lib, autograd
from ...nets.made import MADE
from ...utils import splines
from ...utils.nn import PeriodicFeatures


class MaskedPiecewiseRationalQuadraticAutoregressive(Flow):
    takes_context = True

    def __init__(self, features, hidden_features, context_features=None, num_bins=10, tails=None, tail_bound=1.,
                 num_blocks=2, use_residual_blocks=True, random_mask=False, permute_mask=False, activation=F.relu,
                 dropout_probability=0., use_batch_norm=False, init_identity=True,
                 min_bin_width=splines.DEFAULT_MIN_BIN_WIDTH, min_bin_height=splines.DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=splines.DEFAULT_MIN_DERIVATIVE):
        super().__init__()
        self.features = features
        self.num_bins = num_bins
        self.min_bin_width, self.min_bin_height, self.min_derivative = min_bin_width, min_bin_height, min_derivative
        self.tails = tails
        self.per_feature = isinstance(tails, (list, tuple)) or torch.is_tensor(tail_bound)
        preprocessing = None
        if isinstance(tails, (list, tuple)):                       # autoregressive.py:44-56
            ind_circ = [i for i in range(features) if tails[i] == 'circular']
            scale = np.pi / tail_bound[ind_circ] if torch.is_tensor(tail_bound) else np.pi / tail_bound
            preprocessing = PeriodicFeatures(features, ind_circ, scale)
        self.autoregressive_net = MADE(features=features, hidden_features=hidden_features,
                                       context_features=context_features, num_blocks=num_blocks,
                                       output_multiplier=self._output_dim_multiplier(),
                                       use_residual_blocks=use_residual_blocks, random_mask=random_mask,
                                       permute_mask=permute_mask, activation=activation,
                                       dropout_probability=dropout_probability, use_batch_norm=use_batch_norm,
                                       preprocessing=preprocessing)
        if init_identity:
            torch.nn.init.constant_(self.autoregressive_net.final_layer.weight, 0.)
            torch.nn.init.constant_(self.autoregressive_net.final_layer.bias,
                                    float(np.log(np.exp(1 - min_derivative) - 1)))
        if torch.is_tensor(tail_bound):
            self.register_buffer('tail_bound', tail_bound)
        else:
            self.tail_bound = tail_bound

    def _output_dim_multiplier(self):
        if self.tails == 'linear':
            return self.num_bins * 3 - 1
        if self.tails == 'circular':
            return self.num_bins * 3
        return self.num_bins * 3 + 1

    def _logit_scale(self):
        # autoregressive.py:104-106: scaled only if the network exposes ``hidden_features`` (the reference's
        # MADE does not, so its widths / heights stay unscaled; neither does ours)
        net = self.autoregressive_net
        return float(1.0 / np.sqrt(net.hidden_features)) if hasattr(net, 'hidden_features') else 1.0

    def _elementwise(self, inputs, params, inverse):
        if inputs.dim() != 2 or inputs.shape[1] != self.features:
            raise ValueError('Expected inputs [B, {}], got {}.'.format(self.features, tuple(inputs.shape)))
        k = self.num_bins
        splines._check_bins(k, self.min_bin_width, self.min_bin_height)
        if self.per_feature:
            p = params.view(inputs.shape[0], self.features, -1)
            sc = self._logit_scale()
            out, lad = splines.unconstrained_rational_quadratic_spline(
                inputs, p[..., :k] * sc, p[..., k:2 * k] * sc, p[..., 2 * k:], inverse=inverse, tails=self.tails,
                tail_bound=self.tail_bound, min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                min_derivative=self.min_derivative)
            return out, lad.sum(1)
        cfg = _lib.make_cfg(k, self.tails, tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
                            min_bin_height=self.min_bin_height, min_derivative=self.min_derivative,
                            wh_scale=self._logit_scale())
        if autograd.needs_grad(inputs, params):
            return autograd.rqs_packed(inputs.contiguous(), params, cfg, inverse=inverse)
        out, lad = _lib.rqs_elementwise_image(inputs, params, cfg, inverse)
        return out, lad.sum(1)

    def forward(self, inputs, context=None):
        """Density direction: one pass (flows/affine/autoregressive.py:24-27)."""
        return self._elementwise(inputs, self.autoregressive_net(inputs, context), False)

    def inverse(self, inputs, context=None):
        """Sampling direction: D passes, each fixing one more variable (:29-36)."""
        outputs = torch.zeros_like(inputs)
        logabsdet = None
        for _ in range(int(np.prod(inputs.shape[1:]))):
            outputs, logabsdet = self._elementwise(inputs, self.autoregressive_net(outputs, context), True)
        return outputs, logabsdet
