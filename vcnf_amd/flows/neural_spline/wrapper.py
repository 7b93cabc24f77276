"""User-facing neural-spline coupling layer.
Reference: normflow/flows/neural_spline/wrapper.py:15-75."""
import torch
from torch import nn

from ..base import Flow
from .coupling import PiecewiseRationalQuadraticCoupling
from ...nets.resnet import ResidualNet
from ...utils.masks import create_alternating_binary_mask


class CoupledRationalQuadraticSpline(Flow):
    """RQS coupling with a ResidualNet conditioner on an alternating mask
    (odd features transformed unless ``reverse_mask``), linear tails and the
    unconditional spline on the identity half.  normflow convention:
    ``forward`` samples, ``inverse`` evaluates density (wrapper.py:69-75).

    Extension over the reference signature: ``num_context_channels`` builds the
    conditional variant of config C3 (ResidualNet with context_features); the
    reference reaches the same module by constructing
    PiecewiseRationalQuadraticCoupling directly.  State-dict keys are identical."""
    takes_context = True

    def __init__(self, num_input_channels, num_blocks, num_hidden_channels, num_bins=8,
                 tails='linear', tail_bound=3., activation=nn.ReLU, dropout_probability=0.,
                 reverse_mask=False, num_context_channels=None):
        super().__init__()

        def make_net(in_features, out_features):
            return ResidualNet(in_features=in_features, out_features=out_features,
                               context_features=num_context_channels,
                               hidden_features=num_hidden_channels, num_blocks=num_blocks,
                               activation=activation(), dropout_probability=dropout_probability,
                               use_batch_norm=False)

        self.prqct = PiecewiseRationalQuadraticCoupling(
            mask=create_alternating_binary_mask(num_input_channels, even=reverse_mask),
            transform_net_create_fn=make_net, num_bins=num_bins, tails=tails, tail_bound=tail_bound,
            apply_unconditional_transform=True)

    def forward(self, z, context=None):
        z, log_det = self.prqct.inverse(z, context)
        return z, log_det.view(-1)

    def inverse(self, z, context=None):
        z, log_det = self.prqct(z, context)
        return z, log_det.view(-1)

    # accumulate-into-log_q forms used by NormalizingFlow (core.py:153-155 / :179-181)
    def forward_into(self, z, log_q, context=None):
        return self.prqct._run(z, context, True, log_q, -1.0)[0]

    def inverse_into(self, z, log_q, context=None):
        return self.prqct._run(z, context, False, log_q, 1.0)[0]


class CircularCoupledRationalQuadraticSpline(Flow):
    """Spline coupling in which the coordinates ``ind_circ`` are angles: their splines have circular
    tails (bound = half the period) and the conditioner sees them through periodic features
    (wrapper.py:90-187).  ``tail_bound``: a number or a per-coordinate tensor."""

    def __init__(self, num_input_channels, num_blocks, num_hidden_channels, ind_circ, num_bins=8,
                 tail_bound=3., activation=nn.ReLU, dropout_probability=0., reverse_mask=False, mask=None,
                 init_identity=True):
        super().__init__()
        import numpy as np
        from ...utils.nn import PeriodicFeatures
        from ...utils.splines import DEFAULT_MIN_DERIVATIVE
        if mask is None:
            mask = create_alternating_binary_mask(num_input_channels, even=reverse_mask)
        identity = [int(i) for i in torch.arange(num_input_channels).masked_select(torch.as_tensor(mask) <= 0)]
        circ = set(int(i) for i in ind_circ)
        circ_in_identity = [pos for pos, feat in enumerate(identity) if feat in circ]
        if torch.is_tensor(tail_bound):
            scale_pf = np.pi / tail_bound[circ_in_identity]
        else:
            scale_pf = np.pi / tail_bound

        def make_net(in_features, out_features):
            pf = PeriodicFeatures(in_features, circ_in_identity, scale_pf) if circ_in_identity else None
            net = ResidualNet(in_features=in_features, out_features=out_features, context_features=None,
                              hidden_features=num_hidden_channels, num_blocks=num_blocks, activation=activation(),
                              dropout_probability=dropout_probability, use_batch_norm=False, preprocessing=pf)
            if init_identity:
                nn.init.constant_(net.final_layer.weight, 0.)
                nn.init.constant_(net.final_layer.bias, float(np.log(np.exp(1 - DEFAULT_MIN_DERIVATIVE) - 1)))
            return net

        tails = ['circular' if i in circ else 'linear' for i in range(num_input_channels)]
        self.prqct = PiecewiseRationalQuadraticCoupling(
            mask=mask, transform_net_create_fn=make_net, num_bins=num_bins, tails=tails, tail_bound=tail_bound,
            apply_unconditional_transform=True)

    def forward(self, z):
        z, log_det = self.prqct.inverse(z)
        return z, log_det.view(-1)

    def inverse(self, z):
        z, log_det = self.prqct(z)
        return z, log_det.view(-1)


class AutoregressiveRationalQuadraticSpline(Flow):
    """Autoregressive neural spline layer (wrapper.py:197-259): ``forward`` samples (D sequential
    passes), ``inverse`` evaluates density (one pass)."""

    def __init__(self, num_input_channels, num_blocks, num_hidden_channels, num_bins=8, tail_bound=3,
                 activation=nn.ReLU, dropout_probability=0., permute_mask=False, init_identity=True):
        super().__init__()
        from .autoregressive import MaskedPiecewiseRationalQuadraticAutoregressive
        self.mprqat = MaskedPiecewiseRationalQuadraticAutoregressive(
            features=num_input_channels, hidden_features=num_hidden_channels, context_features=None,
            num_bins=num_bins, tails='linear', tail_bound=tail_bound, num_blocks=num_blocks,
            use_residual_blocks=True, random_mask=False, permute_mask=permute_mask, activation=activation(),
            dropout_probability=dropout_probability, use_batch_norm=False, init_identity=init_identity)

    def forward(self, z):
        z, log_det = self.mprqat.inverse(z)
        return z, log_det.view(-1)

    def inverse(self, z):
        z, log_det = self.mprqat(z)
        return z, log_det.view(-1)


class CircularAutoregressiveRationalQuadraticSpline(Flow):
    """Autoregressive spline layer with circular coordinates ``ind_circ`` (wrapper.py:262-330)."""

    def __init__(self, num_input_channels, num_blocks, num_hidden_channels, ind_circ, num_bins=8, tail_bound=3,
                 activation=nn.ReLU, dropout_probability=0., permute_mask=True, init_identity=True):
        super().__init__()
        from .autoregressive import MaskedPiecewiseRationalQuadraticAutoregressive
        circ = set(int(i) for i in ind_circ)
        tails = ['circular' if i in circ else 'linear' for i in range(num_input_channels)]
        self.mprqat = MaskedPiecewiseRationalQuadraticAutoregressive(
            features=num_input_channels, hidden_features=num_hidden_channels, context_features=None,
            num_bins=num_bins, tails=tails, tail_bound=tail_bound, num_blocks=num_blocks, use_residual_blocks=True,
            random_mask=False, permute_mask=permute_mask, activation=activation(),
            dropout_probability=dropout_probability, use_batch_norm=False, init_identity=init_identity)

    def forward(self, z):
        z, log_det = self.mprqat.inverse(z)
        return z, log_det.view(-1)

    def inverse(self, z):
        z, log_det = self.mprqat(z)
        return z, log_det.view(-1)
