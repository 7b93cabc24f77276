"""Glow block: affine coupling (HIP kernel, conv conditioner) -> invertible 1x1
convolution -> ActNorm.  Reference: normflow/flows/affine/glow.py:12-74."""
import torch
from torch import nn

from ..base import Flow
from .coupling import AffineCouplingBlock
from ..mixing import Invertible1x1Conv
from ..normalization import ActNorm
from ... import nets


class GlowBlock(Flow):
    def __init__(self, channels, hidden_channels, scale=True, scale_map='sigmoid',
                 split_mode='channel', leaky=0.0, init_zeros=True, use_lu=True, net_actnorm=False):
        super().__init__()
        n_par = 2 if scale else 1
        if split_mode == 'channel':
            widths = (channels // 2, hidden_channels, hidden_channels, n_par * ((channels + 1) // 2))
        elif split_mode == 'channel_inv':
            widths = ((channels + 1) // 2, hidden_channels, hidden_channels, n_par * (channels // 2))
        elif 'checkerboard' in split_mode:
            raise NotImplementedError('Mode ' + split_mode + ' is not built yet (SURVEY 8f row 2).')
        else:
            raise NotImplementedError('Mode ' + split_mode + ' is not implemented.')
        param_map = nets.ConvNet2d(widths, (3, 1, 3), leaky, init_zeros, actnorm=net_actnorm)
        self.flows = nn.ModuleList([AffineCouplingBlock(param_map, scale, scale_map, split_mode)])
        if channels > 1:
            self.flows.append(Invertible1x1Conv(channels, use_lu))
        self.flows.append(ActNorm((channels,) + (1, 1)))

    def forward(self, z):
        total = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        for flow in self.flows:
            z, log_det = flow(z)
            total += log_det
        return z, total

    def inverse(self, z):
        total = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        for flow in reversed(self.flows):
            z, log_det = flow.inverse(z)
            total += log_det
        return z, total
