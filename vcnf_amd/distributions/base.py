"""Base distribution end caps of the flow.  Only the diagonal Gaussian is on the
hot path; the reference's research distributions are out of
scope.  Reference: normflow/distributions/base.py:609-652."""
import numpy as np
import torch
from torch import nn

from .. import _lib, autograd


class BaseDistribution(nn.Module):
    def forward(self, num_samples=1):
        raise NotImplementedError

    def log_prob(self, z):
        raise NotImplementedError


class DiagGaussian(BaseDistribution):
    """N(loc, diag(exp(log_scale))^2) over ``shape``; parameters [1, *shape].
    ``temperature`` T (optional) widens the scale by T, i.e. adds log T to
    log_scale (base.py:635-638)."""

    def __init__(self, shape, trainable=True):
        super().__init__()
        if isinstance(shape, int):
            shape = (shape,)
        self.shape = tuple(shape)
        self.n_dim = len(self.shape)
        self.d = int(np.prod(self.shape))
        if trainable:
            self.loc = nn.Parameter(torch.zeros(1, *self.shape))
            self.log_scale = nn.Parameter(torch.zeros(1, *self.shape))
        else:
            self.register_buffer("loc", torch.zeros(1, *self.shape))
            self.register_buffer("log_scale", torch.zeros(1, *self.shape))
        self.temperature = None

    def _flat(self):
        return self.loc.reshape(-1), self.log_scale.reshape(-1)

    def forward(self, num_samples=1):
        """Draw on the device with torch.randn, then one kernel for z and log p."""
        eps = torch.randn((num_samples,) + self.shape, dtype=self.loc.dtype, device=self.loc.device)
        return self.from_noise(eps)

    def from_noise(self, eps):
        """base.py:639-641 with the standard-normal draw supplied (parity tests
        feed the reference's captured draw; RNG streams differ across devices)."""
        loc, ls = self._flat()
        if autograd.needs_grad(eps, loc, ls):
            return autograd.DiagGaussianSampleFn.apply(eps, loc, ls, self.temperature)
        return _lib.diag_gaussian_sample(eps, loc, ls, self.temperature)

    def log_prob(self, z, out=None):
        """base.py:644-652.  ``out`` [B]: accumulate into it instead of allocating."""
        loc, ls = self._flat()
        if autograd.needs_grad(z, loc, ls, out):
            lp = autograd.DiagGaussianLogProbFn.apply(z, loc, ls, self.temperature)
            return lp if out is None else out.add_(lp)
        return _lib.diag_gaussian_log_prob(z, loc, ls, self.temperature, logp=out)
