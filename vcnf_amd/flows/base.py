"""Flow contract of the reference: ``forward(z) -> (z', log_det)`` is the
sampling direction, ``inverse(z)`` the density direction; ``log_det`` is a [B]
tensor, a 0-dim tensor or the Python scalar 0.
Reference: normflow/flows/base.py:6-70."""
import torch
from torch import nn


class Flow(nn.Module):
    # layers that accept ``context=`` in forward/inverse set this
    takes_context = False

    def forward(self, z):
        raise NotImplementedError('Forward pass has not been implemented.')

    def inverse(self, z):
        raise NotImplementedError('This flow has no algebraic inverse.')


class Reverse(Flow):
    """Swaps forward and inverse of the wrapped flow (base.py:24-40)."""

    def __init__(self, flow):
        super().__init__()
        self.flow = flow

    def forward(self, z):
        return self.flow.inverse(z)

    def inverse(self, z):
        return self.flow.forward(z)


class Composite(Flow):
    """Chains flows in the given order (base.py:43-70).  The running log-det
    starts on the input's device and dtype; the reference allocates it on the
    CPU (base.py:59), which fails for GPU inputs - deliberate deviation."""

    def __init__(self, flows):
        super().__init__()
        self._flows = nn.ModuleList(flows)

    @staticmethod
    def _chain(inputs, steps):
        total = torch.zeros(inputs.shape[0], dtype=inputs.dtype, device=inputs.device)
        out = inputs
        for step in steps:
            out, ld = step(out)
            total += ld
        return out, total

    def forward(self, inputs):
        return self._chain(inputs, self._flows)

    def inverse(self, inputs):
        return self._chain(inputs, [f.inverse for f in reversed(self._flows)])
