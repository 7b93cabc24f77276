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
    """The wrapped flow with its two directions exchanged (base.py:24-40); the wrapped module
    keeps the attribute name ``flow`` (state-dict prefix)."""

    def __init__(self, flow):
        super().__init__()
        self.add_module('flow', flow)

    def forward(self, z):
        inner = self.flow
        return inner.inverse(z)

    def inverse(self, z):
        inner = self.flow
        return inner.forward(z)


class Composite(Flow):
    """Flows applied one after the other, log-dets summed (base.py:43-70); sub-modules live in
    ``_flows`` (state-dict prefix).  The running log-det starts on the input's device and dtype;
    the reference allocates it on the CPU (base.py:59), which fails for GPU inputs - deliberate
    deviation."""

    def __init__(self, flows):
        super().__init__()
        self._flows = nn.ModuleList(list(flows))

    def _run(self, z, backwards):
        log_det = z.new_zeros(z.shape[0])
        members = list(self._flows)
        for member in (reversed(members) if backwards else members):
            z, ld = (member.inverse(z) if backwards else member(z))
            log_det = log_det + ld
        return z, log_det

    def forward(self, inputs):
        return self._run(inputs, False)

    def inverse(self, inputs):
        return self._run(inputs, True)
