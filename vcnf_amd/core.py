"""Model container: the log_prob / sample loops over a list of flows.
Reference: normflow/core.py - NormalizingFlow.log_prob :170-183, .sample
:144-168, save/load :185-197.

Layers that expose ``inverse_into`` / ``forward_into`` add their log|det|
straight into the running ``log_q`` inside the coupling kernel (one [B] buffer
for the whole stack, no per-layer allocation or add); every other Flow goes
through the plain ``(z, log_det)`` contract.

Training objectives (forward_kld / reverse_kld / reverse_alpha_div) need the
VJP kernels of SURVEY 8f row 1 and raise until those exist.  The reference's
categorical-dequantisation branch is out of scope; ``categoricals`` is always
defined (None) so ``sample`` follows core.py:150-155 instead of failing with
the AttributeError the reference has at this HEAD.
"""
import torch
from torch import nn


class NormalizingFlow(nn.Module):
    def __init__(self, q0, flows, p=None, categoricals=None, catlevels=None, catvdeqs=None):
        super().__init__()
        if categoricals is not None:
            raise NotImplementedError("variational dequantisation of categorical columns is out of scope")
        self.q0 = q0
        self.flows = nn.ModuleList(flows)
        self.p = p
        self.categoricals = None

    # ------------------------------------------------------------ density
    def log_prob(self, x, context=None):
        """log q(x) [B]: flows inverted last to first, log-dets added, base
        log-density at the end (core.py:176-183)."""
        log_q = torch.zeros(len(x), dtype=x.dtype, device=x.device)
        z = x
        for flow in reversed(self.flows):
            ctx = {'context': context} if (context is not None and getattr(flow, 'takes_context', False)) else {}
            if hasattr(flow, 'inverse_into'):
                z = flow.inverse_into(z, log_q, **ctx)
            else:
                z, log_det = flow.inverse(z, **ctx)
                log_q += log_det
        if hasattr(self.q0, 'from_noise'):
            self.q0.log_prob(z, out=log_q)
        else:
            log_q += self.q0.log_prob(z)
        return log_q

    # ------------------------------------------------------------ sampling
    def sample(self, num_samples=1, context=None):
        """(z, log q(z)) for fresh base draws (core.py:150-155, :168)."""
        z, log_q = self.q0(num_samples)
        return self._push(z, log_q, context)

    def sample_from(self, eps, context=None):
        """Same as ``sample`` with the standard-normal base draw given."""
        z, log_q = self.q0.from_noise(eps)
        return self._push(z, log_q, context)

    def _push(self, z, log_q, context):
        for flow in self.flows:
            ctx = {'context': context} if (context is not None and getattr(flow, 'takes_context', False)) else {}
            if hasattr(flow, 'forward_into'):
                z = flow.forward_into(z, log_q, **ctx)
            else:
                z, log_det = flow(z, **ctx)
                log_q -= log_det
        return z, log_q

    # ------------------------------------------------------------ objectives (next row)
    def forward_kld(self, x, extended=False):
        raise NotImplementedError("training objectives need the VJP kernels (SURVEY 8f row 1)")

    def reverse_kld(self, num_samples=1, beta=1., score_fn=True, extended=False):
        raise NotImplementedError("training objectives need the VJP kernels (SURVEY 8f row 1)")

    def reverse_alpha_div(self, num_samples=1, alpha=1, dreg=False, extended=False):
        raise NotImplementedError("training objectives need the VJP kernels (SURVEY 8f row 1)")

    # ------------------------------------------------------------ checkpoints
    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        self.load_state_dict(torch.load(path, weights_only=True))
