"""Affine bijectors on the HIP kernels of csrc/affine_kernels.hip.
Reference: normflow/flows/affine/coupling.py (AffineConstFlow :10-53,
AffineCoupling :94-168, MaskedAffineFlow :171-222, AffineCouplingBlock :225-258).
"""
import numpy as np
import torch
from torch import nn

from ..base import Flow
from ..reshape import Split, Merge
from ... import _lib, autograd, fused_affine


def _scale_code(scale, scale_map):
    if not scale:
        return _lib.SCALE_NONE
    if scale_map not in _lib.SCALE_MAPS:
        raise NotImplementedError('This scale map is not implemented.')
    return _lib.SCALE_MAPS[scale_map]


class AffineConstFlow(Flow):
    """Per-feature learned scale and shift, ``z * exp(s) + t``; ``s`` / ``t`` of
    shape [1, *shape].  log_det is the 0-dim tensor sum(s) times the number of
    positions the parameters broadcast over (coupling.py:37-53)."""

    def __init__(self, shape, scale=True, shift=True):
        super().__init__()
        zeros = torch.zeros(shape)[None]
        if scale:
            self.s = nn.Parameter(zeros.clone())
        else:
            self.register_buffer('s', zeros.clone())
        if shift:
            self.t = nn.Parameter(zeros.clone())
        else:
            self.register_buffer('t', zeros.clone())
        self.n_dim = self.s.dim()
        self.batch_dims = [i for i, n in enumerate(self.s.shape) if n == 1]

    def _broadcast_count(self, z):
        return int(np.prod([z.size(i) for i in self.batch_dims[1:]])) if len(self.batch_dims) > 1 else 1

    def _per_channel(self, z):
        # kernel form: z viewed as [B, C, inner] with parameters constant over inner
        c = self.s.shape[1]
        if self.s.numel() != c:
            raise NotImplementedError("AffineConstFlow kernels cover per-channel parameters ([C] or [C,1,1])")
        return self.s.reshape(c).contiguous(), self.t.reshape(c).contiguous()

    def _apply_const(self, z, inverse):
        s, t = self._per_channel(z)
        if autograd.needs_grad(z, s, t):
            return autograd.AffineConstFn.apply(z, s, t, inverse)
        return _lib.affine_const(z, s, t, inverse)

    def forward(self, z):
        return self._apply_const(z, False), self._broadcast_count(z) * torch.sum(self.s)

    def inverse(self, z):
        return self._apply_const(z, True), -self._broadcast_count(z) * torch.sum(self.s)


class AffineCoupling(Flow):
    """RealNVP coupling on an already split pair ``[z1, z2]``: z2 is transformed
    with shift/scale computed from z1 by ``param_map``; param channels interleave
    (shift, scale) (coupling.py:122-123)."""

    def __init__(self, param_map, scale=True, scale_map='exp'):
        super().__init__()
        self.add_module('param_map', param_map)
        self.scale = scale
        self.scale_map = scale_map

    def _run(self, z, inverse):
        z1, z2 = z
        code = _scale_code(self.scale, self.scale_map)
        param = self.param_map(z1)
        if autograd.needs_grad(z2, param):
            out, log_det = autograd.AffineCouplingFn.apply(z2, param, 0, z2.shape[1], code, inverse)
            return [z1, out], (0 if code == _lib.SCALE_NONE else log_det)
        out, log_det = _lib.affine_coupling(z2, param, 0, z2.shape[1], code, inverse)
        if code == _lib.SCALE_NONE:
            # the reference shifts z2 in place and reports the scalar 0 (coupling.py:139-141)
            z2.copy_(out)
            return [z1, z2], 0
        return [z1, out], log_det

    def forward(self, z):
        return self._run(z, False)

    def inverse(self, z):
        return self._run(z, True)


class MaskedAffineFlow(Flow):
    """f(z) = b z + (1-b)(z exp(s(b z)) + t(b z)); ``b`` is a 0/1 tensor shaped
    like one data point, stored as buffer ``b`` [1, *shape] (coupling.py:179-200)."""

    def __init__(self, b, t=None, s=None):
        super().__init__()
        self.b_cpu = b.view(1, *b.size())
        self.register_buffer('b', self.b_cpu)
        if s is None:
            self.s = None
        else:
            self.add_module('s', s)
        if t is None:
            self.t = None
        else:
            self.add_module('t', t)

    def _run(self, z, inverse):
        if z.dim() != 2:
            raise NotImplementedError("MaskedAffineFlow kernel covers [B, D] inputs")
        z_masked = self.b * z
        s = self.s(z_masked) if self.s is not None else None
        t = self.t(z_masked) if self.t is not None else None
        if autograd.needs_grad(z, s, t):
            return autograd.MaskedAffineFn.apply(z, s, t, self.b.reshape(-1).contiguous(), inverse)
        return _lib.masked_affine(z, s, t, self.b.reshape(-1).contiguous(), inverse)

    def forward(self, z):
        return self._run(z, False)

    def inverse(self, z):
        return self._run(z, True)


class AffineCouplingBlock(Flow):
    """Split + AffineCoupling + Merge.  The three sub-flows are kept (state_dict
    keys ``flows.1.param_map...``) but one kernel does the work: the channel
    split/merge is addressing, identity channels are copied through
    (coupling.py:225-258, reshape.py:25-29, :50-55)."""

    def __init__(self, param_map, scale=True, scale_map='exp', split_mode='channel'):
        super().__init__()
        self.flows = nn.ModuleList([Split(split_mode), AffineCoupling(param_map, scale, scale_map),
                                    Merge(split_mode)])
        self.split_mode = split_mode
        # single-kernel layer when the conditioner is an MLP with two equal hidden layers on [B, D]
        # inputs (fused_affine.eligible); False forces the three-step path
        self.fused = True

    # matrix path of the one-launch stack kernel for this block's conditioner: None = vcnf_amd.fused.DEFAULT_PRECISION
    # ('fp16x3': second / third dense layer on split-half operands with an on-device fp32 fallback for out-of-range
    # activations; 'fp32': exact fp32 matrix instructions throughout)
    fused_precision = None

    def fusable(self, z):
        """True when this call would take the one-kernel path (which can also absorb a Permute)."""
        if not self.fused:
            return False
        if torch.is_grad_enabled() and autograd.needs_grad(z, *self.flows[1].param_map.parameters()):
            return False
        return fused_affine.eligible(self, z)

    def run_with_permute(self, z, inverse, log_q, sign, in_gather=None, out_gather=None):
        """One-kernel layer with a neighbouring Permute folded into its load / store indexing."""
        core = self.flows[1]
        return fused_affine.run(self, z, _scale_code(core.scale, core.scale_map), inverse, log_q, sign,
                                in_gather, out_gather)[0]

    def _run(self, z, inverse, log_q=None, sign=1.0):
        if self.split_mode not in ('channel', 'channel_inv'):
            # checkerboard: explicit Split -> AffineCoupling (pair form of the kernel) -> Merge
            pair, _ = (self.flows[0].forward(z) if not inverse else self.flows[2].inverse(z))
            pair, ld = (self.flows[1].forward(pair) if not inverse else self.flows[1].inverse(pair))
            out, _ = (self.flows[2].forward(pair) if not inverse else self.flows[0].inverse(pair))
            if not torch.is_tensor(ld):
                ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
            if log_q is not None:
                return out, log_q.add_(ld, alpha=sign)
            return out, (ld if sign == 1.0 else sign * ld)
        core = self.flows[1]
        code = _scale_code(core.scale, core.scale_map)
        if (self.fused and not (torch.is_grad_enabled() and autograd.needs_grad(z, log_q, *core.param_map.parameters()))
                and fused_affine.eligible(self, z)):
            # conditioner + affine map + log|det| in one kernel (csrc/fused_affine.hip)
            return fused_affine.run(self, z, code, inverse, log_q, sign)
        c = z.shape[1]
        head = c - c // 2                              # chunk(2): first chunk has ceil(C/2) channels
        if self.split_mode == 'channel':
            cond_in, t_off, d_t = z[:, :head], head, c - head
        else:
            cond_in, t_off, d_t = z[:, head:], 0, head
        if z.dim() > 2:
            cond_in = cond_in.contiguous()
        param = core.param_map(cond_in)
        if autograd.needs_grad(z, param, log_q):
            out, ld = autograd.AffineCouplingFn.apply(z, param, t_off, d_t, code, inverse)
            if log_q is not None:
                return out, log_q.add_(ld, alpha=sign)
            return out, (ld if sign == 1.0 else sign * ld)
        ld = log_q
        if ld is None:
            ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)   # coupling.py:248
        out, ld = _lib.affine_coupling(z, param, t_off, d_t, code, inverse, logdet=ld, sign=sign)
        return out, ld

    def forward(self, z):
        return self._run(z, False)

    def inverse(self, z):
        return self._run(z, True)

    # accumulate-into-log_q forms used by NormalizingFlow (core.py:153-155 / :179-181)
    def forward_into(self, z, log_q):
        return self._run(z, False, log_q, -1.0)[0]

    def inverse_into(self, z, log_q):
        return self._run(z, True, log_q, 1.0)[0]
