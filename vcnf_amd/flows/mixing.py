"""Channel mixing layers.  Permutations are a HIP column gather.  The invertible 1x1 convolution and the
LU-parameterised linear layer are dense C x C maps with a parameter-only log-det: at inference they run on
csrc/channel_mix.hip (exact fp32 matrix instructions, one pass over the activations; inside a GlowBlock the 1x1
convolution is composed with the ActNorm next to it, flows/affine/glow.py) for C a multiple of 4 up to 64; other
widths and the differentiable path use PyTorch-ROCm library calls.
Reference: normflow/flows/mixing.py:10-54 (Permute), :57-128 (Invertible1x1Conv), :352-492 (_LULinear)."""
import torch
from torch import nn
from torch.nn import functional as F

from .base import Flow
from .. import _lib, autograd


class Permute(Flow):
    """``shuffle``: fixed random permutation drawn with torch.randperm at
    construction (buffers ``perm`` / ``inv_perm``, int64, part of the state
    dict); ``swap``: rotate the channel halves - forward moves the block that
    starts at floor(C/2) to the front, inverse the block at ceil(C/2)."""

    def __init__(self, num_channels, mode='shuffle'):
        super().__init__()
        self.mode = mode
        self.num_channels = num_channels
        if mode == 'shuffle':
            perm = torch.randperm(num_channels)
            inv_perm = torch.empty_like(perm)
            inv_perm[perm] = torch.arange(num_channels)
            self.register_buffer("perm", perm)
            self.register_buffer("inv_perm", inv_perm)
        self._idx_cache = {}

    def gather_index(self, inverse):
        """int64 CPU/any-device index vector such that out[:, j] = z[:, idx[j]]."""
        if self.mode == 'shuffle':
            return self.inv_perm if inverse else self.perm
        if self.mode == 'swap':
            c = self.num_channels
            cut = (c + 1) // 2 if inverse else c // 2
            ar = torch.arange(c)
            return torch.cat([ar[cut:], ar[:cut]])
        raise NotImplementedError('The mode ' + self.mode + ' is not implemented.')

    def _idx32(self, inverse, device):
        src = self.gather_index(inverse)
        key = (bool(inverse), str(device), src.data_ptr() if self.mode == 'shuffle' else 0,
               src._version if self.mode == 'shuffle' else 0)
        hit = self._idx_cache.get(key)
        if hit is None:
            if len(self._idx_cache) > 8:
                self._idx_cache.clear()
            hit = src.to(device=device, dtype=torch.int32).contiguous()
            self._idx_cache[key] = hit
        return hit

    def _gather(self, z, inverse):
        if torch.is_grad_enabled() and z.requires_grad:
            from .. import autograd
            return autograd.PermuteFn.apply(z, self._idx32(inverse, z.device), self._idx32(not inverse, z.device))
        return _lib.permute(z, self._idx32(inverse, z.device))

    def forward(self, z):
        return self._gather(z, False), 0

    def inverse(self, z):
        return self._gather(z, True), 0


class Invertible1x1Conv(Flow):
    """Glow's invertible 1x1 convolution on NCHW inputs.  ``use_lu``: W = P L U with a
    fixed permutation P, unit-lower L, upper U whose diagonal is sign_S * exp(log_S)
    (mixing.py:71-84); log|det| = sum(log_S) per pixel."""

    def __init__(self, num_channels, use_lu=False):
        super().__init__()
        self.num_channels = num_channels
        self.use_lu = use_lu
        q = torch.linalg.qr(torch.randn(num_channels, num_channels))[0]
        if use_lu:
            p, l, u = torch.linalg.lu(q)
            diag = u.diag()
            self.register_buffer('P', p)
            self.L = nn.Parameter(l)
            self.register_buffer('sign_S', torch.sign(diag))
            self.log_S = nn.Parameter(torch.log(torch.abs(diag)))
            self.U = nn.Parameter(torch.triu(u, diagonal=1))
            self.register_buffer('eye', torch.diag(torch.ones(num_channels)))
        else:
            self.W = nn.Parameter(q)

    def _assemble_W(self, inverse=False):
        lower = torch.tril(self.L, diagonal=-1) + self.eye
        upper = torch.triu(self.U, diagonal=1) + torch.diag(self.sign_S * torch.exp(self.log_S))
        if not inverse:
            return self.P @ lower @ upper
        # the reference inverts in fp64 and casts back (mixing.py:90-95)
        dt = self.log_S.dtype
        return (torch.inverse(upper.double()) @ torch.inverse(lower.double())).to(dt) @ self.P.t()

    def _conv(self, z, inverse_weight):
        if self.use_lu:
            w = self._assemble_W(inverse=inverse_weight)
            log_det = torch.sum(self.log_S)
        else:
            w = torch.inverse(self.W.double()).to(self.W.dtype) if inverse_weight else self.W
            log_det = torch.slogdet(self.W)[1]
        if inverse_weight:
            log_det = -log_det
        out = F.conv2d(z, w.view(self.num_channels, self.num_channels, 1, 1))
        return out, log_det * z.size(2) * z.size(3)

    def forward(self, z):
        return self._conv(z, True)          # sampling direction applies W^-1 (mixing.py:100-116)

    def inverse(self, z):
        return self._conv(z, False)


class _RandomPermutation(Flow):
    """Fixed random permutation of dimension 1, buffer ``_permutation`` (mixing.py:198-235).
    Stand-alone it is the HIP column gather; inside LULinearPermute it is folded into the
    weight matrix."""

    def __init__(self, features):
        super().__init__()
        self.register_buffer('_permutation', torch.randperm(features))
        self._idx_cache = {}

    def _idx32(self, inverse, device):
        src = self._permutation
        key = (bool(inverse), str(device), src.data_ptr(), src._version)
        hit = self._idx_cache.get(key)
        if hit is None:
            if len(self._idx_cache) > 8:
                self._idx_cache.clear()
            idx = torch.argsort(src) if inverse else src
            hit = idx.to(device=device, dtype=torch.int32).contiguous()
            self._idx_cache[key] = hit
        return hit

    def _gather(self, z, inverse):
        if z.dim() != 2 or z.shape[1] != len(self._permutation):
            raise ValueError("Dimension 1 in inputs must be of size {}.".format(len(self._permutation)))
        if torch.is_grad_enabled() and z.requires_grad:
            from .. import autograd
            out = autograd.PermuteFn.apply(z, self._idx32(inverse, z.device), self._idx32(not inverse, z.device))
        else:
            out = _lib.permute(z, self._idx32(inverse, z.device))
        return out, torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)

    def forward(self, inputs, context=None):
        return self._gather(inputs, False)

    def inverse(self, inputs, context=None):
        return self._gather(inputs, True)


class _LULinear(Flow):
    """y = x (L U)^T + bias with unit-lower L and upper U whose diagonal is softplus(.) + eps
    (mixing.py:352-470).  State-dict keys as in the reference: ``bias``, ``lower_entries``,
    ``upper_entries``, ``unconstrained_upper_diag``.

    Dense D x D contraction: a library GEMM on PyTorch-ROCm (not a hand-kernel
    target).  Where the reference runs two triangular products / solves per call, this runs ONE
    GEMM against W = L U (or its inverse, formed in fp64 by two triangular solves of the
    identity and cast back), optionally with a column permutation folded in; the matrices are
    cached per parameter version when no gradient is required."""

    def __init__(self, features, using_cache=False, identity_init=True, eps=1e-3):
        super().__init__()
        import numpy as np
        self.features = features
        self.eps = eps
        self.using_cache = using_cache
        self.bias = nn.Parameter(torch.zeros(features))
        n_tri = ((features - 1) * features) // 2
        self.lower_entries = nn.Parameter(torch.zeros(n_tri))
        self.upper_entries = nn.Parameter(torch.zeros(n_tri))
        self.unconstrained_upper_diag = nn.Parameter(torch.zeros(features))
        li, ui = np.tril_indices(features, k=-1), np.triu_indices(features, k=1)
        self._tri = (torch.as_tensor(li[0]), torch.as_tensor(li[1]), torch.as_tensor(ui[0]), torch.as_tensor(ui[1]))
        if identity_init:
            nn.init.constant_(self.unconstrained_upper_diag, float(np.log(np.exp(1 - eps) - 1)))
        else:
            stdv = 1.0 / np.sqrt(features)
            for p in (self.lower_entries, self.upper_entries, self.unconstrained_upper_diag):
                nn.init.uniform_(p, -stdv, stdv)
        self._mats = {}

    @property
    def upper_diag(self):
        return F.softplus(self.unconstrained_upper_diag) + self.eps

    def _create_lower_upper(self):
        d, dev = self.features, self.lower_entries.device
        l0, l1, u0, u1 = (t.to(dev) for t in self._tri)
        diag = torch.arange(d, device=dev)
        lower = self.lower_entries.new_zeros(d, d).index_put((l0, l1), self.lower_entries)
        lower = lower.index_put((diag, diag), self.lower_entries.new_ones(d))
        upper = self.upper_entries.new_zeros(d, d).index_put((u0, u1), self.upper_entries)
        upper = upper.index_put((diag, diag), self.upper_diag)
        return lower, upper

    def weight(self):
        lower, upper = self._create_lower_upper()
        return lower @ upper

    def weight_inverse(self):
        lower, upper = self._create_lower_upper()
        eye = torch.eye(self.features, dtype=torch.float64, device=lower.device)
        linv = torch.linalg.solve_triangular(lower.double(), eye, upper=False, unitriangular=True)
        return torch.linalg.solve_triangular(upper.double(), linv, upper=True).to(lower.dtype)

    def logabsdet(self):
        return torch.sum(torch.log(self.upper_diag))

    def _matrix(self, inverse, col_index=None):
        """W^T (or W^-T) with an optional permutation folded in, cached when no grad is needed."""
        params = (self.lower_entries, self.upper_entries, self.unconstrained_upper_diag)
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        key = (bool(inverse), None if col_index is None else (col_index.data_ptr(), col_index._version),
               str(params[0].device)) + tuple((p.data_ptr(), p._version) for p in params)
        if not grad and key in self._mats:
            return self._mats[key]
        m = (self.weight_inverse() if inverse else self.weight()).t()
        if col_index is not None:
            # forward map then gather columns: (x M)[:, idx] = x M[:, idx];
            # gather columns then forward map: x[:, idx] M = x M[argsort(idx), :]
            m = m[:, col_index] if inverse else m[torch.argsort(col_index), :]
        m = m.contiguous()
        if not grad:
            if len(self._mats) > 4:
                self._mats.clear()
            self._mats[key] = m.detach()
        return m

    def _mix_operands(self, inverse, col_index):
        """(M, v) of csrc/channel_mix.hip for this direction: out = x M^T + v, cached like ``_matrix``."""
        params = (self.lower_entries, self.upper_entries, self.unconstrained_upper_diag, self.bias)
        key = ('mix', bool(inverse), None if col_index is None else (col_index.data_ptr(), col_index._version),
               str(params[0].device)) + tuple((p.data_ptr(), p._version) for p in params)
        hit = self._mats.get(key)
        if hit is None:
            with torch.no_grad():
                m = self._matrix(inverse, col_index)                   # out = (x - bias) m   or   x m + bias
                vec = -(self.bias.double() @ m.double()).float() if inverse else self.bias.detach().clone()
                hit = (m.t().contiguous(), vec.contiguous())
            if len(self._mats) > 8:
                self._mats.clear()
            self._mats[key] = hit
        return hit

    def _apply_linear(self, inputs, inverse, col_index=None):
        params = (self.lower_entries, self.upper_entries, self.unconstrained_upper_diag, self.bias)
        if (self.fused_mix and inputs.dim() == 2 and inputs.is_cuda and inputs.dtype == torch.float32
                and not autograd.needs_grad(inputs, *params)
                and _lib.lib().vcnf_channel_mix_supported(self.features)):
            # one launch on the exact-fp32 matrix instructions instead of a library GEMM + elementwise kernels
            mat, vec = self._mix_operands(inverse, col_index)
            out = _lib.channel_mix(inputs, mat, vec)
            ld = -self.logabsdet() if inverse else self.logabsdet()
            return out, ld * inputs.new_ones(out.shape[0])
        m = self._matrix(inverse, col_index)
        if inverse:
            out = (inputs - self.bias) @ m
            ld = -self.logabsdet()
        else:
            out = torch.addmm(self.bias, inputs, m)
            ld = self.logabsdet()
        return out, ld * inputs.new_ones(out.shape[0])

    fused_mix = True

    def forward(self, inputs, context=None):
        return self._apply_linear(inputs, False)

    def inverse(self, inputs, context=None):
        return self._apply_linear(inputs, True)


class LULinearPermute(Flow):
    """Fixed permutation + LU-parameterised linear map between spline couplings
    (mixing.py:473-492, arXiv 1906.04032).  ``forward`` = sampling direction: inverse linear
    map, then inverse permutation; ``inverse`` = permutation, then linear map.  Each direction
    is one GEMM: the permutation is folded into the matrix."""

    def __init__(self, num_channels, identity_init=True):
        super().__init__()
        self.permutation = _RandomPermutation(num_channels)
        self.linear = _LULinear(num_channels, identity_init=identity_init)

    def forward(self, z):
        inv_perm = torch.argsort(self.permutation._permutation)
        z, log_det = self.linear._apply_linear(z, True, inv_perm)
        return z, log_det.view(-1)

    def inverse(self, z):
        z, log_det = self.linear._apply_linear(z, False, self.permutation._permutation)
        return z, log_det.view(-1)
