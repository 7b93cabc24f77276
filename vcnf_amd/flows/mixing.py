"""Channel mixing layers.  Permutations are a HIP column gather; the invertible 1x1
convolution is a dense C x C contraction with a parameter-only log-det and runs on
PyTorch-ROCm (not a hand-kernel target).
Reference: normflow/flows/mixing.py:10-54 (Permute), :57-128 (Invertible1x1Conv)."""
import torch
from torch import nn
from torch.nn import functional as F

from .base import Flow
from .. import _lib


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
