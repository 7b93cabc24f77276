"""Channel permutations as a HIP column gather.
Reference: normflow/flows/mixing.py:10-54 (Permute)."""
import torch

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

    def forward(self, z):
        return _lib.permute(z, self._idx32(False, z.device)), 0

    def inverse(self, z):
        return _lib.permute(z, self._idx32(True, z.device)), 0
