"""Split / Merge / Squeeze.  In the fused coupling blocks the channel split and
merge are pure addressing inside the HIP kernel; these modules exist for the
reference's composition API (list-of-two-tensors convention) and are views plus
one concatenation.  Checkerboard modes are a "next" row.
Reference: normflow/flows/reshape.py:9-116."""
import torch

from .base import Flow

_CHANNEL_MODES = ('channel', 'channel_inv')


class Split(Flow):
    def __init__(self, mode='channel'):
        super().__init__()
        self.mode = mode

    def _check(self):
        if self.mode not in _CHANNEL_MODES:
            if 'checkerboard' in self.mode:
                raise NotImplementedError('Mode ' + self.mode + ' is not built yet (SURVEY 8f row 2).')
            raise NotImplementedError('Mode ' + self.mode + ' is not implemented.')

    def forward(self, z):
        self._check()
        first, second = z.chunk(2, dim=1)           # first chunk takes ceil(C/2) channels
        pair = [second, first] if self.mode == 'channel_inv' else [first, second]
        return pair, 0

    def inverse(self, z):
        self._check()
        z1, z2 = z
        return torch.cat([z2, z1] if self.mode == 'channel_inv' else [z1, z2], 1), 0


class Merge(Split):
    """Split with the two directions exchanged (reshape.py:79-90)."""

    def forward(self, z):
        return Split.inverse(self, z)

    def inverse(self, z):
        return Split.forward(self, z)


class Squeeze(Flow):
    """Space-to-depth reshuffle of the multiscale architecture
    (reshape.py:93-116): forward trades 4 channels for a 2x2 pixel block."""

    def forward(self, z):
        b, c, h, w = z.shape
        z = z.view(b, c // 4, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3).contiguous()
        return z.view(b, c // 4, 2 * h, 2 * w), 0

    def inverse(self, z):
        b, c, h, w = z.shape
        z = z.view(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4).contiguous()
        return z.view(b, 4 * c, h // 2, w // 2), 0
