"""Split / Merge / Squeeze.  In the fused coupling blocks the channel split and
merge are pure addressing inside the HIP kernel; these modules exist for the
reference's composition API (list-of-two-tensors convention) and are views plus
one concatenation; the checkerboard modes are an index gather / scatter along the last
dimension.
Reference: normflow/flows/reshape.py:9-116."""
import torch

from .base import Flow

_CHANNEL_MODES = ('channel', 'channel_inv')
_CHECKER_MODES = ('checkerboard', 'checkerboard_inv')


def _checker_index(shape, inv, device):
    """Last-dimension positions of the two colours for one data point of ``shape`` (without
    batch): colour of an element = parity of the sum of its indices (reshape.py:30-38); z1
    takes the odd-parity elements ('checkerboard') or the even ones ('checkerboard_inv').
    Returns (idx1, idx2), each [*shape[:-1], shape[-1] // 2] int64."""
    if shape[-1] % 2:
        raise ValueError("checkerboard split needs an even last dimension")
    lead = torch.zeros(shape[:-1], dtype=torch.int64, device=device)
    for ax, n in enumerate(shape[:-1]):
        view = [1] * (len(shape) - 1)
        view[ax] = n
        lead = lead + torch.arange(n, device=device).view(view)
    lead = (lead % 2)[..., None]                                 # parity of the leading indices
    half = 2 * torch.arange(shape[-1] // 2, device=device)
    odd = half + (1 - lead)                                      # positions with odd total parity
    even = half + lead
    return (even, odd) if inv else (odd, even)


class Split(Flow):
    def __init__(self, mode='channel'):
        super().__init__()
        self.mode = mode

    def _check(self):
        if self.mode not in _CHANNEL_MODES + _CHECKER_MODES:
            raise NotImplementedError('Mode ' + self.mode + ' is not implemented.')

    def forward(self, z):
        self._check()
        if self.mode in _CHECKER_MODES:
            i1, i2 = _checker_index(tuple(z.shape[1:]), self.mode.endswith('inv'), z.device)
            b = z.shape[0]
            return [torch.gather(z, -1, i1.expand(b, *i1.shape)), torch.gather(z, -1, i2.expand(b, *i2.shape))], 0
        first, second = z.chunk(2, dim=1)           # first chunk takes ceil(C/2) channels
        pair = [second, first] if self.mode == 'channel_inv' else [first, second]
        return pair, 0

    def inverse(self, z):
        self._check()
        z1, z2 = z
        if self.mode in _CHECKER_MODES:
            shape = tuple(z1.shape[1:-1]) + (2 * z1.shape[-1],)
            i1, i2 = _checker_index(shape, self.mode.endswith('inv'), z1.device)
            b = z1.shape[0]
            out = z1.new_empty((b,) + shape)
            out = out.scatter(-1, i1.expand(b, *i1.shape), z1).scatter(-1, i2.expand(b, *i2.shape), z2)
            return out, 0
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
