"""Feature-mask and permutation index builders, CPU oracle (test infrastructure).

Restates ``normflow/utils/masks.py`` and the index arithmetic of
``normflow/flows/mixing.py`` (Permute :10-54, _Permutation :203-234).  These are
integer results: parity is bit-exact.
"""
import torch


def alternating_mask(features, even=True):
    """masks.py:5-15: uint8 ones at ``start::2`` with start 0 (even) or 1."""
    m = torch.zeros(features, dtype=torch.uint8)
    m[(0 if even else 1)::2] += 1
    return m


def mid_split_mask(features):
    """masks.py:18-27: ones on the first ceil(features/2) entries."""
    m = torch.zeros(features, dtype=torch.uint8)
    m[:(features + 1) // 2] += 1
    return m


def random_mask(features, seed=None):
    """masks.py:30-53: ceil(features/2) ones at positions drawn without
    replacement by ``torch.multinomial`` over uniform weights; an explicit
    ``seed`` uses a private generator, otherwise the global RNG stream."""
    m = torch.zeros(features, dtype=torch.uint8)
    gen = None
    if seed is not None:
        gen = torch.Generator()
        gen.manual_seed(seed)
    picks = torch.multinomial(torch.ones(features).float(), (features + 1) // 2,
                              replacement=False, generator=gen)
    m[picks] += 1
    return m


def split_features(mask):
    """neural_spline/coupling.py:43-46: (identity, transform) index vectors;
    ``mask > 0`` marks transformed features."""
    mask = torch.as_tensor(mask)
    ar = torch.arange(len(mask))
    return ar.masked_select(mask <= 0), ar.masked_select(mask > 0)


def shuffle_perm(num_channels):
    """mixing.py:25-30: ``perm = randperm`` (global RNG) and its inverse built
    by scattering arange through perm."""
    perm = torch.randperm(num_channels)
    inv = torch.empty_like(perm).scatter_(0, perm, torch.arange(num_channels))
    return perm, inv


def swap_index(num_channels, inverse=False):
    """mixing.py:35-38 / :47-50 as a gather index: forward moves the block
    starting at floor(C/2) to the front, inverse the block at ceil(C/2)."""
    cut = (num_channels + 1) // 2 if inverse else num_channels // 2
    ar = torch.arange(num_channels)
    return torch.cat([ar[cut:], ar[:cut]])
