"""Host-side feature masks (integer data, bit-exact parity items).
Reference: normflow/utils/masks.py:5-53."""
import torch


def create_alternating_binary_mask(features, even=True):
    """uint8 [features]; ones on even positions (even=True) or odd ones.  masks.py:5-15."""
    mask = torch.zeros(features, dtype=torch.uint8)
    mask[(0 if even else 1)::2] = 1
    return mask


def create_mid_split_binary_mask(features):
    """uint8 [features]; ones on the first ceil(features / 2) positions.  masks.py:18-27."""
    mask = torch.zeros(features, dtype=torch.uint8)
    mask[:features - features // 2] = 1
    return mask


def create_random_binary_mask(features, seed=None):
    """uint8 [features] with ceil(features / 2) ones drawn without replacement by
    torch.multinomial on the CPU generator (private one when seeded).  masks.py:30-53."""
    generator = None
    if seed is not None:
        generator = torch.Generator()
        generator.manual_seed(seed)
    chosen = torch.multinomial(torch.ones(features, dtype=torch.float32), features - features // 2,
                               replacement=False, generator=generator)
    mask = torch.zeros(features, dtype=torch.uint8)
    mask[chosen] = 1
    return mask
