"""Functional spline API of the reference (normflow/utils/splines.py) on HIP.

Same names, argument meaning and error behaviour; tensors must live on the GPU.
Not built: per-feature tail lists, tensor
tail bounds - these raise like an unknown ``tails`` value does in the reference.
"""
import torch

from .. import _lib

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


def _check_bins(num_bins, min_bin_width, min_bin_height):
    # splines.py:104-107
    if min_bin_width * num_bins > 1.0:
        raise ValueError("Minimal bin width too large for the number of bins")
    if min_bin_height * num_bins > 1.0:
        raise ValueError("Minimal bin height too large for the number of bins")


def rational_quadratic_spline(inputs, unnormalized_widths, unnormalized_heights,
                              unnormalized_derivatives, inverse=False,
                              left=0., right=1., bottom=0., top=1.,
                              min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                              min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                              min_derivative=DEFAULT_MIN_DERIVATIVE):
    """splines.py:88-193: spline on [left, right] -> [bottom, top] with K+1
    derivative logits per element.  Inputs outside the interval are evaluated in
    the nearest edge bin (the reference indexes out of range there)."""
    if torch.is_tensor(left):
        raise NotImplementedError("tensor interval limits are not built (SURVEY 8f row 4)")
    num_bins = unnormalized_widths.shape[-1]
    _check_bins(num_bins, min_bin_width, min_bin_height)
    cfg = _lib.make_cfg(num_bins, None, left=left, right=right, bottom=bottom, top=top,
                        min_bin_width=min_bin_width, min_bin_height=min_bin_height,
                        min_derivative=min_derivative)
    return _lib.rqs_elementwise(inputs, unnormalized_widths, unnormalized_heights,
                                unnormalized_derivatives, cfg, inverse)


def unconstrained_rational_quadratic_spline(inputs, unnormalized_widths, unnormalized_heights,
                                            unnormalized_derivatives, inverse=False,
                                            tails='linear', tail_bound=1.,
                                            min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                                            min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                                            min_derivative=DEFAULT_MIN_DERIVATIVE):
    """splines.py:20-85: identity and zero log-det outside [-tail_bound, tail_bound]; 'linear'
    tails take K-1 derivative logits per element (boundary derivatives fixed at 1), 'circular'
    tails K (the last knot shares the first knot's derivative, :44-49)."""
    if tails not in ('linear', 'circular'):
        raise RuntimeError('{} tails are not implemented.'.format(tails))
    if torch.is_tensor(tail_bound):
        raise NotImplementedError("tensor tail bounds are not built (SURVEY 8f row 4)")
    num_bins = unnormalized_widths.shape[-1]
    _check_bins(num_bins, min_bin_width, min_bin_height)
    cfg = _lib.make_cfg(num_bins, tails, tail_bound=tail_bound, min_bin_width=min_bin_width,
                        min_bin_height=min_bin_height, min_derivative=min_derivative)
    return _lib.rqs_elementwise(inputs, unnormalized_widths, unnormalized_heights,
                                unnormalized_derivatives, cfg, inverse)
