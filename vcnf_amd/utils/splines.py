"""Functional spline API of the reference (normflow/utils/splines.py) on HIP.

Same names, argument meaning and error behaviour; tensors must live on the GPU.
Per-feature tail lists and tensor tail bounds (splines.py:50-66) are evaluated group by group:
features sharing (kind, bound) go through one uniform kernel call.
"""
import torch

from .. import _lib

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


def feature_groups(tails, tail_bound, num_features):
    """Features with the same (tail kind, tail bound) form a group that one uniform kernel call
    evaluates: [(kind, bound, [feature indices])].  ``tails``: one kind or a per-feature list;
    ``tail_bound``: a number or a per-feature tensor (splines.py:50-66)."""
    kinds = list(tails) if isinstance(tails, (list, tuple)) else [tails] * num_features
    if len(kinds) != num_features:
        raise ValueError("expected %d tail kinds, got %d" % (num_features, len(kinds)))
    if torch.is_tensor(tail_bound):
        bounds = [float(v) for v in torch.broadcast_to(tail_bound.detach().cpu(), (num_features,))]
    else:
        bounds = [float(tail_bound)] * num_features
    groups = {}
    for j, (kd, bd) in enumerate(zip(kinds, bounds)):
        if kd not in ('linear', 'circular'):
            raise RuntimeError('{} tails are not implemented.'.format(kd))
        groups.setdefault((kd, bd), []).append(j)
    return [(kd, bd, idx) for (kd, bd), idx in groups.items()]


def derivative_slice(kind, num_bins):
    """Columns of a K+1-wide derivative row a uniform call takes: with per-feature tails every
    feature carries K+1 logits; 'linear' overrides both ends with the constant (keeps 1..K-1),
    'circular' ties the last to the first (keeps 0..K-1) - splines.py:50-57."""
    return slice(1, num_bins) if kind == 'linear' else slice(0, num_bins)


def _per_feature_spline(inputs, uw, uh, ud, inverse, tails, tail_bound, min_bin_width, min_bin_height,
                        min_derivative):
    """splines.py:50-66: tails given per feature (dimension -1 of ``inputs``) and / or a tensor of
    per-feature bounds.  Features are grouped by (kind, bound); each group is one call of the uniform
    kernel on its gathered columns.  With a tails LIST the derivative rows are K+1 wide (the
    reference's layout); with one kind and a tensor bound they keep that kind's own width."""
    from .. import autograd
    num_bins = uw.shape[-1]
    _check_bins(num_bins, min_bin_width, min_bin_height)
    listed = isinstance(tails, (list, tuple))
    d = inputs.shape[-1]
    out, lad = torch.empty_like(inputs), torch.empty_like(inputs)
    grad = autograd.needs_grad(inputs, uw, uh, ud)
    for kind, bound, idx in feature_groups(tails, tail_bound, d):
        ix = torch.as_tensor(idx, device=inputs.device)
        cfg = _lib.make_cfg(num_bins, kind, tail_bound=bound, min_bin_width=min_bin_width,
                            min_bin_height=min_bin_height, min_derivative=min_derivative)
        sl = derivative_slice(kind, num_bins) if listed else slice(None)
        args = (inputs.index_select(-1, ix).contiguous(), uw.index_select(-2, ix).contiguous(),
                uh.index_select(-2, ix).contiguous(), ud.index_select(-2, ix)[..., sl].contiguous())
        if grad:
            y, ld = autograd.rqs_spline(*args, cfg, inverse=inverse)
        else:
            y, ld = _lib.rqs_elementwise(*args, cfg, inverse)
        out = out.index_copy(-1, ix, y) if grad else out.index_copy_(-1, ix, y)
        lad = lad.index_copy(-1, ix, ld) if grad else lad.index_copy_(-1, ix, ld)
    return out, lad


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
    if isinstance(tails, (list, tuple)) or torch.is_tensor(tail_bound):
        return _per_feature_spline(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives,
                                   inverse, tails, tail_bound, min_bin_width, min_bin_height, min_derivative)
    if tails not in ('linear', 'circular'):
        raise RuntimeError('{} tails are not implemented.'.format(tails))
    num_bins = unnormalized_widths.shape[-1]
    _check_bins(num_bins, min_bin_width, min_bin_height)
    cfg = _lib.make_cfg(num_bins, tails, tail_bound=tail_bound, min_bin_width=min_bin_width,
                        min_bin_height=min_bin_height, min_derivative=min_derivative)
    return _lib.rqs_elementwise(inputs, unnormalized_widths, unnormalized_heights,
                                unnormalized_derivatives, cfg, inverse)
