"""Rational-quadratic spline, CPU oracle (test infrastructure, see package doc).

Restates ``normflow/utils/splines.py`` of the reference, keeping its operation
order so that fp32 results agree with it to rounding:

* ``count_bin``          <- ``searchsorted``                         splines.py:12-17
* ``rq_spline``          <- ``rational_quadratic_spline``            splines.py:88-193
* ``rq_spline_tails``    <- ``unconstrained_rational_quadratic_spline`` splines.py:20-85
"""
import math

import torch
import torch.nn.functional as F

MIN_BIN_WIDTH = 1e-3     # splines.py:6
MIN_BIN_HEIGHT = 1e-3    # splines.py:7
MIN_DERIVATIVE = 1e-3    # splines.py:8


def boundary_derivative_logit(min_derivative=MIN_DERIVATIVE):
    """Logit whose softplus is ``1 - min_derivative`` (splines.py:38): the
    boundary derivative of a linear-tailed spline is then exactly one."""
    return math.log(math.exp(1.0 - min_derivative) - 1.0)


def count_bin(knots, values, eps=1e-6):
    """splines.py:12-17.  Bin index = (number of knots <= value) - 1.  The
    reference bumps the LAST knot by ``eps`` in place before comparing, so a
    value sitting exactly on the right edge lands in the last bin.  The bump is
    applied to the caller's tensor here as well (the reference's later gathers
    see the bumped knot table)."""
    knots[..., -1] += eps
    return (values[..., None] >= knots).sum(dim=-1) - 1


def _partition(logits, lo, hi, floor, tensor_limits):
    """Knot positions and bin sizes from unnormalised logits.
    splines.py:109-119 (widths) and :123-133 (heights) are this same sequence:
    softmax -> floor + (1 - floor*K)*p -> cumsum -> left-pad 0 -> affine map
    onto [lo, hi] -> overwrite both ends exactly -> sizes by differencing."""
    k = logits.shape[-1]
    p = F.softmax(logits, dim=-1)
    p = floor + (1 - floor * k) * p
    cum = torch.cumsum(p, dim=-1)
    cum = F.pad(cum, pad=(1, 0), mode="constant", value=0.0)
    if tensor_limits:
        cum = (hi[..., None] - lo[..., None]) * cum + lo[..., None]
    else:
        cum = (hi - lo) * cum + lo
    cum[..., 0] = lo
    cum[..., -1] = hi
    size = cum[..., 1:] - cum[..., :-1]
    return cum, size


def _pick(table, idx):
    return table.gather(-1, idx)[..., 0]


def rq_spline(inputs, unnormalized_widths, unnormalized_heights,
              unnormalized_derivatives, inverse=False,
              left=0.0, right=1.0, bottom=0.0, top=1.0,
              min_bin_width=MIN_BIN_WIDTH, min_bin_height=MIN_BIN_HEIGHT,
              min_derivative=MIN_DERIVATIVE):
    """splines.py:88-193.  ``unnormalized_derivatives`` carries K+1 logits."""
    k = unnormalized_widths.shape[-1]
    tensor_limits = torch.is_tensor(left)                       # :99-102
    if min_bin_width * k > 1.0:                                 # :104-105
        raise ValueError("Minimal bin width too large for the number of bins")
    if min_bin_height * k > 1.0:                                # :106-107
        raise ValueError("Minimal bin height too large for the number of bins")

    xk, wk = _partition(unnormalized_widths, left, right, min_bin_width, tensor_limits)
    dk = min_derivative + F.softplus(unnormalized_derivatives)  # :121
    yk, hk = _partition(unnormalized_heights, bottom, top, min_bin_height, tensor_limits)

    # :135-138 - search the y knots when inverting, the x knots otherwise
    idx = count_bin(yk if inverse else xk, inputs)[..., None]

    x_lo = _pick(xk, idx)                                       # :140
    w = _pick(wk, idx)                                          # :141
    y_lo = _pick(yk, idx)                                       # :143
    slope = hk / wk                                             # :144
    s = _pick(slope, idx)                                       # :145
    d0 = _pick(dk, idx)                                         # :147
    d1 = _pick(dk[..., 1:], idx)                                # :148
    h = _pick(hk, idx)                                          # :150

    if inverse:
        # :153-161 quadratic a*r^2 + b*r + c = 0 in the bin coordinate r
        a = (inputs - y_lo) * (d0 + d1 - 2 * s) + h * (s - d0)
        b = h * d0 - (inputs - y_lo) * (d0 + d1 - 2 * s)
        c = -s * (inputs - y_lo)
        disc = b.pow(2) - 4 * a * c                             # :163
        assert (disc >= 0).all()                                # :164
        r = (2 * c) / (-b - torch.sqrt(disc))                   # :166
        out = r * w + x_lo                                      # :167
        rr = r * (1 - r)                                        # :169
        den = s + (d0 + d1 - 2 * s) * rr                        # :170-171
        dnum = s.pow(2) * (d1 * r.pow(2) + 2 * s * rr + d0 * (1 - r).pow(2))
        lad = torch.log(dnum) - 2 * torch.log(den)              # :175
        return out, -lad                                        # :177

    t = (inputs - x_lo) / w                                     # :179
    tt = t * (1 - t)                                            # :180
    num = h * (s * t.pow(2) + d0 * tt)                          # :182-183
    den = s + (d0 + d1 - 2 * s) * tt                            # :184-185
    out = y_lo + num / den                                      # :186
    dnum = s.pow(2) * (d1 * t.pow(2) + 2 * s * tt + d0 * (1 - t).pow(2))
    lad = torch.log(dnum) - 2 * torch.log(den)                  # :191
    return out, lad


def rq_spline_tails(inputs, unnormalized_widths, unnormalized_heights,
                    unnormalized_derivatives, inverse=False, tails="linear",
                    tail_bound=1.0, min_bin_width=MIN_BIN_WIDTH,
                    min_bin_height=MIN_BIN_HEIGHT, min_derivative=MIN_DERIVATIVE):
    """splines.py:20-85.  Elements outside [-tail_bound, tail_bound] pass
    through with zero log-det; the rest go through ``rq_spline`` after the
    derivative logits were padded according to ``tails``."""
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)   # :30
    outside = ~inside
    out = torch.zeros_like(inputs)
    lad = torch.zeros_like(inputs)

    if tails == "linear":                                       # :36-43
        ud = F.pad(unnormalized_derivatives, pad=(1, 1))
        edge = boundary_derivative_logit(min_derivative)
        ud[..., 0] = edge
        ud[..., -1] = edge
        out[outside] = inputs[outside]
        lad[outside] = 0
    elif tails == "circular":                                   # :44-49
        ud = F.pad(unnormalized_derivatives, pad=(0, 1))
        ud[..., -1] = ud[..., 0]
        out[outside] = inputs[outside]
        lad[outside] = 0
    elif isinstance(tails, (list, tuple)):                      # :50-57
        ud = unnormalized_derivatives.clone()
        lin = [t == "linear" for t in tails]
        circ = [t == "circular" for t in tails]
        edge = boundary_derivative_logit(min_derivative)
        ud[..., lin, 0] = edge
        ud[..., lin, -1] = edge
        ud[..., circ, -1] = ud[..., circ, 0]
    else:
        raise RuntimeError("{} tails are not implemented.".format(tails))  # :59

    if torch.is_tensor(tail_bound):                             # :61-66
        tb = torch.broadcast_to(tail_bound, inputs.shape)
        left = -tb[inside]
        right = tb[inside]
        bottom = -tb[inside]
        top = tb[inside]
    else:                                                       # :67-71
        left, right, bottom, top = -tail_bound, tail_bound, -tail_bound, tail_bound

    out[inside], lad[inside] = rq_spline(                       # :73-83
        inputs[inside], unnormalized_widths[inside, :], unnormalized_heights[inside, :],
        ud[inside, :], inverse=inverse, left=left, right=right, bottom=bottom, top=top,
        min_bin_width=min_bin_width, min_bin_height=min_bin_height,
        min_derivative=min_derivative)
    return out, lad
