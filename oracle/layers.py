"""Bijector layers and the log_prob / sample loops, CPU oracle (test
infrastructure, see package doc).  Plain Python objects holding tensors; every
layer exposes the reference's convention ``forward(z) -> (z', log_det)``
(sampling direction) and ``inverse(z)`` (density direction).

Restated reference code (paths relative to /root/reference/normflow):
  flows/neural_spline/coupling.py  Coupling :70-125, PiecewiseCoupling :147-159,
      PiecewiseRationalQuadraticCDF :211-246, PiecewiseRationalQuadraticCoupling :309-343
  flows/neural_spline/wrapper.py   CoupledRationalQuadraticSpline :69-75 (direction flip)
  flows/affine/coupling.py         AffineConstFlow :37-53, AffineCoupling :113-168,
      MaskedAffineFlow :202-222, AffineCouplingBlock :247-258
  flows/reshape.py                 Split / Merge channel modes :25-29, :50-55, checkerboard :30-42, :56-72
  flows/mixing.py                  Permute :32-54, LULinearPermute :352-492
  distributions/base.py            DiagGaussian :632-652
  core.py                          NormalizingFlow.log_prob :170-183, .sample :150-155
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import rqs, masks


def _row_sum(x):
    """utils/nn.py:131-134 sum_except_batch."""
    return torch.sum(x, dim=list(range(1, x.dim())))


class RQSCDF:
    """Per-feature spline with parameters shared over the batch
    (coupling.py:165-246).  No 1/sqrt(hidden) scaling is applied here."""

    def __init__(self, uw, uh, ud, tails, tail_bound,
                 min_bin_width=rqs.MIN_BIN_WIDTH, min_bin_height=rqs.MIN_BIN_HEIGHT,
                 min_derivative=rqs.MIN_DERIVATIVE):
        self.uw, self.uh, self.ud = uw, uh, ud
        self.tails, self.tail_bound = tails, tail_bound
        self.mins = (min_bin_width, min_bin_height, min_derivative)

    def _run(self, x, inverse):
        n = x.shape[0]
        uw = self.uw[None, ...].expand(n, *self.uw.shape)      # :208-209
        uh = self.uh[None, ...].expand(n, *self.uh.shape)
        ud = self.ud[None, ...].expand(n, *self.ud.shape)
        kw = dict(inverse=inverse, min_bin_width=self.mins[0],
                  min_bin_height=self.mins[1], min_derivative=self.mins[2])
        if self.tails is None:                                  # :218-226
            y, lad = rqs.rq_spline(x, uw, uh, ud, **kw)
        else:
            y, lad = rqs.rq_spline_tails(x, uw, uh, ud, tails=self.tails,
                                         tail_bound=self.tail_bound, **kw)
        return y, _row_sum(lad)                                 # :240

    def forward(self, x):
        return self._run(x, False)

    def inverse(self, x):
        return self._run(x, True)


class RQSCoupling:
    """nsf-convention RQS coupling: ``nsf_forward`` is the density direction.
    ``conditioner(identity_split, context) -> [B, d_t * P]``."""

    def __init__(self, identity_idx, transform_idx, conditioner, num_bins, tails,
                 tail_bound, hidden_features=None, uncond=None,
                 min_bin_width=rqs.MIN_BIN_WIDTH, min_bin_height=rqs.MIN_BIN_HEIGHT,
                 min_derivative=rqs.MIN_DERIVATIVE):
        self.idf, self.tf = identity_idx, transform_idx
        self.conditioner = conditioner
        self.k, self.tails, self.tail_bound = num_bins, tails, tail_bound
        self.hidden = hidden_features
        self.uncond = uncond
        self.mins = (min_bin_width, min_bin_height, min_derivative)

    def _spline(self, x, params, inverse):
        """coupling.py:147-159 + :309-343; 4-D inputs: Bx(C*P)xHxW -> BxCxHxWxP (:148-151)."""
        if x.dim() == 4:
            b, c, h, w = x.shape
            p = params.reshape(b, c, -1, h, w).permute(0, 1, 3, 4, 2)   # :151
        else:
            b, d = x.shape
            p = params.reshape(b, d, -1)                        # :155
        uw = p[..., :self.k]
        uh = p[..., self.k:2 * self.k]
        ud = p[..., 2 * self.k:]
        if self.hidden is not None:                             # :314-316 (in place on views)
            uw /= np.sqrt(self.hidden)
            uh /= np.sqrt(self.hidden)
        kw = dict(inverse=inverse, min_bin_width=self.mins[0],
                  min_bin_height=self.mins[1], min_derivative=self.mins[2])
        if self.tails is None:
            y, lad = rqs.rq_spline(x, uw, uh, ud, **kw)
        else:
            y, lad = rqs.rq_spline_tails(x, uw, uh, ud, tails=self.tails,
                                         tail_bound=self.tail_bound, **kw)
        return y, _row_sum(lad)                                 # :159

    def nsf_forward(self, x, context=None):
        """coupling.py:70-96: conditioner on the untouched identity half, spline
        on the transform half, THEN the unconditional spline on the identity
        half; both halves scattered into a fresh tensor."""
        xi, xt = x[:, self.idf], x[:, self.tf]
        params = self.conditioner(xi, context)
        yt, lad = self._spline(xt, params, False)
        if self.uncond is not None:
            xi, lad_i = self.uncond.forward(xi)
            lad = lad + lad_i
        y = torch.empty_like(x)
        y[:, self.idf] = xi
        y[:, self.tf] = yt
        return y, lad

    def nsf_inverse(self, x, context=None):
        """coupling.py:98-125: the unconditional inverse runs FIRST and the
        conditioner sees its output."""
        xi, xt = x[:, self.idf], x[:, self.tf]
        lad = 0.0
        if self.uncond is not None:
            xi, lad = self.uncond.inverse(xi)
        params = self.conditioner(xi, context)
        yt, lad_t = self._spline(xt, params, True)
        lad = lad + lad_t
        y = torch.empty_like(x)
        y[:, self.idf] = xi
        y[:, self.tf] = yt
        return y, lad

    # normflow convention used by the wrapper (wrapper.py:69-75): forward is the
    # nsf inverse and vice versa, log_det flattened to [B].
    def forward(self, z, context=None):
        y, ld = self.nsf_inverse(z, context)
        return y, ld.view(-1)

    def inverse(self, z, context=None):
        y, ld = self.nsf_forward(z, context)
        return y, ld.view(-1)


class AffineCoupling:
    """flows/affine/coupling.py:113-168 on an already split pair [z1, z2].
    ``param_fn(z1)`` returns interleaved (shift, scale) along dim 1."""

    def __init__(self, param_fn, scale=True, scale_map="exp"):
        self.param_fn, self.scale, self.scale_map = param_fn, scale, scale_map

    def _apply(self, z, inverse):
        z1, z2 = z
        param = self.param_fn(z1)
        if not self.scale:                                      # :139-141 / :165-167
            return [z1, z2 - param if inverse else z2 + param], 0
        shift = param[:, 0::2, ...]                             # :122-123
        sc = param[:, 1::2, ...]
        dims = list(range(1, shift.dim()))
        if self.scale_map == "exp":                             # :124-126 / :150-152
            if inverse:
                return [z1, (z2 - shift) * torch.exp(-sc)], -torch.sum(sc, dim=dims)
            return [z1, z2 * torch.exp(sc) + shift], torch.sum(sc, dim=dims)
        if self.scale_map not in ("sigmoid", "sigmoid_inv"):
            raise NotImplementedError("This scale map is not implemented.")
        sg = torch.sigmoid(sc + 2)
        lsum = torch.sum(torch.log(sg), dim=dims)
        divide = (self.scale_map == "sigmoid") != inverse       # :127-136 / :153-162
        if inverse:
            z2n = (z2 - shift) / sg if divide else (z2 - shift) * sg
        else:
            z2n = z2 / sg + shift if divide else z2 * sg + shift
        return [z1, z2n], (-lsum if divide else lsum)

    def forward(self, z):
        return self._apply(z, False)

    def inverse(self, z):
        return self._apply(z, True)


def checker_mask(shape, inv):
    """reshape.py:30-38: 0/1 colouring of one data point of ``shape`` (no batch dim); an
    element's colour is the parity of the sum of its indices, 'inv' flips it."""
    cb0, cb1 = 0, 1
    for n in reversed(shape):
        cb0, cb1 = ([cb0 if j % 2 == 0 else cb1 for j in range(n)],
                    [cb1 if j % 2 == 0 else cb0 for j in range(n)])
    return torch.tensor(cb1 if inv else cb0)


def checker_split(z, inv):
    """reshape.py:39-42: z1 = the elements coloured 1, z2 = the others, last dim halved."""
    cb = checker_mask(tuple(z.shape[1:]), inv)[None].repeat(len(z), *((z.dim() - 1) * [1]))
    size = z.size()
    z1 = z.reshape(-1)[torch.nonzero(cb.view(-1), as_tuple=False)].view(*size[:-1], -1)
    z2 = z.reshape(-1)[torch.nonzero((1 - cb).view(-1), as_tuple=False)].view(*size[:-1], -1)
    return z1, z2


def checker_merge(z1, z2, inv):
    """reshape.py:56-72: both halves repeated along the last dim and blended by the mask."""
    n = z1.dim()
    size = list(z1.size())
    size[-1] *= 2
    cb = checker_mask(tuple(size[1:]), inv)[None].repeat(size[0], *((n - 1) * [1]))
    a = z1[..., None].repeat(*(n * [1]), 2).view(*size[:-1], -1)
    b = z2[..., None].repeat(*(n * [1]), 2).view(*size[:-1], -1)
    return cb * a + (1 - cb) * b


class AffineCouplingBlock:
    """Split -> AffineCoupling -> Merge (coupling.py:225-258; reshape.py channel /
    channel_inv modes :25-29, :50-55; checkerboard modes :30-42, :56-72)."""

    def __init__(self, param_fn, scale=True, scale_map="exp", split_mode="channel"):
        if split_mode not in ("channel", "channel_inv", "checkerboard", "checkerboard_inv"):
            raise NotImplementedError("Mode " + split_mode + " is not implemented.")
        self.core = AffineCoupling(param_fn, scale, scale_map)
        self.checker = "checkerboard" in split_mode
        self.flip = split_mode.endswith("_inv")

    def _run(self, z, inverse):
        tot = torch.zeros(z.shape[0], dtype=z.dtype)
        if self.checker:
            pair = list(checker_split(z, self.flip))
        else:
            a, b = z.chunk(2, dim=1)
            pair = [b, a] if self.flip else [a, b]
        pair, ld = (self.core.inverse if inverse else self.core.forward)(pair)
        tot += ld
        if self.checker:
            return checker_merge(pair[0], pair[1], self.flip), tot
        out = torch.cat([pair[1], pair[0]] if self.flip else pair, 1)
        return out, tot

    def forward(self, z):
        return self._run(z, False)

    def inverse(self, z):
        return self._run(z, True)


class AutoregressiveRQS:
    """neural_spline/autoregressive.py:92-136 on top of affine/autoregressive.py:24-36: ``nsf_forward``
    (density) is one conditioner pass, ``nsf_inverse`` (sampling) D passes from zeros.  ``conditioner(x)``
    returns [B, D * P]; the reference's MADE has no ``hidden_features`` attribute, so the width / height
    logits are NOT scaled (:104-106)."""

    def __init__(self, conditioner, features, num_bins, tails, tail_bound,
                 min_bin_width=rqs.MIN_BIN_WIDTH, min_bin_height=rqs.MIN_BIN_HEIGHT,
                 min_derivative=rqs.MIN_DERIVATIVE):
        self.conditioner, self.d, self.k = conditioner, features, num_bins
        self.tails, self.tail_bound = tails, tail_bound
        self.mins = (min_bin_width, min_bin_height, min_derivative)

    def _elementwise(self, x, params, inverse):
        p = params.view(x.shape[0], self.d, -1)                                     # :95-99
        uw, uh, ud = p[..., :self.k], p[..., self.k:2 * self.k], p[..., 2 * self.k:]
        kw = dict(inverse=inverse, min_bin_width=self.mins[0], min_bin_height=self.mins[1],
                  min_derivative=self.mins[2])
        if self.tails is None:
            y, lad = rqs.rq_spline(x, uw, uh, ud, **kw)
        else:
            y, lad = rqs.rq_spline_tails(x, uw, uh, ud, tails=self.tails, tail_bound=self.tail_bound, **kw)
        return y, _row_sum(lad)

    def nsf_forward(self, x):
        return self._elementwise(x, self.conditioner(x), False)

    def nsf_inverse(self, x):
        out, lad = torch.zeros_like(x), None
        for _ in range(int(np.prod(x.shape[1:]))):                                  # affine/autoregressive.py:30-35
            out, lad = self._elementwise(x, self.conditioner(out), True)
        return out, lad

    def forward(self, z):                                                           # wrapper.py:251-253
        y, ld = self.nsf_inverse(z)
        return y, ld.view(-1)

    def inverse(self, z):
        y, ld = self.nsf_forward(z)
        return y, ld.view(-1)


class MaskedAffineAutoregressive:
    """affine/autoregressive.py:48-103 on top of :24-36: ``forward`` is one conditioner pass with
    scale = sigmoid(u + 2) + 1e-3, y = scale x + shift, log|det| = sum log scale (:75-81); ``inverse`` D passes from
    zeros with y = (x - shift) / scale, log|det| = -sum log scale (:83-89).  ``conditioner(x)`` returns [B, D * 2] laid
    out (u, shift) per feature (:96-103)."""

    def __init__(self, conditioner, features):
        self.conditioner, self.d = conditioner, features

    def _scale_shift(self, params):
        p = params.view(-1, self.d, 2)                                              # :96-103
        return torch.sigmoid(p[..., 0] + 2.) + 1e-3, p[..., 1]                      # :77, :85

    def forward(self, x):
        scale, shift = self._scale_shift(self.conditioner(x))
        return scale * x + shift, _row_sum(torch.log(scale))                        # :78-81

    def inverse(self, x):
        out, lad = torch.zeros_like(x), None
        for _ in range(int(np.prod(x.shape[1:]))):                                  # :30-35
            scale, shift = self._scale_shift(self.conditioner(out))
            out, lad = (x - shift) / scale, -_row_sum(torch.log(scale))             # :86-89
        return out, lad


class LULinearPermute:
    """mixing.py:352-492: fixed permutation + linear map y = x (L U)^T + bias, L unit lower,
    diag(U) = softplus(.) + eps.  ``forward`` (sampling direction) = inverse linear map by two
    triangular solves, then inverse permutation; ``inverse`` = permutation, then the map."""

    def __init__(self, perm, bias, lower_entries, upper_entries, unconstrained_upper_diag, eps=1e-3):
        d = bias.numel()
        self.perm, self.bias, self.d = perm, bias, d
        li, ui = np.tril_indices(d, k=-1), np.triu_indices(d, k=1)
        diag = F.softplus(unconstrained_upper_diag) + eps                         # :453-455
        lower = lower_entries.new_zeros(d, d)                                       # :388-399
        lower[li[0], li[1]] = lower_entries
        lower[range(d), range(d)] = 1.0
        upper = upper_entries.new_zeros(d, d)
        upper[ui[0], ui[1]] = upper_entries
        upper[range(d), range(d)] = diag
        self.lower, self.upper = lower, upper
        self.logabsdet = torch.sum(torch.log(diag))                                 # :457-464

    def forward(self, z):                                                           # :484-487
        out = z - self.bias                                                         # :420
        out = torch.linalg.solve_triangular(self.lower, out.t(), upper=False, unitriangular=True)
        out = torch.linalg.solve_triangular(self.upper, out, upper=True).t()        # :421-423
        out = out[:, torch.argsort(self.perm)]                                      # :212, :226-229
        return out, (-self.logabsdet * z.new_ones(z.shape[0])).view(-1)

    def inverse(self, z):                                                           # :489-492
        out = z[:, self.perm]
        out = F.linear(out, self.upper)                                             # :409-410
        out = F.linear(out, self.lower, self.bias)
        return out, (self.logabsdet * z.new_ones(z.shape[0])).view(-1)


class MaskedAffine:
    """coupling.py:171-222; ``b`` is the [1, D] float mask buffer, ``s_fn`` /
    ``t_fn`` map the masked input to full-width scale / shift (None -> zeros)."""

    def __init__(self, b, s_fn=None, t_fn=None):
        self.b = b
        self.s_fn = s_fn if s_fn is not None else torch.zeros_like
        self.t_fn = t_fn if t_fn is not None else torch.zeros_like

    def _st(self, z):
        zm = self.b * z
        nan = torch.tensor(np.nan, dtype=z.dtype)
        s = self.s_fn(zm)
        s = torch.where(torch.isfinite(s), s, nan)              # :205-206
        t = self.t_fn(zm)
        t = torch.where(torch.isfinite(t), t, nan)              # :207-208
        return zm, s, t

    def forward(self, z):
        zm, s, t = self._st(z)
        out = zm + (1 - self.b) * (z * torch.exp(s) + t)        # :209
        return out, torch.sum((1 - self.b) * s, dim=list(range(1, self.b.dim())))

    def inverse(self, z):
        zm, s, t = self._st(z)
        out = zm + (1 - self.b) * (z - t) * torch.exp(-s)       # :220
        return out, -torch.sum((1 - self.b) * s, dim=list(range(1, self.b.dim())))


class AffineConst:
    """AffineConstFlow, coupling.py:10-53: per-feature scale/shift, log_det a
    0-dim tensor times the product of broadcast (size-1) non-batch dims."""

    def __init__(self, s, t):
        self.s, self.t = s, t
        self.bdims = [i for i, n in enumerate(s.shape) if n == 1]

    def _mult(self, z):
        return int(np.prod([z.size(i) for i in self.bdims[1:]])) if len(self.bdims) > 1 else 1

    def forward(self, z):
        return z * torch.exp(self.s) + self.t, self._mult(z) * torch.sum(self.s)

    def inverse(self, z):
        return (z - self.t) * torch.exp(-self.s), -self._mult(z) * torch.sum(self.s)


class Permute:
    """mixing.py:10-54."""

    def __init__(self, num_channels, mode="shuffle", perm=None, inv_perm=None):
        self.c, self.mode, self.perm, self.inv_perm = num_channels, mode, perm, inv_perm
        if mode not in ("shuffle", "swap"):
            raise NotImplementedError("The mode " + mode + " is not implemented.")

    def forward(self, z):
        idx = self.perm if self.mode == "shuffle" else masks.swap_index(self.c, False)
        return z[:, idx, ...], 0

    def inverse(self, z):
        idx = self.inv_perm if self.mode == "shuffle" else masks.swap_index(self.c, True)
        return z[:, idx, ...], 0


class DiagGaussian:
    """distributions/base.py:609-652; ``loc`` / ``log_scale`` are [1, *shape]."""

    def __init__(self, loc, log_scale, temperature=None):
        self.loc, self.log_scale, self.temperature = loc, log_scale, temperature
        self.d = int(np.prod(loc.shape[1:]))
        self.dims = list(range(1, loc.dim()))

    def _ls(self):
        if self.temperature is None:
            return self.log_scale
        return self.log_scale + np.log(self.temperature)

    def from_noise(self, eps):
        """base.py:632-642 with the standard-normal draw ``eps`` supplied by the
        caller (RNG streams differ between devices; the draw is an input)."""
        ls = self._ls()
        z = self.loc + torch.exp(ls) * eps
        logp = -0.5 * self.d * np.log(2 * np.pi) - torch.sum(ls + 0.5 * torch.pow(eps, 2), self.dims)
        return z, logp

    def log_prob(self, z):
        """base.py:644-652."""
        ls = self._ls()
        return -0.5 * self.d * np.log(2 * np.pi) - torch.sum(
            ls + 0.5 * torch.pow((z - self.loc) / torch.exp(ls), 2), self.dims)


def _call(fn, z, context):
    return fn(z, context) if context is not None else fn(z)


class Stack:
    """core.py NormalizingFlow loops.  ``context`` (optional) is handed to every
    layer that takes one (the conditional RQS couplings of config C3)."""

    def __init__(self, q0, flows):
        self.q0, self.flows = q0, list(flows)

    def log_prob(self, x, context=None, trace=None):
        """core.py:176-183: walk the flows backwards through ``inverse``, add
        each log_det, finish with the base log-density."""
        log_q = torch.zeros(len(x), dtype=x.dtype)
        z = x
        for f in reversed(self.flows):
            takes_ctx = isinstance(f, RQSCoupling)
            z, ld = _call(f.inverse, z, context if takes_ctx else None)
            log_q += ld
            if trace is not None:
                trace.append((z, ld))
        log_q += self.q0.log_prob(z)
        return log_q

    def sample_from(self, eps, context=None, trace=None):
        """core.py:150-155 with the base noise supplied: z0, log_q from q0, then
        every flow forward, subtracting its log_det."""
        z, log_q = self.q0.from_noise(eps)
        for f in self.flows:
            takes_ctx = isinstance(f, RQSCoupling)
            z, ld = _call(f.forward, z, context if takes_ctx else None)
            log_q -= ld
            if trace is not None:
                trace.append((z, ld))
        return z, log_q


class Invertible1x1ConvLU:
    """flows/mixing.py:57-128 with use_lu=True: W = P (tril(L,-1)+I) (triu(U,1)+diag(sign_S e^{log_S}))."""

    def __init__(self, P, L, U, sign_S, log_S, eye):
        self.P, self.L, self.U, self.sign_S, self.log_S, self.eye = P, L, U, sign_S, log_S, eye
        self.c = P.shape[0]

    def _w(self, inverse):
        lo = torch.tril(self.L, diagonal=-1) + self.eye                       # :85
        up = torch.triu(self.U, diagonal=1) + torch.diag(self.sign_S * torch.exp(self.log_S))
        if inverse:                                                           # :87-95
            li = torch.inverse(lo.double()).type(self.log_S.dtype)
            ui = torch.inverse(up.double()).type(self.log_S.dtype)
            return ui @ li @ self.P.t()
        return self.P @ lo @ up                                               # :97

    def forward(self, z):                                                     # :100-116
        w = self._w(True).view(self.c, self.c, 1, 1)
        return torch.nn.functional.conv2d(z, w), -torch.sum(self.log_S) * z.size(2) * z.size(3)

    def inverse(self, z):                                                     # :118-128
        w = self._w(False).view(self.c, self.c, 1, 1)
        return torch.nn.functional.conv2d(z, w), torch.sum(self.log_S) * z.size(2) * z.size(3)


class Squeeze:
    """flows/reshape.py:93-116."""

    def forward(self, z):
        s = z.size()
        z = z.view(s[0], s[1] // 4, 2, 2, s[2], s[3]).permute(0, 1, 4, 2, 5, 3).contiguous()
        return z.view(s[0], s[1] // 4, 2 * s[2], 2 * s[3]), 0

    def inverse(self, z):
        s = z.size()
        z = z.view(*s[:2], s[2] // 2, 2, s[3] // 2, 2).permute(0, 1, 3, 5, 2, 4).contiguous()
        return z.view(s[0], 4 * s[1], s[2] // 2, s[3] // 2), 0


class Chain:
    """GlowBlock-style composite (flows/affine/glow.py:63-74): flows in order, log-dets summed."""

    def __init__(self, flows):
        self.flows = list(flows)

    def forward(self, z):
        tot = torch.zeros(z.shape[0], dtype=z.dtype)
        for f in self.flows:
            z, ld = f.forward(z)
            tot = tot + ld
        return z, tot

    def inverse(self, z):
        tot = torch.zeros(z.shape[0], dtype=z.dtype)
        for f in reversed(self.flows):
            z, ld = f.inverse(z)
            tot = tot + ld
        return z, tot


class Multiscale:
    """core.py MultiscaleFlow: log_prob :348-367, sample :320-340 (channel Merge between
    levels: reshape.py:50-55, :25-29)."""

    def __init__(self, q0, flows):
        self.q0, self.flows = list(q0), [list(f) for f in flows]

    def log_prob(self, x):
        log_q, z = 0, x
        for i in range(len(self.q0) - 1, -1, -1):
            for f in reversed(self.flows[i]):
                z, ld = f.inverse(z)
                log_q = log_q + ld
            if i > 0:
                z, z_ = z.chunk(2, dim=1)           # Merge.inverse = Split.forward, mode 'channel'
            else:
                z_ = z
            log_q = log_q + self.q0[i].log_prob(z_)
        return log_q

    def sample_from(self, noise):
        z = log_q = None
        for i in range(len(self.q0)):
            z_, lq_ = self.q0[i].from_noise(noise[i])
            if i == 0:
                z, log_q = z_, lq_
            else:
                log_q = log_q + lq_
                z = torch.cat([z, z_], 1)           # Merge.forward
            for f in self.flows[i]:
                z, ld = f.forward(z)
                log_q = log_q - ld
        return z, log_q
