"""Training path: gradients of the HIP spline VJP kernel and of the
layers built on it, against torch autograd over the oracle in fp64 on the same inputs.

Tolerance: gradients are fp32 results of fp32 forward values; the comparison is
|got - want| <= rtol*|want| + atol*scale with scale = rms of the oracle gradient, which
is the usual gradcheck convention for single precision (rtol 2e-3: the spline's
derivative terms amplify the 1e-7 relative error of fp32 knots by 1/bin-width).
"""
import numpy as np
import pytest
import torch

import vcnf_amd as nf
from vcnf_amd import _lib, autograd as vag
from oracle import rqs as orqs
from helpers import fixture, T, state_for, oracle_c3_stack, oracle_crqs_stack, within_reference_noise

pytestmark = pytest.mark.gpu


def dev(t):
    return t.to("cuda")


def close(got, want, what, want32=None, rtol=2e-3, atol=2e-4):
    """``want``: oracle fp64 gradient.  ``want32``: the oracle's own fp32 autograd gradient
    (what the reference computes when it trains); when given, the build's error against fp64
    must stay within the reference's own fp32 error distribution (helpers.within_reference_noise)
    and the elementwise bound is widened by that noise."""
    got = got.detach().cpu().double()
    want = want.detach().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.isfinite(got).all(), what + ": non-finite gradient"
    scale = float(want.pow(2).mean().sqrt()) + 1e-12
    err = (got - want).abs()
    bound = rtol * want.abs() + atol * scale
    if want32 is not None:
        w32 = want32.detach().double()
        within_reference_noise(got, w32, want, slack=3.0, what=what)
        bound = bound + 8.0 * (w32 - want).abs()
    bad = err > bound
    frac = float(bad.double().mean())
    # bin-boundary elements: a knot that rounds to the other side of x in fp32 puts the
    # element in the neighbouring bin, whose parameter gradient is a different (equally valid)
    # one-sided derivative; allow a 2e-3 fraction of such elements
    assert frac <= 2e-3, "%s: %d / %d outside tolerance, max err %.3e (scale %.3e)" % (
        what, int(bad.sum()), bad.numel(), float(err.max()), scale)


@pytest.mark.parametrize("tails", ["linear", None])
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("k", [8, 5, 16])
def test_spline_vjp_vs_oracle_autograd(hip, tails, inverse, k):
    g = torch.Generator().manual_seed(100 + k)
    n, bound = 20000, 3.0
    nd = k - 1 if tails == "linear" else k + 1
    if tails == "linear":
        x = (torch.rand(n, generator=g) * 2 - 1) * bound * 1.2          # includes the tails
    else:
        x = torch.rand(n, generator=g) * 0.998 + 0.001
    uw, uh, ud = (torch.randn(n, m, generator=g) * 1.5 for m in (k, k, nd))
    gy, gl = torch.randn(n, generator=g), torch.randn(n, generator=g)

    def oracle_grads(dtype):
        leaves = [t.to(dtype).requires_grad_() for t in (x, uw, uh, ud)]
        if tails == "linear":
            y, lad = orqs.rq_spline_tails(*leaves, inverse=inverse, tails="linear", tail_bound=bound)
        else:
            y, lad = orqs.rq_spline(*leaves, inverse=inverse)
        return y, torch.autograd.grad([y, lad], leaves, [gy.to(dtype), gl.to(dtype)])
    y, want = oracle_grads(torch.float64)
    _, want32 = oracle_grads(torch.float32)

    cfg = _lib.make_cfg(k, tails, tail_bound=bound if tails else 1.0)
    dl = [dev(t).requires_grad_() for t in (x, uw, uh, ud)]
    yy, ll = vag.rqs_spline(*dl, cfg, inverse=inverse)
    got = torch.autograd.grad([yy, ll], dl, [dev(gy), dev(gl)])
    assert torch.allclose(yy.cpu().double(), y.detach(), rtol=1e-4, atol=1e-4)
    for a, b, b32, nm in zip(got, want, want32, ("g_x", "g_uw", "g_uh", "g_ud")):
        close(a, b, "%s tails=%s inverse=%s K=%d" % (nm, tails, inverse, k), want32=b32)


def test_spline_vjp_tail_elements_are_identity(hip):
    k, bound = 8, 2.0
    x = torch.tensor([-5.0, -2.5, 2.0001, 7.0], device="cuda", requires_grad=True)
    uw, uh, ud = (torch.randn(4, m, device="cuda", requires_grad=True) for m in (k, k, k - 1))
    y, lad = vag.rqs_spline(x, uw, uh, ud, _lib.make_cfg(k, "linear", tail_bound=bound))
    gy = torch.tensor([1.0, -2.0, 3.0, 0.5], device="cuda")
    gx, gw, gh, gd = torch.autograd.grad([y, lad], [x, uw, uh, ud], [gy, torch.ones(4, device="cuda")])
    assert torch.equal(gx, gy)
    assert not gw.any() and not gh.any() and not gd.any()


def _c3_small(layers, d=64, c=16, hidden=128, blocks=2, k=8):
    flows = [nf.flows.CoupledRationalQuadraticSpline(d, blocks, hidden, k, reverse_mask=bool(i % 2),
                                                     num_context_channels=c) for i in range(layers)]
    return nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows)


def _leaf_state(sd):
    out = {}
    for k_, v in sd.items():
        out[k_] = v.double().requires_grad_() if v.is_floating_point() else v
    return out


@pytest.mark.parametrize("direction", ["log_prob", "sample"])
def test_c3_stack_parameter_gradients(hip, direction):
    """d(mean log q)/d(theta) of a 3-layer C3 stack (context 16) against autograd over the
    oracle stack in fp64: every parameter tensor, plus the input gradient."""
    fx = fixture("g5_c3_stack")
    sd, _ = state_for(fx, "c3", 501, final_gain=1.0)
    layers, b = 3, 512
    sd = {k_: v for k_, v in sd.items() if not k_.startswith("flows.") or int(k_.split(".")[1]) < layers}
    model = _c3_small(layers)
    model.load_state_dict(sd)
    model = model.to("cuda")
    x, ctx, eps = (T(fx[n])[:b] for n in ("x", "ctx", "eps"))

    sd64 = _leaf_state(sd)
    ora = oracle_c3_stack(sd64, layers=layers)
    if direction == "log_prob":
        xin = x.double().requires_grad_()
        want_val = ora.log_prob(xin, ctx.double()).mean()
        xg = dev(x).requires_grad_()
        got_val = model.log_prob(xg, dev(ctx)).mean()
    else:
        xin = eps.double().requires_grad_()
        z, lq = ora.sample_from(xin, ctx.double())
        want_val = lq.mean() + (z * z).mean()
        xg = dev(eps).requires_grad_()
        z2, lq2 = model.sample_from(xg, dev(ctx))
        got_val = lq2.mean() + (z2 * z2).mean()
    names = [n for n, p in model.named_parameters()]
    want = torch.autograd.grad(want_val, [xin] + [sd64[n] for n in names], allow_unused=True)
    got_val.backward()
    assert abs(float(got_val) - float(want_val)) < 1e-4 * max(1.0, abs(float(want_val)))
    close(xg.grad, want[0], "input gradient (%s)" % direction, atol=1e-3)
    params = dict(model.named_parameters())
    for n, w in zip(names, want[1:]):
        gp = params[n].grad
        if w is None:
            assert gp is None or not gp.any(), n
            continue
        assert gp is not None, n + ": no gradient reached this parameter"
        close(gp, w, "%s (%s)" % (n, direction), atol=1e-3)
