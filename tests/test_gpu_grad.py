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
from helpers import fixture, T, state_for, oracle_c3_stack, oracle_crqs_stack, within_reference_noise, assert_close

pytestmark = pytest.mark.gpu


def dev(t):
    return t.to("cuda")


def close(got, want, what, want32=None, rtol=2e-3, atol=2e-4, max_frac=2e-3):
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
    # one-sided derivative; allow a 2e-3 fraction of such elements (``max_frac``: 12-layer stacks pass
    # 12 x 32 splines per sample, a handful of weights of a 2048-entry matrix can collect such samples)
    assert frac <= max_frac, "%s: %d / %d outside tolerance, max err %.3e (scale %.3e)" % (
        what, int(bad.sum()), bad.numel(), float(err.max()), scale)


@pytest.mark.parametrize("tails", ["linear", None, "circular"])
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("k", [8, 5, 16])
def test_spline_vjp_vs_oracle_autograd(hip, tails, inverse, k):
    g = torch.Generator().manual_seed(100 + k)
    n, bound = 20000, 3.0
    nd = k - 1 if tails == "linear" else k if tails == "circular" else k + 1
    if tails is not None:
        x = (torch.rand(n, generator=g) * 2 - 1) * bound * 1.2          # includes the tails
    else:
        x = torch.rand(n, generator=g) * 0.998 + 0.001
    uw, uh, ud = (torch.randn(n, m, generator=g) * 1.5 for m in (k, k, nd))
    gy, gl = torch.randn(n, generator=g), torch.randn(n, generator=g)

    def oracle_grads(dtype):
        leaves = [t.clone().to(dtype).requires_grad_() for t in (x, uw, uh, ud)]
        if tails is not None:
            y, lad = orqs.rq_spline_tails(*leaves, inverse=inverse, tails=tails, tail_bound=bound)
        else:
            y, lad = orqs.rq_spline(*leaves, inverse=inverse)
        return y, torch.autograd.grad([y, lad], leaves, [gy.to(dtype), gl.to(dtype)])
    y, want = oracle_grads(torch.float64)
    y32, want32 = oracle_grads(torch.float32)

    cfg = _lib.make_cfg(k, tails, tail_bound=bound if tails else 1.0)
    dl = [dev(t).requires_grad_() for t in (x, uw, uh, ud)]
    yy, ll = vag.rqs_spline(*dl, cfg, inverse=inverse)
    got = torch.autograd.grad([yy, ll], dl, [dev(gy), dev(gl)])
    noise = float((y32.detach().double() - y.detach()).abs().max())      # the oracle's own fp32 error
    assert float((yy.detach().cpu().double() - y.detach()).abs().max()) <= 1e-4 + 8 * noise
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
    assert abs(float(got_val.detach()) - float(want_val.detach())) < 1e-4 * max(1.0, abs(float(want_val.detach())))
    close(xg.grad, want[0], "input gradient (%s)" % direction, atol=1e-3)
    params = dict(model.named_parameters())
    for n, w in zip(names, want[1:]):
        gp = params[n].grad
        if w is None:
            assert gp is None or not gp.any(), n
            continue
        assert gp is not None, n + ": no gradient reached this parameter"
        close(gp, w, "%s (%s)" % (n, direction), atol=1e-3)


# ---------------------------------------------------------------- affine family
def _grad_compare(call, ora_fn, sd, inputs, what, atol=1e-3, loss_of=None):
    """Backward through ``model`` on the GPU and through the oracle (fp64 and fp32) on the CPU
    for the same scalar; compare every parameter gradient and the input gradient."""
    loss_of = loss_of or (lambda out: out.mean())
    model = call.__self__                      # ``call`` is a bound method of the module under test
    model.zero_grad()
    names = [n for n, _ in model.named_parameters()]

    def oracle(dtype):
        leaves = {k_: (v.detach().clone().to(dtype).requires_grad_() if v.is_floating_point() else v)
                  for k_, v in sd.items()}
        xin = [t.detach().clone().to(dtype).requires_grad_() for t in inputs]
        val = loss_of(ora_fn(leaves, *xin))
        gs = torch.autograd.grad(val, xin + [leaves[n] for n in names], allow_unused=True)
        return val, gs
    v64, g64 = oracle(torch.float64)
    _, g32 = oracle(torch.float32)
    xg = [dev(t.detach().clone()).requires_grad_() for t in inputs]
    val = loss_of(call(*xg))
    val.backward()
    val, v64 = val.detach(), v64.detach()
    assert abs(float(val) - float(v64)) < 1e-4 * max(1.0, abs(float(v64))), (float(val), float(v64))
    got = [t.grad for t in xg] + [dict(model.named_parameters())[n].grad for n in names]
    labels = ["input%d" % i for i in range(len(xg))] + names
    for lab, a, b, b32 in zip(labels, got, g64, g32):
        if b is None:
            assert a is None or not a.any(), lab
            continue
        assert a is not None, lab + ": no gradient reached this tensor"
        close(a, b.reshape(a.shape), "%s %s" % (what, lab), want32=b32.reshape(a.shape), atol=atol)


@pytest.mark.parametrize("name,tag,layers,d,widths,seed", [
    ("g10_c1_two_moons", "c1", 4, 2, [1, 32, 32, 2], 1001),
    ("g12_c2_tabular", "c2", 8, 32, [16, 64, 64, 32], 1201)])
@pytest.mark.parametrize("direction", ["log_prob", "sample"])
def test_affine_stack_gradients(hip, name, tag, layers, d, widths, seed, direction):
    """Configs C1 / C2 (AffineCouplingBlock + Permute stacks): gradient of mean log q."""
    from helpers import oracle_affine_stack
    fx = fixture(name)
    sd, _ = state_for(fx, tag, seed, weight_gain=0.4 if tag == "c2" else 1.0)
    flows = []
    for _ in range(layers):
        flows += [nf.flows.AffineCouplingBlock(nf.nets.MLP(widths)), nf.flows.Permute(d, mode="swap")]
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows)
    model.load_state_dict(sd)
    model = model.to("cuda")
    if direction == "log_prob":
        _grad_compare(model.log_prob, lambda s, x: oracle_affine_stack(s, layers, d).log_prob(x), sd,
                      [T(fx["x"])[:512]], tag + " log_prob")
    else:
        pick = lambda out: out[1].mean() + (out[0] ** 2).mean()
        _grad_compare(model.sample_from, lambda s, e: oracle_affine_stack(s, layers, d).sample_from(e), sd,
                      [T(fx["eps"])[:512]], tag + " sample", loss_of=pick)


@pytest.mark.parametrize("sm", ["exp", "sigmoid", "sigmoid_inv", "noscale"])
@pytest.mark.parametrize("dirn", ["forward", "inverse"])
def test_affine_coupling_block_scale_map_gradients(hip, sm, dirn):
    from oracle import layers as OL, nets as ON
    fx = fixture("g6_affine")
    d, mode = 33, "channel_inv"
    tag = "d%d/%s/%s" % (d, sm, mode)
    d1 = (d + 1) // 2
    cin, cout = d - d1, d1
    scale = sm != "noscale"
    blk = nf.flows.AffineCouplingBlock(nf.nets.MLP([cin, 24, 24, (2 if scale else 1) * cout]), scale=scale,
                                       scale_map=sm if scale else "exp", split_mode=mode)
    sd, _ = state_for(fx, tag, 601 + d)
    blk.load_state_dict(sd)
    blk = blk.to("cuda")

    def ora(s, x):
        o = OL.AffineCouplingBlock(lambda z: ON.mlp(s, "flows.1.param_map.", z, 0.0), scale=scale,
                                   scale_map=sm if scale else "exp", split_mode=mode)
        return getattr(o, dirn)(x)
    pick = lambda out: (out[0] ** 2).mean() + (out[1].mean() if torch.is_tensor(out[1]) else 0.0)
    _grad_compare(getattr(blk, dirn), ora, sd, [T(fx["d%d/x" % d])], "%s %s" % (tag, dirn), loss_of=pick)


@pytest.mark.parametrize("variant", ["st", "t_only", "s_only"])
@pytest.mark.parametrize("dirn", ["forward", "inverse"])
def test_masked_affine_gradients(hip, variant, dirn):
    from oracle import layers as OL, nets as ON
    fx = fixture("g7_masked_affine")
    d = 30
    tag = "d%d/%s" % (d, variant)
    s = nf.nets.MLP([d, 16, d]) if variant != "t_only" else None
    t = nf.nets.MLP([d, 16, d]) if variant != "s_only" else None
    b = T(fx["d%d/b" % d])
    m = nf.flows.MaskedAffineFlow(b, t, s)
    sd, _ = state_for(fx, tag, 701 + d)
    sd["b"] = b.view(1, -1)
    m.load_state_dict(sd)
    m = m.to("cuda")

    def ora(st, x):
        o = OL.MaskedAffine(st["b"].to(x.dtype),
                            s_fn=(lambda z: ON.mlp(st, "s.", z, 0.0)) if s is not None else None,
                            t_fn=(lambda z: ON.mlp(st, "t.", z, 0.0)) if t is not None else None)
        return getattr(o, dirn)(x)
    pick = lambda out: (out[0] ** 2).mean() + out[1].mean()
    _grad_compare(getattr(m, dirn), ora, sd, [T(fx["d%d/x" % d])], "%s %s" % (tag, dirn), loss_of=pick)


def test_glow_multiscale_gradients(hip):
    """Config C4's family: d(mean log q)/d(theta) through GlowBlocks (4-D affine coupling with
    sigmoid scale map, ActNorm, 1x1 convolution, squeeze, multiscale split)."""
    from helpers import oracle_glow_multiscale, glow_state
    from test_gpu_parity import _glow_model
    fx = fixture("g11_glow_multiscale")
    sd = glow_state(fx, 1101)
    model = _glow_model()
    model.load_state_dict(sd)
    model = model.to("cuda")
    _grad_compare(model.log_prob, lambda s, x: oracle_glow_multiscale(s).log_prob(x), sd, [T(fx["x"])],
                  "glow log_prob", atol=2e-3)
    # MultiscaleFlow.forward_kld (core.py:296-308) = -mean(log_prob), differentiable
    xg = dev(T(fx["x"]))
    loss = model.forward_kld(xg)
    assert_close(loss, -model.log_prob(xg).mean().detach().cpu(), rtol=1e-6, atol=1e-5, what="multiscale forward_kld")
    loss.backward()
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())


# ---------------------------------------------------------------- objectives
def test_forward_kld_training_reduces_loss(hip):
    """A few Adam steps of maximum likelihood on a C3-shaped stack (2 layers, context 16): the
    loss is finite, every parameter receives a gradient, and the loss goes down."""
    torch.manual_seed(5)
    model = _c3_small(2).to("cuda")
    x = torch.randn(2048, 64, device="cuda") * 0.7 + 0.3
    ctx = torch.randn(2048, 16, device="cuda")
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = model.forward_kld(x, context=ctx)
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
        opt.step()
        losses.append(float(loss))
    assert np.isfinite(losses).all() and losses[-1] < losses[0] - 0.5, losses
    with torch.no_grad():                       # evaluation path (fused kernel) agrees with the training path
        lp_eval = model.log_prob(x, ctx)
    lp_train = model.log_prob(x, ctx)
    assert lp_train.requires_grad
    assert torch.allclose(lp_eval, lp_train.detach(), rtol=1e-4, atol=2e-3)


def test_reverse_kld_and_alpha_div(hip):
    torch.manual_seed(6)
    flows = []
    for _ in range(3):
        flows += [nf.flows.AffineCouplingBlock(nf.nets.MLP([1, 16, 16, 2], init_zeros=True)),
                  nf.flows.Permute(2, mode="swap")]
    target = nf.distributions.DiagGaussian(2, trainable=False).to("cuda")
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(2), flows, p=target).to("cuda")
    for kw in (dict(), dict(score_fn=False)):
        model.zero_grad()
        loss = model.reverse_kld(512, **kw)
        loss.backward()
        assert torch.isfinite(loss) and abs(float(loss)) < 1e-3      # identity flow, target == base
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    for kw in (dict(alpha=0.5), dict(alpha=0.5, dreg=True)):
        model.zero_grad()
        loss = model.reverse_alpha_div(512, **kw)
        loss.backward()
        assert torch.isfinite(loss)
    out = model.forward_kld(torch.randn(64, 2, device="cuda"), extended=True)
    assert len(out) == 4 and len(out[1]) == len(model.flows) and out[3].shape == (64,)


def test_graphed_train_step_matches_eager_steps(hip):
    """nf.GraphedTrainStep: zero_grad + forward_kld + backward + optimiser step captured into ONE HIP graph.  Replayed
    on fresh batches it follows the eager loop on a copy of the model: same losses, same parameters afterwards (plain
    SGD: the update is linear in the gradient, so rounding-level differences stay at rounding level; with Adam the
    update lr * m / sqrt(v) turns a rounding-level gradient into a full +-lr step and two EAGER runs already differ by
    several lr), and the evaluation path sees the updated weights.  An Adam step is captured as well and must train."""
    import copy
    torch.manual_seed(9)
    model = _c3_small(3).to("cuda")
    twin = copy.deepcopy(model)
    B = 1024
    data = [(torch.randn(B, 64, device="cuda") * 0.7 + 0.3, torch.randn(B, 16, device="cuda")) for _ in range(6)]
    opt = torch.optim.SGD(model.parameters(), lr=2e-3)
    opt2 = torch.optim.SGD(twin.parameters(), lr=2e-3)
    step = nf.GraphedTrainStep(model, opt, batch=B, context_features=16, warmup=2)
    # the capture's warm-up performs 2 real steps on the first batch: the eager twin does the same
    losses, want = [], []
    for i, (x, c) in enumerate(data):
        for _ in range(3 if i == 0 else 1):
            opt2.zero_grad(set_to_none=True)
            l2 = twin.forward_kld(x, context=c)
            l2.backward()
            opt2.step()
        want.append(float(l2.detach()))
        losses.append(float(step(x, c).detach()))
    assert np.isfinite(losses).all()
    assert np.allclose(losses, want, rtol=1e-5, atol=1e-4), [a - b for a, b in zip(losses, want)]
    for (n, p), (_, q) in zip(model.named_parameters(), twin.named_parameters()):
        d = (p - q).abs()
        assert float(d.max()) <= 1e-5 + 1e-4 * float(q.abs().max()), (n, float(d.max()))
    with torch.no_grad():
        x, c = data[-1]
        assert torch.allclose(model.log_prob(x, c), twin.log_prob(x, c), rtol=1e-4, atol=2e-3)
    # Adam: state on the device (capturable=True) - a host-side step counter is refused
    with pytest.raises(ValueError):
        nf.GraphedTrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3), batch=B, context_features=16)
    adam = nf.GraphedTrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True), batch=B, context_features=16)
    x, c = data[0]
    la = [float(adam(x, c).detach()) for _ in range(8)]
    assert np.isfinite(la).all() and la[-1] < la[0] - 0.3, la
    nf.check_discriminant()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["fp64", "fp32"])
@pytest.mark.parametrize("d,h,nets,leaky", [(2, 16, "st", 0.0), (15, 30, "st", 0.0), (3, 8, "s", 0.1), (6, 40, "t", 0.0)])
def test_masked_affine_stack_vjp_matches_per_layer_autograd(hip, d, h, nets, leaky, dtype):
    """Training through the reference drivers' own models (K x [MaskedAffineFlow with MLP conditioners, ActNorm],
    /root/reference/run.py:58-68): the run is ONE autograd node (fused_masked.MaskedStackFn) whose backward is one
    launch of vcnf_masked_affine_stack_bwd_* - it keeps only the run's output and rebuilds every layer's input from
    it.  Against autograd over the per-layer path (torch GEMMs + the closed-form VJPs of vcnf_amd.autograd, themselves
    checked against the oracle / the reference's own autograd, G23): gradients of every parameter and of the input, for
    forward_kld (density direction) and for a loss on samples and their log-density (sampling direction)."""
    b = torch.tensor([1.0 if i % 2 == 0 else 0.0 for i in range(d)])
    torch.manual_seed(31 * d + h)
    flows = []
    for i in range(5):
        s = nf.nets.MLP([d, h, d], leaky=leaky, init_zeros=True) if "s" in nets else None
        t = nf.nets.MLP([d, h, d], leaky=leaky, init_zeros=True) if "t" in nets else None
        flows += [nf.flows.MaskedAffineFlow(b if i % 2 == 0 else 1 - b, t, s), nf.flows.ActNorm(d)]
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".net.2." in n:
                p.normal_(0.0, 0.5 / h ** 0.5)
    model = model.to(dtype).cuda()
    B = 300 + 7
    x0 = torch.randn(B, d, device="cuda", dtype=dtype)
    with torch.no_grad():
        model.log_prob(x0)                                     # ActNorm initialisation
        for f in model.flows:
            if isinstance(f, nf.flows.ActNorm):
                f.s.add_(0.05 * torch.randn_like(f.s))
    w = torch.randn(B, d, device="cuda", dtype=dtype)

    def grads(stacks, which):
        model.fuse_masked_stacks = stacks
        model.zero_grad(set_to_none=True)
        xin = x0.clone().requires_grad_(True)
        if which == "density":
            loss = model.forward_kld(xin) + 0.0
        else:
            z, lq = model.sample_from(xin)
            loss = (z * w).sum() / B + lq.mean()
        loss.backward()
        return [float(loss.detach())], [xin.grad] + [p.grad.clone() for p in model.parameters()]
    tol = 1e-9 if dtype == torch.float64 else 3e-4
    for which in ("density", "sampling"):
        l1, g1 = grads(True, which)
        l0, g0 = grads(False, which)
        assert np.isfinite(l0[0]) and abs(l1[0] - l0[0]) <= tol * (1 + abs(l0[0]))
        for i, (a, b_) in enumerate(zip(g1, g0)):
            assert a.shape == b_.shape
            err, scale = float((a - b_).abs().max()), float(b_.abs().max())
            assert err <= tol * (scale + 1e-3), (which, i, err, scale)
    model.fuse_masked_stacks = True


def test_nsf_stack_with_lu_linear_permute_trains(hip):
    """The neural-spline-flow layout of arXiv 1906.04032 (spline coupling + LULinearPermute per
    layer): gradients against the oracle stack in fp64, then a few optimiser steps."""
    from oracle import layers as OL
    from helpers import oracle_rqs_coupling
    torch.manual_seed(11)
    d, layers, hidden, k = 8, 2, 32, 6
    flows = []
    for i in range(layers):
        flows += [nf.flows.CoupledRationalQuadraticSpline(d, 1, hidden, k, reverse_mask=bool(i % 2)),
                  nf.flows.LULinearPermute(d, identity_init=False)]
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows)
    with torch.no_grad():                            # non-trivial conditioner outputs
        for n, p in model.named_parameters():
            if "final_layer" in n:
                p.normal_(0, 0.3)
    sd = {n: v.detach().clone() for n, v in model.state_dict().items()}
    model = model.to("cuda")
    x = torch.randn(256, d)

    def ora(s, xx):
        fl = []
        for i in range(layers):
            fl.append(oracle_rqs_coupling(s, "flows.%d.prqct." % (2 * i), k, 3.0, hidden))
            pre = "flows.%d." % (2 * i + 1)
            fl.append(OL.LULinearPermute(s[pre + "permutation._permutation"], s[pre + "linear.bias"],
                                         s[pre + "linear.lower_entries"], s[pre + "linear.upper_entries"],
                                         s[pre + "linear.unconstrained_upper_diag"]))
        return OL.Stack(OL.DiagGaussian(s["q0.loc"], s["q0.log_scale"]), fl).log_prob(xx)
    _grad_compare(model.log_prob, ora, sd, [x], "nsf log_prob")
    opt = torch.optim.Adam(model.parameters(), lr=3e-3)
    xs = dev(x) * 0.5 + 1.0
    first = last = None
    for _ in range(15):
        opt.zero_grad()
        loss = model.forward_kld(xs)
        loss.backward()
        opt.step()
        first = float(loss.detach()) if first is None else first
        last = float(loss.detach())
    assert np.isfinite(last) and last < first - 0.3, (first, last)


def test_image_rqs_coupling_gradients(hip):
    from helpers import oracle_image_rqs_coupling
    from test_gpu_parity import _image_coupling
    fx = fixture("g17_image_rqs")
    sd, _ = state_for(fx, "ctx", 1701, final_gain=2.0)
    m = _image_coupling(2)
    m.load_state_dict(sd)
    m = m.to("cuda")
    pick = lambda out: (out[0] ** 2).mean() + out[1].mean()
    for dirn in ("forward", "inverse"):
        ofn = "nsf_forward" if dirn == "forward" else "nsf_inverse"
        _grad_compare(getattr(m, dirn), lambda s, x, c: getattr(oracle_image_rqs_coupling(s), ofn)(x, c), sd,
                      [T(fx["x"]), T(fx["ctx"])], "image rqs " + dirn, loss_of=pick)

@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["fp32", "fp64"])
@pytest.mark.parametrize("c,first,b", [(64, 32, 4099), (7, 3, 1000), (5, 0, 17), (4, 4, 33), (1, 1, 3)])
def test_split_and_merge_columns_are_the_channel_partition_and_its_vjp(hip, c, first, b, dtype):
    """coupling.py:86-88 / :122-124 as one pass each way: exact copies, and each direction is the other's VJP."""
    g = torch.Generator().manual_seed(c * 131 + first)
    perm = torch.randperm(c, generator=g)
    gather = perm.to(torch.int32).cuda()
    scatter = torch.argsort(perm).to(torch.int32).cuda()
    z = torch.randn(b, c, generator=g, dtype=dtype).cuda().requires_grad_(True)
    pa, pb = vag.SplitColumnsFn.apply(z, gather, scatter, first)
    ref = z.detach()[:, perm.cuda()]
    assert torch.equal(pa.detach(), ref[:, :first]) and torch.equal(pb.detach(), ref[:, first:])
    assert pa.is_contiguous() and pb.is_contiguous()
    out = vag.MergeColumnsFn.apply(pa * 2.0, pb * 3.0, gather, scatter)
    want = torch.cat([ref[:, :first] * 2.0, ref[:, first:] * 3.0], 1)[:, torch.argsort(perm).cuda()]
    assert torch.equal(out.detach(), want)
    w = torch.randn(b, c, generator=g, dtype=dtype).cuda()
    (out * w).sum().backward()
    scale = torch.empty(c, dtype=dtype)
    scale[perm[:first]] = 2.0
    scale[perm[first:]] = 3.0
    assert torch.equal(z.grad, w * scale.cuda())
    # one of the two parts unused downstream: its gradient arrives as zeros
    z2 = z.detach().clone().requires_grad_(True)
    qa, qb = vag.SplitColumnsFn.apply(z2, gather, scatter, first)
    (qb.sum() if first < c else qa.sum()).backward()
    expect = torch.zeros(c, dtype=dtype)
    expect[perm[first:] if first < c else perm[:first]] = 1.0
    assert torch.equal(z2.grad, expect.cuda().expand(b, c))


@pytest.mark.parametrize("n_in,n_out,b", [(128, 128, 16384 + 37), (48, 128, 8192), (16, 128, 9000), (128, 736, 10000), (64, 5, 300)])
def test_linear_wgrad_kernel(hip, n_in, n_out, b):
    """csrc/linear_wgrad.hip: dW = dy^T x and db = sum dy with the batch reduction split over the chip, against fp64
    beside torch's own fp32 GEMM / column sum (the reference's autograd path, nets/resnet.py:92-106); ragged batches,
    partial row tiles (OUT not a multiple of 128 / 32)."""
    from vcnf_amd import _lib
    g = torch.Generator().manual_seed(n_in + n_out)
    x = torch.randn(b, n_in, generator=g).cuda()
    dy = torch.randn(b, n_out, generator=g).cuda()
    dw, db = _lib.linear_wgrad(x, dy)
    w64, b64 = dy.double().t() @ x.double(), dy.double().sum(0)
    w32, b32 = dy.t() @ x, dy.sum(0)
    for got, r64, r32, what in ((dw, w64, w32, "dW"), (db, b64, b32, "db")):
        e_got, e_ref = float((got.double() - r64).abs().max()), float((r32.double() - r64).abs().max())
        assert got.shape == r64.shape and e_got <= 2.0 * e_ref + 1e-6 * float(r64.abs().max()), (what, e_got, e_ref)
    dw2, none = _lib.linear_wgrad(x, dy, want_bias=False)
    assert none is None and torch.equal(dw2, dw)                     # deterministic
    # the split-half form of the same kernel (the training path's default): mean error against fp64 not above the
    # fp32 GEMM's, deterministic, nothing clamped on these inputs; a value beyond the fp16 range is counted
    nf.check_saturation()
    dwh, dbh = _lib.linear_wgrad(x, dy, f16x3=True)
    for got, r64, r32, what in ((dwh, w64, w32, "dW split-half"), (dbh, b64, b32, "db split-half")):
        e_got, e_ref = (got.double() - r64).abs(), (r32.double() - r64).abs()
        assert got.shape == r64.shape and float(e_got.mean()) <= 1.1 * float(e_ref.mean()) + 1e-7 * float(r64.abs().max()), \
            (what, float(e_got.mean()), float(e_ref.mean()))
        assert float(e_got.max()) <= 2.0 * float(e_ref.max()) + 1e-6 * float(r64.abs().max()), (what, float(e_got.max()))
    assert torch.equal(_lib.linear_wgrad(x, dy, f16x3=True)[0], dwh)
    # ReLU on the layer input applied on load: the same values as materialising relu(x) first
    assert torch.equal(_lib.linear_wgrad(x, dy, f16x3=True, relu_x=True)[0], _lib.linear_wgrad(torch.relu(x), dy, f16x3=True)[0])
    assert nf.check_saturation() == 0
    xb = x.clone()
    xb[b // 2, 0] = 1.0e6
    _lib.linear_wgrad(xb, dy, f16x3=True)
    with pytest.raises(nf.VcnfError):
        nf.check_saturation()


@pytest.mark.parametrize("ctx_dim", [16, None])
def test_resnet_training_uses_wgrad_kernel_and_matches_autograd(hip, ctx_dim):
    """ResidualNet at a training batch size: the dense layers' weight / bias gradients come from the weight-gradient
    kernel (vcnf_amd/autograd.py::LinearFn), every residual block is one autograd node with fused elementwise maps
    (ResBlockFn, csrc/resblock_ops.hip; with and without the context gate), and all gradients equal PyTorch's own
    autograd of the same network to rounding."""
    from vcnf_amd import _lib, autograd
    torch.manual_seed(3)
    net = nf.nets.ResidualNet(48 if ctx_dim else 64, 96, 128, context_features=ctx_dim, num_blocks=2).cuda()
    x = torch.randn(8192 + 5, 48 if ctx_dim else 64, device="cuda")
    ctx = torch.randn(8192 + 5, 16, device="cuda") if ctx_dim else None
    up = torch.randn(8192 + 5, 96, device="cuda")

    def grads(min_batch):
        old = autograd.WGRAD_MIN_BATCH
        autograd.WGRAD_MIN_BATCH = min_batch
        try:
            net.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_(True)
            events = []
            _lib.EVENT_SINK = events
            (net(xi, ctx) * up).sum().backward()
            _lib.EVENT_SINK = None
            return [xi.grad] + [p.grad.clone() for p in net.parameters()], len([e for e in events if e[2] == "linear_wgrad"])
        finally:
            autograd.WGRAD_MIN_BATCH = old
    mine, launches = grads(8192)
    ref, none = grads(1 << 40)
    assert launches == 1 + 2 * 2 + (2 if ctx_dim else 0) + 1 and none == 0     # initial, 2 x 2 block layers, context layers, final
    # ``mine`` also runs the forward products (and the square layers' input gradients) on the split-half kernel
    # (autograd.TRAIN_MATRIX_PATH): pre-activations differ from the library's at the 1e-7 level, so among the 4 M
    # hidden units of this batch about one has a pre-activation at rounding level and takes the other side of its ReLU.
    # That sample's contribution then moves between the two runs: its row of the input gradient, and ONE row (the
    # unit's) of a weight gradient by a single sample's term (~1 % of the largest entry at this batch size).  Allowed:
    # violations of the rounding-level bound in at most 2 % of a tensor's entries, none above 5 % of its largest entry.
    for i, (a, b) in enumerate(zip(mine, ref)):
        d = (a - b).abs()
        bad = d > 2e-4 * float(b.abs().max()) + 1e-6
        assert float(bad.float().mean()) <= 0.02 and float(d.max()) <= 0.05 * float(b.abs().max()) + 1e-6, \
            (i, float(bad.float().mean()), float(d.max()), float(b.abs().max()))
    # with the library's GEMMs on both sides (only the weight-gradient kernel differs) everything agrees to rounding
    autograd.TRAIN_MATRIX_PATH = 'fp32'
    try:
        mine32, _ = grads(8192)
    finally:
        autograd.TRAIN_MATRIX_PATH = 'fp16x3'
    for a, b in zip(mine32, ref):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-6



# ---------------------------------------------------------------- G23: against the REFERENCE's own autograd
def _g23_build(tag, fx):
    d = 9
    if tag.startswith("layer"):
        m = nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8)
        direction = tag.split("/")[1]
        return m, lambda mod, x, gz: (lambda z, ld: ld.reshape(-1).sum() + (z * gz).sum())(*getattr(mod, direction)(x))
    if tag.startswith("c3"):
        m = _c3_small(12)
        if tag.endswith("log_prob"):
            return m, lambda mod, x, c, gz: mod.log_prob(x, c).sum()
        return m, lambda mod, e, c, gz: (lambda z, lq: lq.sum() + (z * gz).sum())(*mod.sample_from(e, c))
    if tag.startswith("affine"):
        m = nf.flows.AffineCouplingBlock(nf.nets.MLP([16, 24, 24, 32], init_zeros=False), scale=True, scale_map="exp",
                                         split_mode="channel")
    else:
        m = nf.flows.MaskedAffineFlow(T(fx["masked/b"]), nf.nets.MLP([d, 16, d], init_zeros=False),
                                      nf.nets.MLP([d, 16, d], init_zeros=False))
    direction = tag.split("/")[1]
    return m, lambda mod, x, gz: (lambda z, ld: ld.reshape(-1).sum() + (z * gz).sum())(*getattr(mod, direction)(x))


def _g23_ids():
    from helpers import g23_cases
    return [c[0] for c in g23_cases()]


@pytest.mark.parametrize("idx", range(8), ids=_g23_ids())
def test_g23_hip_gradients_vs_reference_autograd(hip, idx):
    """The training path pinned to the reference: gradients of the HIP VJP kernels / closed-form VJPs
    against the gradients the reference's own autograd produced in fp64 (fixture G23: input gradients and two to four
    named parameters per case), judged like every other gradient test here - elementwise 2e-3 relative + 1e-3 of the
    rms + 8x the reference's own fp32 gradient error, and the error distribution within 3x the reference's fp32 one."""
    from helpers import g23_cases, g23_reference
    fx = fixture("g23_gradients")
    tag, seed, gains, in_names, gz_name, _, _ = g23_cases()[idx]
    sd, _ = state_for(fx, tag, seed, **gains)
    model, loss_fn = _g23_build(tag, fx)
    if tag.startswith("masked"):
        sd["b"] = model.state_dict()["b"]
    model.load_state_dict(sd)
    model = model.to("cuda").train()
    names, ref = g23_reference(fx, tag, len(in_names))
    xs = [dev(T(fx[n])).requires_grad_() for n in in_names]
    gz = dev(T(fx[gz_name])) if gz_name else None
    val = loss_fn(model, *xs, gz)
    val.backward()
    loss64 = ref["64"][0]
    assert abs(float(val.detach()) - loss64) <= 2e-5 * max(1.0, abs(loss64)), (float(val.detach()), loss64)
    params = dict(model.named_parameters())
    for i, xg in enumerate(xs):
        close(xg.grad, ref["64"][1][i], "%s input%d" % (tag, i), want32=ref["32"][1][i], atol=1e-3)
    for n in names:
        assert params[n].grad is not None, n
        close(params[n].grad, ref["64"][2][n], "%s %s" % (tag, n), want32=ref["32"][2][n], atol=1e-3,
              max_frac=5e-3 if tag.startswith("c3") else 2e-3)
