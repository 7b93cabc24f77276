"""fp64 variants of the elementwise kernels: a model converted with ``.double()`` - what the
reference's drivers do (/root/reference/run.py:114, rundiag.py:90, runadultvdeq.py:183: MaskedAffineFlow + ActNorm
stacks over Gaussian bases) - runs on the ``_f64`` entry points (csrc/affine_kernels.hip templates, csrc/rqs_f64.hip)
and reproduces the fixtures' fp64 outputs, which the reference itself produced in fp64.

Tolerance: |got - ref64| <= 1e-12 + 1e-11 |ref64| for the elementwise maps and Gaussian end caps; 1e-10 for stacks and
splines (different but equivalent summation orders in softmax / cumsum / row sums); permutations bit-exact."""
import numpy as np
import pytest
import torch

import vcnf_amd as nf
from helpers import fixture, T, state_for

pytestmark = pytest.mark.gpu


def close64(got, want, what, rtol=1e-11, atol=1e-12):
    assert got.dtype == torch.float64, what + ": result is not fp64"
    g = got.detach().cpu().numpy()
    w = np.asarray(want)
    assert g.shape == w.shape, (what, g.shape, w.shape)
    nan_g, nan_w = np.isnan(g), np.isnan(w)
    assert np.array_equal(nan_g, nan_w), what + ": NaN pattern differs"
    err = np.abs(np.where(nan_w, 0.0, g - w))
    bound = atol + rtol * np.abs(np.where(nan_w, 0.0, w))
    assert (err <= bound).all(), "%s: max err %.3e (bound %.3e)" % (what, float(err.max()), float(bound[err.argmax()] if err.ndim else bound))


def load64(module, sd):
    module.load_state_dict(sd, strict=True)
    return module.double().cuda().eval()


@pytest.mark.parametrize("d", [2, 32, 33])
@pytest.mark.parametrize("sm", ["exp", "sigmoid", "sigmoid_inv", "noscale"])
@pytest.mark.parametrize("mode", ["channel", "channel_inv"])
def test_g6_affine_coupling_block_f64(hip, d, sm, mode):
    fx = fixture("g6_affine")
    tag = "d%d/%s/%s" % (d, sm, mode)
    d1 = (d + 1) // 2
    cin, cout = (d1, d - d1) if mode == "channel" else (d - d1, d1)
    scale = sm != "noscale"
    blk = nf.flows.AffineCouplingBlock(nf.nets.MLP([cin, 24, 24, (2 if scale else 1) * cout]), scale=scale,
                                       scale_map=sm if scale else "exp", split_mode=mode)
    sd, _ = state_for(fx, tag, 601 + d, torch.float64)
    blk = load64(blk, sd)
    x = T(fx["d%d/x" % d], torch.float64).cuda()
    with torch.no_grad():
        for dirn, fn in (("fwd", blk.forward), ("inv", blk.inverse)):
            z, ld = fn(x.clone())
            close64(z, fx["%s/%s_z64" % (tag, dirn)], dirn + " z")
            close64(ld, fx["%s/%s_ld64" % (tag, dirn)], dirn + " ld")


@pytest.mark.parametrize("d", [2, 9, 30])
@pytest.mark.parametrize("variant", ["st", "t_only", "s_only", "inf"])
def test_g7_masked_affine_flow_f64(hip, d, variant):
    """The layer the reference's fp64 drivers stack (runadultvdeq.py:101-108)."""
    fx = fixture("g7_masked_affine")
    tag = "d%d/%s" % (d, variant)
    s = nf.nets.MLP([d, 16, d]) if variant != "t_only" else None
    t = nf.nets.MLP([d, 16, d]) if variant != "s_only" else None
    m = nf.flows.MaskedAffineFlow(T(fx["d%d/b" % d]), t, s)
    sd, _ = state_for(fx, tag, 701 + d, torch.float64)
    sd["b"] = T(fx["d%d/b" % d], torch.float64).view(1, -1)
    if variant == "inf":
        sd["s.net.2.bias"][1] = float("inf")
        sd["t.net.2.bias"][d - 1] = float("-inf")
    m = load64(m, sd)
    x = T(fx["d%d/x" % d], torch.float64).cuda()
    with torch.no_grad():
        for dirn, fn in (("fwd", m.forward), ("inv", m.inverse)):
            z, ld = fn(x)
            close64(z, fx["%s/%s_z64" % (tag, dirn)], dirn + " z")             # NaN pattern checked too
            close64(ld, fx["%s/%s_ld64" % (tag, dirn)], dirn + " ld")


def test_g8_permute_f64_bit_exact(hip):
    fx = fixture("g8_indices")
    for d in (2, 5, 32, 33):
        x = T(fx["swap/d%d/x" % d], torch.float64).cuda()
        p = nf.flows.Permute(d, mode="swap").cuda()
        assert np.array_equal(p.forward(x)[0].cpu().numpy(), fx["swap/d%d/fwd" % d].astype(np.float64))
        assert np.array_equal(p.inverse(x)[0].cpu().numpy(), fx["swap/d%d/inv" % d].astype(np.float64))
    g = torch.Generator().manual_seed(3)
    x = torch.randn(77, 64, dtype=torch.float64, generator=g).cuda()
    p = nf.flows.Permute(64, mode="shuffle").cuda()
    y, _ = p.forward(x)
    assert torch.equal(y, x[:, p.perm]) and torch.equal(p.inverse(y)[0], x)


@pytest.mark.parametrize("tag", ["d2_Tnone", "d64_Tnone", "d64_T0.7", "d33_T1.9"])
def test_g9_diag_gaussian_f64(hip, tag):
    fx = fixture("g9_diag_gaussian")
    d = int(tag[1:].split("_")[0])
    temp = fx[tag + "/temp"][0]
    q = nf.distributions.DiagGaussian(d)
    sd, _ = state_for(fx, tag, 901 + d, torch.float64)
    q = load64(q, sd)
    q.temperature = None if np.isnan(temp) else float(temp)
    with torch.no_grad():
        close64(q.log_prob(T(fx[tag + "/z"], torch.float64).cuda()), fx[tag + "/logp64"], "log_prob")
        z, lp = q.from_noise(T(fx[tag + "/eps64"]).cuda())
        close64(z, fx[tag + "/s_z64"], "sample z")
        close64(lp, fx[tag + "/s_logp64"], "sample logp")


def _affine_model(layers, d, widths):
    flows = []
    for _ in range(layers):
        flows += [nf.flows.AffineCouplingBlock(nf.nets.MLP(widths)), nf.flows.Permute(d, mode="swap")]
    return nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows)


@pytest.mark.parametrize("name,tag,layers,d,widths,seed", [
    ("g10_c1_two_moons", "c1", 4, 2, [1, 32, 32, 2], 1001),
    ("g12_c2_tabular", "c2", 8, 32, [16, 64, 64, 32], 1201)])
def test_affine_stacks_c1_c2_f64(hip, name, tag, layers, d, widths, seed):
    """NormalizingFlow.log_prob / sample of a .double() model: every layer on its fp64 kernel (the fused fp32 stack
    kernel declines fp64 inputs), against the reference's fp64 run."""
    fx = fixture(name)
    sd, _ = state_for(fx, tag, seed, torch.float64, weight_gain=0.4 if tag == "c2" else 1.0)
    model = load64(_affine_model(layers, d, widths), sd)
    with torch.no_grad():
        lp = model.log_prob(T(fx["x"], torch.float64).cuda())
        close64(lp, fx[tag + "/lp64"], "log_prob", rtol=1e-10, atol=1e-10)
        z, lq = model.sample_from(T(fx["eps"], torch.float64).cuda())
        close64(z, fx[tag + "/s_z64"], "sample z", rtol=1e-10, atol=1e-10)
        close64(lq, fx[tag + "/s_logq64"], "sample log_q", rtol=1e-10, atol=1e-10)


def test_actnorm_affine_const_f64(hip):
    """AffineConstFlow / ActNorm arithmetic (flows/affine/coupling.py:37-53) in fp64 against the same formula in torch."""
    g = torch.Generator().manual_seed(9)
    for shape in ((9,), (6, 1, 1)):
        flow = nf.flows.AffineConstFlow(shape).double().cuda()
        with torch.no_grad():
            flow.s.copy_(0.3 * torch.randn(flow.s.shape, dtype=torch.float64, generator=g))
            flow.t.copy_(torch.randn(flow.t.shape, dtype=torch.float64, generator=g))
            x = torch.randn((33, shape[0]) + ((4, 4) if len(shape) == 3 else ()), dtype=torch.float64, generator=g).cuda()
            z, ld = flow.forward(x)
            want = x * torch.exp(flow.s) + flow.t
            assert z.dtype == torch.float64 and float((z - want).abs().max()) <= 1e-13
            xb, ldb = flow.inverse(z)
            assert float((xb - x).abs().max()) <= 1e-12 and float((ld + ldb).abs().max()) <= 1e-12


@pytest.mark.parametrize("k", [8, 10, 16])
@pytest.mark.parametrize("inv", [False, True])
def test_g1_rational_quadratic_spline_f64(hip, k, inv):
    fx = fixture("g1_rqs")
    a = [T(fx["K%d/%s" % (k, n)], torch.float64).cuda() for n in ("x", "uw", "uh", "ud")]
    with torch.no_grad():
        y, ld = nf.utils.splines.rational_quadratic_spline(*a, inverse=inv)
    tag = "K%d_%s" % (k, "inv" if inv else "fwd")
    close64(y, fx[tag + "/y64"], "y", rtol=1e-10, atol=1e-10)
    close64(ld, fx[tag + "/ld64"], "ld", rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("case", [(8, 3.0), (8, 1.0), (16, 5.0), (5, 2.5)])
@pytest.mark.parametrize("inv", [False, True])
def test_g2_unconstrained_spline_tails_f64(hip, case, inv):
    k, tb = case
    fx = fixture("g2_rqs_tails")
    tag0 = "K%d_T%g" % (k, tb)
    a = [T(fx[tag0 + "/" + n], torch.float64).cuda() for n in ("x", "uw", "uh", "ud")]
    with torch.no_grad():
        y, ld = nf.utils.splines.unconstrained_rational_quadratic_spline(*a, inverse=inv, tails="linear", tail_bound=tb)
    tag = tag0 + ("_inv" if inv else "_fwd")
    close64(y, fx[tag + "/y64"], "y", rtol=1e-10, atol=1e-10)
    close64(ld, fx[tag + "/ld64"], "ld", rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("rm", [False, True])
def test_g3_crqs_layer_f64(hip, rm):
    """A whole RQS coupling layer of a .double() model: gather / conditioner on PyTorch-ROCm in fp64, both spline
    families on the fp64 elementwise kernel (csrc/rqs_f64.hip), against the reference's fp64 run."""
    fx = fixture("g3_crqs_layer")
    lay = nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, reverse_mask=rm)
    sd, _ = state_for(fx, "rm%d" % rm, 301 + rm, torch.float64, final_gain=2.0)
    lay = load64(lay, sd)
    x = T(fx["x"], torch.float64).cuda()
    with torch.no_grad():
        z, ld = lay.inverse(x)
        close64(z, fx["rm%d/inv_z64" % rm], "inverse z", rtol=1e-10, atol=1e-10)
        close64(ld, fx["rm%d/inv_ld64" % rm], "inverse ld", rtol=1e-9, atol=1e-9)
        z, ld = lay.forward(x)
        close64(z, fx["rm%d/fwd_z64" % rm], "forward z", rtol=1e-10, atol=1e-10)
        close64(ld, fx["rm%d/fwd_ld64" % rm], "forward ld", rtol=1e-9, atol=1e-9)
