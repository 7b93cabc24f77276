"""Parity of the HIP path (called through the C ABI via vcnf_amd) against the
reference's golden vectors and against the CPU oracle on fresh seeded inputs.

Tolerances (fp32 kernels vs the reference's fp32 AND fp64 outputs, helpers.parity):
  * every float result: |got - ref32| <= 2e-5 + 2e-5*|ref32| + 8*noise, where noise
    = max|ref32 - ref64| is the reference's own fp32 rounding noise on that fixture
    (0 for well-conditioned cases, up to 4e-3 for single splines with randn-scale
    logits, whose tiny bins / floor-level derivatives amplify rounding), and against
    fp64 the build's mean / max error stays within 2x / 4x the reference's own;
  * north star: on the well-conditioned C3 fixture (reference noise ~2e-6 relative)
    per-sample log_prob and sample log_q satisfy |got - ref32| <= 1e-5*|ref32| + 2e-5
    with no noise allowance (test_g5_c3_stack_log_prob_and_sample);
  * integer / index items and pure copies: bit-exact.
"""
import numpy as np
import pytest
import torch

import vcnf_amd as nf
from vcnf_amd import _lib
from helpers import (fixture, T, state_for, assert_close, within_reference_noise, parity,
                     oracle_rqs_coupling, oracle_c3_stack, oracle_crqs_stack, oracle_affine_stack, glow_state,
                     survey71_violators, oracle_round_trip, anchored)
from oracle import rqs as OR, layers as OL, nets as ON

pytestmark = pytest.mark.gpu

TOL = dict(rtol=2e-5, atol=2e-5)
LP = dict(rtol=1e-5, atol=2e-5)


def dev(t):
    return t.cuda()


def load(module, sd, fused=True):
    module.load_state_dict(sd, strict=True)
    return set_fused(module.cuda().eval(), fused)


SMALL_BATCH_ROWS_DEFAULT = 16384


@pytest.fixture(autouse=True)
def _default_tile_threshold():
    """set_fused(..., "fp16x3-t128") moves the process-wide small-batch threshold of the fused RQS layer; every test
    starts and ends on the default."""
    yield
    if torch.cuda.is_available():
        _lib.small_batch_rows(SMALL_BATCH_ROWS_DEFAULT)


def set_fused(module, flag):
    """Route eligible RQS couplings through the single-kernel layer - "fp16x3" (split-half
    fp16 matrix path, the default; at these batch sizes the 32-sample-tile kernel, csrc/fused_layer_v6s.hip),
    "fp16x3-t128" (the same path forced onto the 128-sample-tile kernel of large batches, csrc/fused_layer_v6.hip)
    or "fp32" (exact fp32 matrix path) - or, with False, through the gather kernel + torch GEMMs + spline kernel."""
    if isinstance(flag, str) and flag.startswith("fp16x3"):
        _lib.small_batch_rows(0 if flag == "fp16x3-t128" else SMALL_BATCH_ROWS_DEFAULT)
        flag = "fp16x3"
    for m in module.modules():
        if isinstance(m, nf.flows.PiecewiseRationalQuadraticCoupling):
            m.fused = bool(flag)
            m.fused_precision = flag if isinstance(flag, str) else None
    return module


FUSED = pytest.mark.parametrize("fused", ["fp16x3", "fp16x3-t128", "fp32", False],
                                ids=["fused-fp16x3", "fused-fp16x3-t128", "fused-fp32", "split"])


# ---------------------------------------------------------------- splines (G1, G2)
@pytest.mark.parametrize("k", [8, 10, 16])
@pytest.mark.parametrize("inv", [False, True])
def test_g1_rational_quadratic_spline(hip, k, inv):
    fx = fixture("g1_rqs")
    a = [dev(T(fx["K%d/%s" % (k, n)])) for n in ("x", "uw", "uh", "ud")]
    with torch.no_grad():
        y, ld = nf.utils.splines.rational_quadratic_spline(*a, inverse=inv)
    tag = "K%d_%s" % (k, "inv" if inv else "fwd")
    parity(y, fx[tag + "/y32"], fx[tag + "/y64"], what=tag + " y")
    parity(ld, fx[tag + "/ld32"], fx[tag + "/ld64"], what=tag + " ld")


@pytest.mark.parametrize("case", ["K8_T3", "K8_T1", "K16_T5", "K5_T2.5"])
@pytest.mark.parametrize("inv", [False, True])
def test_g2_unconstrained_spline_tails(hip, case, inv):
    fx = fixture("g2_rqs_tails")
    tb = float(case.split("_T")[1])
    a = [dev(T(fx["%s/%s" % (case, n)])) for n in ("x", "uw", "uh", "ud")]
    with torch.no_grad():
        y, ld = nf.utils.splines.unconstrained_rational_quadratic_spline(
            *a, inverse=inv, tails="linear", tail_bound=tb)
    tag = case + ("_inv" if inv else "_fwd")
    parity(y, fx[tag + "/y32"], fx[tag + "/y64"], what=tag + " y")
    parity(ld, fx[tag + "/ld32"], fx[tag + "/ld64"], what=tag + " ld")
    # outside the interval: bit-exact identity and exactly zero log-det
    x = fx[case + "/x"]
    out = ~((x >= -tb) & (x <= tb))
    assert out.any()
    assert np.array_equal(y.cpu().numpy()[out], x[out])
    assert np.all(ld.cpu().numpy()[out] == 0.0)


def test_spline_round_trip_properties(hip):
    """The reference's own test (utils/splines_test.py:6-56): inverse(forward(x))
    == x and the two log-dets cancel, eps 1e-4 - on its shape [2,3,4] with K=10.
    Larger random draws contain ill-conditioned elements (tiny bins, floor-level
    derivatives) where fp32 cannot hold 1e-4; there the bound is the oracle's own
    fp32 round-trip error on the same inputs (x4) plus 1e-4."""
    g = torch.Generator().manual_seed(5)
    for shape, k in (((2, 3, 4), 10), ((1000, 7), 8), ((5,), 3)):
        uw = torch.randn(*shape, k, generator=g)
        uh = torch.randn(*shape, k, generator=g)
        ud = torch.randn(*shape, k + 1, generator=g)
        x = torch.rand(*shape, generator=g)
        oy, old_ = OR.rq_spline(x, uw, uh, ud)
        oxr, oldi = OR.rq_spline(oy, uw, uh, ud, inverse=True)
        tol_x = 1e-4 + 4 * float((oxr - x).abs().max())
        tol_l = 1e-4 + 4 * float((old_ + oldi).abs().max())
        with torch.no_grad():
            y, ld = nf.utils.splines.rational_quadratic_spline(dev(x), dev(uw), dev(uh), dev(ud))
            xr, ldi = nf.utils.splines.rational_quadratic_spline(y, dev(uw), dev(uh), dev(ud), inverse=True)
        assert y.shape == x.shape and torch.isfinite(y).all() and torch.isfinite(ld).all()
        assert_close(xr, x, rtol=0, atol=tol_x, what="round trip")
        assert_close(ld + ldi, torch.zeros(shape), rtol=0, atol=tol_l, what="log-det cancel")
        udl = torch.randn(*shape, k - 1, generator=g)
        x3 = 3 * torch.randn(*shape, generator=g)
        oy, old_ = OR.rq_spline_tails(x3, uw, uh, udl, tail_bound=1.0)
        oxr, oldi = OR.rq_spline_tails(oy, uw, uh, udl, inverse=True, tail_bound=1.0)
        tol_x = 1e-4 + 4 * float((oxr - x3).abs().max())
        tol_l = 1e-4 + 4 * float((old_ + oldi).abs().max())
        with torch.no_grad():
            y, ld = nf.utils.splines.unconstrained_rational_quadratic_spline(
                dev(x3), dev(uw), dev(uh), dev(udl), tail_bound=1.0)
            xr, ldi = nf.utils.splines.unconstrained_rational_quadratic_spline(
                y, dev(uw), dev(uh), dev(udl), inverse=True, tail_bound=1.0)
        assert_close(xr, x3, rtol=0, atol=tol_x, what="tails round trip")
        assert_close(ld + ldi, torch.zeros(shape), rtol=0, atol=tol_l, what="tails log-det cancel")
    nf.check_discriminant()


def test_spline_generic_bin_counts_vs_oracle(hip):
    """K values without a compile-time specialisation go through the runtime-K path."""
    g = torch.Generator().manual_seed(6)
    for k in (2, 3, 5, 7, 12, 24, 33):
        uw, uh = torch.randn(300, k, generator=g), torch.randn(300, k, generator=g)
        ud = torch.randn(300, k - 1, generator=g)
        x = 2.5 * torch.randn(300, generator=g)
        for inv in (False, True):
            want_y, want_ld = OR.rq_spline_tails(x.double(), uw.double(), uh.double(), ud.double(),
                                                 inverse=inv, tail_bound=2.0)
            with torch.no_grad():
                y, ld = nf.utils.splines.unconstrained_rational_quadratic_spline(
                    dev(x), dev(uw), dev(uh), dev(ud), inverse=inv, tail_bound=2.0)
            assert_close(y, want_y, rtol=1e-4, atol=1e-4, what="K=%d y" % k)
            assert_close(ld, want_ld, rtol=1e-3, atol=1e-3, what="K=%d ld" % k)


# ---------------------------------------------------------------- RQS coupling layers (G3, G4)
@FUSED
@pytest.mark.parametrize("rm", [0, 1])
def test_g3_coupled_rqs_layer(hip, rm, fused):
    fx = fixture("g3_crqs_layer")
    sd, _ = state_for(fx, "rm%d" % rm, 301 + rm, final_gain=2.0)
    m = load(nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, reverse_mask=bool(rm)), sd, fused)
    x = dev(T(fx["x"]))
    pre = "rm%d/" % rm
    with torch.no_grad():
        z, ld = m.inverse(x)
        parity(z, fx[pre + "inv_z32"], fx[pre + "inv_z64"], what="inverse z")
        parity(ld, fx[pre + "inv_ld32"], fx[pre + "inv_ld64"], what="inverse ld")
        z, ld = m.forward(x)
        parity(z, fx[pre + "fwd_z32"], fx[pre + "fwd_z64"], what="forward z")
        parity(ld, fx[pre + "fwd_ld32"], fx[pre + "fwd_ld64"], what="forward ld")
        assert ld.shape == (256,)
    nf.check_discriminant()


def _cond_layer(tag, fx, fused=True):
    d, c, h, nb, k, tb, kind = fx[tag + "/cfg"]
    d, c, h, nb, k = int(d), int(c), int(h), int(nb), int(k)
    mask = (nf.utils.create_alternating_binary_mask(d, even=False) if kind == 0
            else nf.utils.create_mid_split_binary_mask(d))
    net = lambda i, o: nf.nets.ResidualNet(i, o, hidden_features=h, context_features=c, num_blocks=nb)
    m = nf.flows.PiecewiseRationalQuadraticCoupling(mask, net, num_bins=k, tails="linear", tail_bound=float(tb),
                                                    apply_unconditional_transform=True)
    sd, _ = state_for(fx, tag, 401 + d, final_gain=2.0)
    return load(m, sd, fused), d


@FUSED
@pytest.mark.parametrize("tag", ["d64", "d21", "d7k4"])
def test_g4_conditional_coupling(hip, tag, fused):
    fx = fixture("g4_cond_prqc")
    m, d = _cond_layer(tag, fx, fused)
    x, ctx = dev(T(fx[tag + "/x"])), dev(T(fx[tag + "/ctx"]))
    with torch.no_grad():
        z, ld = m.forward(x, ctx)
        parity(z, fx[tag + "/nsf_fwd_z32"], fx[tag + "/nsf_fwd_z64"], what="nsf forward z")
        parity(ld, fx[tag + "/nsf_fwd_ld32"], fx[tag + "/nsf_fwd_ld64"], what="nsf forward ld")
        z, ld = m.inverse(x, ctx)
        parity(z, fx[tag + "/nsf_inv_z32"], fx[tag + "/nsf_inv_z64"], what="nsf inverse z")
        parity(ld, fx[tag + "/nsf_inv_ld32"], fx[tag + "/nsf_inv_ld64"], what="nsf inverse ld")
    nf.check_discriminant()


def test_coupling_properties(hip):
    """flows/neural_spline/coupling_test.py of the reference restated: shape and
    finiteness, identity features untouched without the unconditional transform
    and changed with it, round trip <= 1e-4."""
    torch.manual_seed(3)
    for d in (20, 7, 2):
        mask = nf.utils.create_mid_split_binary_mask(d)
        for uncond in (False, True):
            net = lambda i, o: nf.nets.ResidualNet(i, o, hidden_features=30, num_blocks=5)
            m = nf.flows.PiecewiseRationalQuadraticCoupling(mask, net, tails="linear", tail_bound=2.0,
                                                            apply_unconditional_transform=uncond).cuda()
            if uncond:
                with torch.no_grad():
                    for p in m.unconditional_transform.parameters():
                        p.normal_(0, 0.5)
            x = dev(torch.randn(10, d))
            with torch.no_grad():
                y, ld = m(x)
                xr, ldi = m.inverse(y)
            assert y.shape == x.shape and ld.shape == (10,)
            assert torch.isfinite(y).all() and torch.isfinite(ld).all()
            idf = (mask <= 0).cuda()
            if uncond:
                assert not torch.equal(y[:, idf], x[:, idf])
            else:
                assert torch.equal(y[:, idf], x[:, idf])          # bit-exact pass-through
            assert_close(xr, x.cpu(), rtol=0, atol=1e-4, what="round trip d=%d" % d)
            assert_close(ld + ldi, torch.zeros(10), rtol=0, atol=1e-4, what="log-det cancel")


def test_conditioner_input_matches_kernel_output(hip):
    """Sampling direction: the identity values handed to the conditioner and the
    ones written to the output come from two kernels and must agree bitwise."""
    fx = fixture("g4_cond_prqc")
    m, d = _cond_layer("d64", fx, fused=False)
    x, ctx = dev(T(fx["d64/x"])), dev(T(fx["d64/ctx"]))
    with torch.no_grad():
        first = _lib.rqs_conditioner_input(x, m._index32('id'), ctx, m.unconditional_transform.logits(),
                                           m._cfg(False), True)
        y, _ = m.inverse(x, ctx)
    assert torch.equal(first[:, :32], y[:, m.identity_features])
    assert torch.equal(first[:, 32:], ctx)


# ---------------------------------------------------------------- C3 stack (G5)
def _c3_model(layers=12, d=64, c=16, hidden=128, blocks=2, k=8):
    flows = [nf.flows.CoupledRationalQuadraticSpline(d, blocks, hidden, k, reverse_mask=bool(i % 2),
                                                     num_context_channels=c) for i in range(layers)]
    return nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows)


@FUSED
def test_g5_c3_stack_log_prob_and_sample(hip, fused):
    """North-star parity: config C3's stack on the well-conditioned fixture."""
    fx = fixture("g5_c3_stack")
    sd, _ = state_for(fx, "c3", 501, final_gain=1.0)
    model = load(_c3_model(), sd, fused)
    x, ctx, eps = (dev(T(fx[n])) for n in ("x", "ctx", "eps"))
    with torch.no_grad():
        lp = model.log_prob(x, ctx)
        assert_close(lp, fx["c3/lp32"], what="log_prob (1e-5 rel)", **LP)
        within_reference_noise(lp, fx["c3/lp32"], fx["c3/lp64"], what="log_prob vs fp64")
        z, lq = model.sample_from(eps, ctx)
        parity(z, fx["c3/s_z32"], fx["c3/s_z64"], what="sample z")
        parity(lq, fx["c3/s_logq32"], fx["c3/s_logq64"], rtol=1e-5, atol=2e-5, what="sample log_q")
        # inverse reconstruction: the density pass maps the sampled z back onto the base noise
        zz, lds = z, []
        for f in reversed(model.flows):
            zz, ld = f.inverse(zz, context=ctx)
        rec = (zz.cpu() - (eps.cpu() * torch.exp(model.q0.log_scale.cpu()) + model.q0.loc.cpu())).abs()
        # anchored to the oracle's own fp32 round trip on the same inputs (not to a constant)
        o_lq, o_rec = oracle_round_trip(oracle_c3_stack({k: v.detach().cpu() for k, v in model.state_dict().items()}),
                                        eps.cpu(), ctx.cpu())
        anchored(rec, o_rec, "G5 reconstruction |z0 - base|")
        lp_rt = model.log_prob(z, ctx)
        anchored((lp_rt - lq).abs() / (1.0 + lq.abs()), o_lq, "G5 round trip log_q")
        # SURVEY 7.1's elementwise criterion, reported (see helpers.survey71_violators)
        for name, got, r32, r64 in (("log_prob", lp, "c3/lp32", "c3/lp64"), ("sample z", z, "c3/s_z32", "c3/s_z64"),
                                    ("sample log_q", lq, "c3/s_logq32", "c3/s_logq64")):
            assert survey71_violators(got, fx[r32], fx[r64], "G5 " + name) < 0.25
        # layer by layer through the plain Flow contract (no in-kernel accumulation)
        zz, lds = x, []
        for f in reversed(model.flows):
            zz, ld = f.inverse(zz, context=ctx)
            lds.append(ld)
        parity(torch.stack(lds), fx["c3/lp_lds32"], fx["c3/lp_lds64"], what="per-layer log_dets")
        parity(zz, fx["c3/lp_z32"], fx["c3/lp_z64"], what="latent")
    nf.check_discriminant()


@FUSED
def test_g5_c3_stack_stress_weights(hip, fused):
    """Same stack with wild conditioner logits (derivatives at the 1e-3 floor): the
    reference's own fp32 log_prob is up to 0.3 away from fp64 here; the build must
    stay inside that noise envelope."""
    fx = fixture("g5_c3_stack")
    sd, _ = state_for(fx, "c3_stress", 501, final_gain=6.0)
    sd.update({k[len("c3/int/"):]: T(v) for k, v in fx.items() if k.startswith("c3/int/")})
    model = load(_c3_model(), sd, fused)
    x, ctx, eps = (dev(T(fx[n])) for n in ("x", "ctx", "eps"))
    with torch.no_grad():
        parity(model.log_prob(x, ctx), fx["c3_stress/lp32"], fx["c3_stress/lp64"], what="log_prob")
        z, lq = model.sample_from(eps, ctx)
        parity(z, fx["c3_stress/s_z32"], fx["c3_stress/s_z64"], what="sample z")
        parity(lq, fx["c3_stress/s_logq32"], fx["c3_stress/s_logq64"], what="sample log_q")
    nf.check_discriminant()


# ---------------------------------------------------------------- affine (G6), masked affine (G7)
@pytest.mark.parametrize("d", [2, 32, 33])
@pytest.mark.parametrize("sm", ["exp", "sigmoid", "sigmoid_inv", "noscale"])
@pytest.mark.parametrize("mode", ["channel", "channel_inv"])
def test_g6_affine_coupling_block(hip, d, sm, mode):
    fx = fixture("g6_affine")
    tag = "d%d/%s/%s" % (d, sm, mode)
    d1 = (d + 1) // 2
    cin, cout = (d1, d - d1) if mode == "channel" else (d - d1, d1)
    scale = sm != "noscale"
    blk = nf.flows.AffineCouplingBlock(nf.nets.MLP([cin, 24, 24, (2 if scale else 1) * cout]), scale=scale,
                                       scale_map=sm if scale else "exp", split_mode=mode)
    sd, _ = state_for(fx, tag, 601 + d)
    blk = load(blk, sd)
    x = dev(T(fx["d%d/x" % d]))
    with torch.no_grad():
        for dirn, fn in (("fwd", blk.forward), ("inv", blk.inverse)):
            z, ld = fn(x.clone())
            assert_close(z, fx["%s/%s_z32" % (tag, dirn)], what=dirn + " z", **TOL)
            assert_close(ld, fx["%s/%s_ld32" % (tag, dirn)], what=dirn + " ld", **TOL)
        # the unfused composition Split -> AffineCoupling -> Merge gives the same numbers
        pair, _ = blk.flows[0](x.clone())
        pair, ld2 = blk.flows[1](pair)
        z2, _ = blk.flows[2](pair)
        assert_close(z2, fx[tag + "/fwd_z32"], what="composed z", **TOL)


@pytest.mark.parametrize("d", [2, 9, 30])
@pytest.mark.parametrize("variant", ["st", "t_only", "s_only", "inf"])
def test_g7_masked_affine_flow(hip, d, variant):
    fx = fixture("g7_masked_affine")
    tag = "d%d/%s" % (d, variant)
    s = nf.nets.MLP([d, 16, d]) if variant != "t_only" else None
    t = nf.nets.MLP([d, 16, d]) if variant != "s_only" else None
    m = nf.flows.MaskedAffineFlow(T(fx["d%d/b" % d]), t, s)
    sd, _ = state_for(fx, tag, 701 + d)
    sd["b"] = T(fx["d%d/b" % d]).view(1, -1)
    if variant == "inf":
        sd["s.net.2.bias"][1] = float("inf")
        sd["t.net.2.bias"][d - 1] = float("-inf")
    m = load(m, sd)
    x = dev(T(fx["d%d/x" % d]))
    with torch.no_grad():
        for dirn, fn in (("fwd", m.forward), ("inv", m.inverse)):
            z, ld = fn(x)
            assert_close(z, fx["%s/%s_z32" % (tag, dirn)], what=dirn + " z", **TOL)     # NaN pattern checked too
            assert_close(ld, fx["%s/%s_ld32" % (tag, dirn)], what=dirn + " ld", **TOL)


# ---------------------------------------------------------------- permutations (G8), Gaussian (G9)
def test_g8_permute_bit_exact(hip):
    fx = fixture("g8_indices")
    for d in (2, 5, 32, 33):
        x = dev(T(fx["swap/d%d/x" % d]))
        p = nf.flows.Permute(d, mode="swap").cuda()
        assert np.array_equal(p.forward(x)[0].cpu().numpy(), fx["swap/d%d/fwd" % d])
        assert np.array_equal(p.inverse(x)[0].cpu().numpy(), fx["swap/d%d/inv" % d])
    for d in (5, 64, 1024):
        torch.manual_seed(7)
        p = nf.flows.Permute(d, mode="shuffle").cuda()
        x = torch.randn(33, d)
        y = p.forward(dev(x))[0]
        assert torch.equal(y.cpu(), x[:, torch.as_tensor(fx["perm/d%d_s7/perm" % d])])
        assert torch.equal(p.inverse(y)[0].cpu(), x)
    x4 = torch.randn(3, 6, 4, 5)
    p = nf.flows.Permute(6, mode="swap").cuda()
    assert torch.equal(p.forward(dev(x4))[0].cpu(), torch.cat([x4[:, 3:], x4[:, :3]], 1))


@pytest.mark.parametrize("tag", ["d2_Tnone", "d64_Tnone", "d64_T0.7", "d33_T1.9"])
def test_g9_diag_gaussian(hip, tag):
    fx = fixture("g9_diag_gaussian")
    d = int(tag[1:].split("_")[0])
    temp = fx[tag + "/temp"][0]
    q = nf.distributions.DiagGaussian(d)
    sd, _ = state_for(fx, tag, 901 + d)
    q = load(q, sd)
    q.temperature = None if np.isnan(temp) else float(temp)
    with torch.no_grad():
        assert_close(q.log_prob(dev(T(fx[tag + "/z"]))), fx[tag + "/logp32"], what="log_prob", **TOL)
        z, lp = q.from_noise(dev(T(fx[tag + "/eps32"])))
        assert_close(z, fx[tag + "/s_z32"], what="sample z", **TOL)
        assert_close(lp, fx[tag + "/s_logp32"], what="sample logp", **TOL)
        z, lp = q(17)
        assert z.shape == (17, d) and lp.shape == (17,)


# ---------------------------------------------------------------- affine stacks C1 / C2 (G10, G12)
def _affine_model(layers, d, widths):
    flows = []
    for _ in range(layers):
        flows += [nf.flows.AffineCouplingBlock(nf.nets.MLP(widths)), nf.flows.Permute(d, mode="swap")]
    return nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows)


@pytest.mark.parametrize("name,tag,layers,d,widths,seed", [
    ("g10_c1_two_moons", "c1", 4, 2, [1, 32, 32, 2], 1001),
    ("g12_c2_tabular", "c2", 8, 32, [16, 64, 64, 32], 1201)])
def test_affine_stacks_c1_c2(hip, name, tag, layers, d, widths, seed):
    fx = fixture(name)
    sd, _ = state_for(fx, tag, seed, weight_gain=0.4 if tag == "c2" else 1.0)
    model = load(_affine_model(layers, d, widths), sd)
    with torch.no_grad():
        lp = model.log_prob(dev(T(fx["x"])))
        assert_close(lp, fx[tag + "/lp32"], what="log_prob", **LP)
        z, lq = model.sample_from(dev(T(fx["eps"])))
        assert_close(z, fx[tag + "/s_z32"], what="sample z", **TOL)
        assert_close(lq, fx[tag + "/s_logq32"], what="sample log_q", **LP)


@pytest.mark.parametrize("d,widths,split,scale_map", [
    (32, [16, 64, 64, 32], "channel", "exp"),                    # config C2's stack
    (2, [1, 32, 32, 2], "channel", "exp"),                       # config C1's stack
    (24, [12, 128, 128, 24], "channel_inv", "sigmoid_inv"),
])
def test_affine_stack_split_half_matrix_path(hip, d, widths, split, scale_map):
    """The DEFAULT form of the stack kernel since round 3: second and third dense layer of every conditioner on the fp16
    split-half matrix path (first layer exact fp32).  Against the exact-fp32 stack kernel and the fp64 oracle: the
    split-half result is at least as close to fp64 as the fp32 kernel's (mean error), and both agree to 1e-5 relative;
    a wave whose hidden activations leave the fp16 range takes the fp32 body - those rows are BITWISE the fp32 kernel's."""
    from helpers import oracle_affine_stack
    torch.manual_seed(77 + d)
    flows = []
    for _ in range(8):
        flows.append(nf.flows.AffineCouplingBlock(nf.nets.MLP(widths, init_zeros=False), scale_map=scale_map, split_mode=split))
        flows.append(nf.flows.Permute(d, mode="swap"))
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows).cuda().eval()
    b = 4096 + 19
    x, eps = torch.randn(b, d, device="cuda"), torch.randn(b, d, device="cuda")
    def run(prec):
        for f in model.flows:
            f.fused_precision = prec
        with torch.no_grad():
            return model.log_prob(x), model.sample_from(eps)
    nf.range_redo_count()
    lp_h, (z_h, lq_h) = run("fp16x3")
    assert nf.range_redo_count() == 0
    lp_f, (z_f, lq_f) = run("fp32")
    assert not torch.equal(z_h, z_f) or d == 2                     # it really is another arithmetic
    assert_close(lp_h, lp_f.cpu(), rtol=1e-5, atol=2e-5, what="split-half vs fp32 stack log_prob")
    assert_close(z_h, z_f.cpu(), rtol=1e-5, atol=2e-5, what="split-half vs fp32 stack sample z")
    if split == "channel" and scale_map == "exp":                  # the oracle stack helper covers this block form
        sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in model.state_dict().items()}
        want = oracle_affine_stack(sd64, 8, d, leaky=0.0).log_prob(x.cpu().double())
        e_h = float((lp_h.cpu().double() - want).abs().mean())
        e_f = float((lp_f.cpu().double() - want).abs().mean())
        print("\naffine stack d=%d: mean |log_prob - fp64|: split-half %.3e, exact fp32 %.3e" % (d, e_h, e_f))
        assert e_h <= 1.25 * e_f + 1e-7
    # range: scale the first layer of one block up (and its second layer down by the same factor, so that the outputs
    # stay ordinary) until its hidden activations pass 65504
    with torch.no_grad():
        net = model.flows[4].flows[1].param_map.net
        net[0].weight.mul_(2.0e5)
        net[0].bias.mul_(2.0e5)
        net[2].weight.mul_(5.0e-6)
    lp_h, (z_h, lq_h) = run("fp16x3")
    redone = nf.range_redo_count()
    lp_f, (z_f, lq_f) = run("fp32")
    assert redone > 0, "no wave left the fp16 range: the test does not test the fallback"
    big = ~torch.isclose(lp_h, lp_f, rtol=1e-4, atol=1e-3)
    assert not big.any(), "a clamped value leaked: %d rows differ grossly" % int(big.sum())
    assert torch.isfinite(lp_h).all()


def test_affine_stack_declines_more_than_128_features(hip):
    """D = 129 with split_mode 'channel_inv' and no scale passes the single-layer kernel's shape check but
    not the stack kernel's LDS bound (two strips per wave: 512 D bytes <= 64 KB).  The planner now asks
    vcnf_affine_stack_fused_supported and runs such layers one launch each instead of raising."""
    from vcnf_amd import fused_affine
    torch.manual_seed(129)
    d = 129
    flows = []
    for _ in range(3):
        flows.append(nf.flows.AffineCouplingBlock(nf.nets.MLP([64, 64, 64, 65], init_zeros=False), scale=False,
                                                  split_mode="channel_inv"))
        flows.append(nf.flows.Permute(d, mode="shuffle"))
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows).cuda().eval()
    x = torch.randn(300, d, device="cuda")
    with torch.no_grad():
        assert model.flows[0].fusable(x)                                       # the one-layer kernel takes the shape
        assert fused_affine.plan_stack(list(model.flows), 0, x, False) is None  # the stack kernel does not
        model.fuse_affine_stacks = True
        lp1 = model.log_prob(x)
        z1, lq1 = model.sample_from(x)
        model.fuse_affine_stacks = False
        lp0 = model.log_prob(x)
        z0, lq0 = model.sample_from(x)
    assert torch.equal(lp1, lp0) and torch.equal(z1, z0) and torch.equal(lq1, lq0)


@pytest.mark.parametrize("d,widths,mode,split,scale_map", [
    (32, [16, 64, 64, 32], "swap", "channel", "exp"),            # config C2's stack
    (33, [17, 32, 32, 32], "shuffle", "channel", "sigmoid"),      # odd width, random permutations
    (2, [1, 32, 32, 2], "swap", "channel", "exp"),                # config C1's stack
    (24, [12, 64, 64, 24], "shuffle", "channel_inv", "sigmoid_inv"),
])
def test_affine_stack_single_launch_matches_per_layer(hip, d, widths, mode, split, scale_map):
    """vcnf_affine_stack_fused_f32: a run of [AffineCouplingBlock, Permute] pairs executed by ONE launch
    (NormalizingFlow.fuse_affine_stacks, default) gives the outputs of the one-launch-per-layer path bit for bit
    (same layer body; log|det| to rounding), in both directions, for ragged batches; the per-layer path is the one pinned to the oracle
    and the reference's fixtures (G6, G10, G12)."""
    torch.manual_seed(41 + d)
    flows = []
    for _ in range(8):
        flows.append(nf.flows.AffineCouplingBlock(nf.nets.MLP(widths, init_zeros=False), scale_map=scale_map, split_mode=split))
        flows.append(nf.flows.Permute(d, mode=mode))
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows).cuda().eval()
    for f in model.flows:
        f.fused_precision = "fp32"            # bit-for-bit against the per-layer kernel: both on exact fp32 matrix instructions
    for b in (1, 15, 64, 1000 + 37):
        x, eps = torch.randn(b, d, device="cuda"), torch.randn(b, d, device="cuda")
        with torch.no_grad():
            model.fuse_affine_stacks = True
            lp1 = model.log_prob(x)
            z1, lq1 = model.sample_from(eps)
            model.fuse_affine_stacks = False
            lp0 = model.log_prob(x)
            z0, lq0 = model.sample_from(eps)
        # outputs bit for bit (same layer body); log|det| is summed over the layers in a register instead of one
        # read-modify-write of the [B] buffer per layer: same terms, different association
        assert torch.equal(z1, z0), (d, b)
        assert_close(lp1, lp0.cpu(), rtol=2e-6, atol=2e-5, what="stack log_prob d=%d b=%d" % (d, b))
        assert_close(lq1, lq0.cpu(), rtol=2e-6, atol=2e-5, what="stack sample log_q d=%d b=%d" % (d, b))
    # the run's concatenated weight buffer follows the parameters: an in-place update (bumps _version) is seen at the
    # next call in BOTH directions, at the same address (captured HIP graphs keep reading it); a .data edit after
    # refresh_packed
    with torch.no_grad():
        model.fuse_affine_stacks = True
        model.log_prob(x)
        addr = model.flows[0].__dict__['_fused_affine_stack']['wpack'].data_ptr()
        for step in range(2):
            for p in model.parameters():
                if step == 0:
                    p.add_(0.01 * torch.randn_like(p))
                else:
                    p.data.add_(0.01 * torch.randn_like(p))
            if step == 1:
                nf.refresh_packed(model)
            model.fuse_affine_stacks = True
            lp1 = model.log_prob(x)
            z1, _ = model.sample_from(eps)
            assert model.flows[0].__dict__['_fused_affine_stack']['wpack'].data_ptr() == addr
            model.fuse_affine_stacks = False
            assert_close(lp1, model.log_prob(x).cpu(), rtol=2e-6, atol=2e-5, what="stack after weight update %d" % step)
            assert torch.equal(z1, model.sample_from(eps)[0])
    # a run interrupted by another flow: two launches around it, same results
    model.flows.insert(8, nf.flows.AffineConstFlow((d,)).cuda())
    with torch.no_grad():
        model.flows[8].s.normal_(0, 0.3); model.flows[8].t.normal_(0, 0.3)
        x = torch.randn(200, d, device="cuda")
        model.fuse_affine_stacks = True
        lp1 = model.log_prob(x)
        model.fuse_affine_stacks = False
        assert_close(lp1, model.log_prob(x).cpu(), rtol=2e-6, atol=2e-5, what="interrupted run")


# ---------------------------------------------------------------- fresh inputs vs the oracle, ragged sizes
def _oracle_pair(sd, prefix, k, tb, hid):
    o32 = oracle_rqs_coupling(sd, prefix, k, tb, hid)
    o64 = oracle_rqs_coupling({n: v.double() if v.is_floating_point() else v for n, v in sd.items()},
                              prefix, k, tb, hid)
    return o32, o64


def _check_vs_oracle(m, o32, o64, x, ctx, what, noise=None):
    """Both directions of a wrapper layer against the oracle run in fp32 and fp64.
    Returns the oracle's fp32 noise per (direction, output) for reuse as a floor."""
    seen = {}
    with torch.no_grad():
        for dirn in ("inverse", "forward"):
            a32 = (x,) if ctx is None else (x, ctx)
            a64 = tuple(t.double() for t in a32)
            w32, l32 = getattr(o32, dirn)(*a32)
            w64, l64 = getattr(o64, dirn)(*a64)
            z, ld = getattr(m, dirn)(dev(x)) if ctx is None else getattr(m, dirn)(dev(x), context=dev(ctx))
            nz, nl = (noise or {}).get(dirn, (0.0, 0.0))
            parity(z, w32, w64, what="%s %s z" % (what, dirn), noise_floor=nz)
            parity(ld, l32, l64, what="%s %s ld" % (what, dirn), noise_floor=nl)
            seen[dirn] = (float((w32 - w64).abs().max()), float((l32 - l64).abs().max()))
    return seen


@FUSED
def test_c3_layer_vs_oracle_ragged_batches(hip, fused):
    """Batch sizes around the tile size (8 samples per workgroup pass), 1 row, and
    a large odd size; the oracle's fp32 noise measured at B=4099 is the floor for
    the tiny batches."""
    fx = fixture("g5_c3_stack")
    sd, _ = state_for(fx, "c3", 501, final_gain=2.0)
    sub = {k: v for k, v in sd.items() if k.startswith("flows.0.")}
    m = load(nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, num_context_channels=16),
             {k[len("flows.0."):]: v for k, v in sub.items()}, fused)
    o32, o64 = _oracle_pair(sub, "flows.0.prqct.", 8, 3.0, 128)
    noise = None
    for batch in (4099, 257, 129, 128, 127, 9, 8, 7, 3, 1):
        g = torch.Generator().manual_seed(batch)
        x, ctx = 1.2 * torch.randn(batch, 64, generator=g), torch.randn(batch, 16, generator=g)
        seen = _check_vs_oracle(m, o32, o64, x, ctx, "B=%d" % batch, noise)
        noise = noise or seen


def test_empty_batch(hip):
    m = nf.flows.CoupledRationalQuadraticSpline(8, 1, 16).cuda()
    blk = nf.flows.AffineCouplingBlock(nf.nets.MLP([4, 8, 8])).cuda()
    with torch.no_grad():
        z, ld = m.inverse(torch.empty(0, 8, device="cuda"))
        assert z.shape == (0, 8) and ld.shape == (0,)
        z, ld = blk.forward(torch.empty(0, 8, device="cuda"))
        assert z.shape == (0, 8) and ld.shape == (0,)


def test_wide_layer_chunked_params_vs_oracle(hip):
    """D=1024, K=16 (config C5's layer shape): the params tile does not fit LDS
    whole, the kernel streams it in feature chunks and the unconditional spline
    reads its logits from L2 instead of LDS tables."""
    torch.manual_seed(9)
    d, k, h = 1024, 16, 32
    m = nf.flows.CoupledRationalQuadraticSpline(d, 1, h, k).cuda()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "unnormalized" in n:
                p.normal_(0, 0.5)
            elif "final_layer.weight" in n:
                p.normal_(0, 6.0 / np.sqrt(h))
    sd = {k_: v.detach().cpu() for k_, v in m.state_dict().items()}
    o32, o64 = _oracle_pair(sd, "prqct.", k, 3.0, h)
    _check_vs_oracle(m, o32, o64, 1.3 * torch.randn(37, d), None, "D=1024")
    nf.check_discriminant()


def test_odd_shapes_vs_oracle(hip):
    """Odd feature counts, even P (bank-conflicting stride), unaligned row lengths."""
    torch.manual_seed(10)
    for d, k, hid in ((3, 5, 8), (11, 7, 16), (33, 4, 8), (2, 8, 8), (130, 10, 16)):
        for rm in (False, True):
            m = nf.flows.CoupledRationalQuadraticSpline(d, 1, hid, k, tail_bound=2.0, reverse_mask=rm).cuda()
            with torch.no_grad():
                for n, p in m.named_parameters():
                    if "unnormalized" in n:
                        p.normal_(0, 0.5)
                    elif "final_layer.weight" in n:
                        p.normal_(0, 2.0 / np.sqrt(hid))
            sd = {k_: v.detach().cpu() for k_, v in m.state_dict().items()}
            o32, o64 = _oracle_pair(sd, "prqct.", k, 2.0, hid)
            _check_vs_oracle(m, o32, o64, 1.2 * torch.randn(101, d), None, "d=%d K=%d" % (d, k))


def test_affine_image_shaped_vs_oracle(hip):
    """4-D inputs (Glow family, config C4's coupling): channel split with a conv
    conditioner, sigmoid scale map, reduction over all non-batch dims."""
    torch.manual_seed(12)
    for c, hw, mode in ((12, 4, "channel"), (7, 3, "channel_inv"), (4, 8, "channel")):
        head = c - c // 2
        cin, cout = (head, c - head) if mode == "channel" else (c - head, head)
        conv = torch.nn.Sequential(torch.nn.Conv2d(cin, 8, 3, padding=1), torch.nn.LeakyReLU(0.0),
                                   torch.nn.Conv2d(8, 2 * cout, 3, padding=1))
        blk = nf.flows.AffineCouplingBlock(conv, scale_map="sigmoid", split_mode=mode).cuda()
        conv64 = torch.nn.Sequential(torch.nn.Conv2d(cin, 8, 3, padding=1), torch.nn.LeakyReLU(0.0),
                                     torch.nn.Conv2d(8, 2 * cout, 3, padding=1)).double()
        conv64.load_state_dict({k: v.detach().cpu().double() for k, v in conv.state_dict().items()})
        ora = OL.AffineCouplingBlock(lambda z: conv64(z), scale_map="sigmoid", split_mode=mode)
        x = torch.randn(5, c, hw, hw)
        with torch.no_grad():
            for dirn in ("forward", "inverse"):
                want_z, want_ld = getattr(ora, dirn)(x.double())
                z, ld = getattr(blk, dirn)(dev(x))
                assert_close(z, want_z, what="%s z" % dirn, rtol=1e-5, atol=1e-5)
                assert_close(ld, want_ld, what="%s ld" % dirn, rtol=1e-5, atol=1e-4)


def test_affine_const_flow_vs_oracle(hip):
    torch.manual_seed(13)
    for shape, zshape in (((6,), (9, 6)), ((5, 1, 1), (4, 5, 3, 3))):
        m = nf.flows.AffineConstFlow(shape).cuda()
        with torch.no_grad():
            m.s.normal_(0, 0.3)
            m.t.normal_(0, 0.3)
        ora = OL.AffineConst(m.s.detach().cpu().double(), m.t.detach().cpu().double())
        x = torch.randn(*zshape)
        with torch.no_grad():
            for dirn in ("forward", "inverse"):
                want_z, want_ld = getattr(ora, dirn)(x.double())
                z, ld = getattr(m, dirn)(dev(x))
                assert_close(z, want_z, what=dirn + " z", rtol=1e-5, atol=1e-5)
                assert_close(ld, want_ld, what=dirn + " ld", rtol=1e-5, atol=1e-5)


def test_plain_kernel_wrappers_refuse_requires_grad(hip):
    """The raw kernel wrappers never drop a gradient silently: outside vcnf_amd.autograd a
    tensor that requires grad raises (the layers route through autograd instead)."""
    x = torch.rand(4, 8, device="cuda")
    uw = torch.randn(4, 8, 5, device="cuda", requires_grad=True)
    with pytest.raises(NotImplementedError):
        _lib.rqs_elementwise(x, uw, uw.detach(), torch.randn(4, 8, 6, device="cuda"), _lib.make_cfg(5, None), False)


# ---------------------------------------------------------------- full-size properties (BASELINE configs)
@FUSED
def test_c3_full_size_round_trip(hip, fused):
    """Config C3 at the benchmark batch (1M x 64, 12 layers, context 16): sample
    then log_prob must reproduce log_q and the base noise (size-independent
    properties; the oracle cannot run this size in seconds)."""
    torch.manual_seed(0)
    model = set_fused(_c3_model().cuda().eval(), fused)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "unnormalized" in n:
                p.normal_(0, 0.5)
    b = 1 << 20
    eps = torch.randn(b, 64, device="cuda")
    ctx = torch.randn(b, 16, device="cuda")
    with torch.no_grad():
        z, lq = model.sample_from(eps, ctx)
        lp = model.log_prob(z, ctx)
        assert torch.isfinite(z).all() and torch.isfinite(lq).all()
        err = (lp - lq).abs() / (1.0 + lq.abs())
        print("C3 1M round trip: log_q rel err max %.3e mean %.3e" % (float(err.max()), float(err.mean())))
        # worst element of 8e8 spline evaluations sits in a floor-derivative bin; the mean is the tight bound
        # base noise recovered by walking the flows backwards
        zz = z
        for f in reversed(model.flows):
            zz, _ = f.inverse(zz, context=ctx)
        rec = (zz - eps).abs()
        print("C3 1M reconstruction: |z0 - eps| max %.3e mean %.3e" % (float(rec.max()), float(rec.mean())))
        # yardstick: the oracle's own fp32 round trip on the first 4096 samples (the oracle cannot run 1M in
        # seconds).  On that subset the build is held to the usual factors; over the full batch the mean to the
        # same factor and the worst of 6.7e7 elements to 64x the oracle's worst of 2.6e5 (an extreme of a
        # 256x larger sample of a heavy-tailed error: ill-conditioned bins amplify rounding by up to 1e3).
        n = 4096
        o_lq, o_rec = oracle_round_trip(oracle_c3_stack({k: v.detach().cpu() for k, v in model.state_dict().items()}),
                                        eps[:n].cpu(), ctx[:n].cpu())
        anchored(err[:n], o_lq, "C3 1M round trip log_q, first 4096")
        anchored(rec[:n], o_rec, "C3 1M reconstruction, first 4096")
        assert float(err.mean()) <= 2.0 * float(o_lq.mean()) + 1e-6 and float(err.max()) <= 64.0 * float(o_lq.max())
        assert float(rec.mean()) <= 2.0 * float(o_rec.mean()) + 1e-6 and float(rec.max()) <= 64.0 * float(o_rec.max())
        # linearity of the log_q accumulation: accumulating into a non-zero buffer adds exactly
        part = model.flows[0].inverse(z[:4096], context=ctx[:4096])[1]
        acc = torch.full((4096,), 2.5, device="cuda")
        model.flows[0].inverse_into(z[:4096], acc, context=ctx[:4096])
        assert_close(acc - 2.5, part.cpu(), rtol=1e-6, atol=1e-5, what="accumulate")
    nf.check_discriminant()


def test_c2_full_size_round_trip(hip):
    torch.manual_seed(1)
    model = _affine_model(8, 32, [16, 64, 64, 32]).cuda().eval()
    b = 262144
    eps = torch.randn(b, 32, device="cuda")
    with torch.no_grad():
        z, lq = model.sample_from(eps)
        lp = model.log_prob(z)
        err = (lp - lq).abs() / (1.0 + lq.abs())
        assert float(err.max()) < 1e-5, float(err.max())


def test_fused_path_is_taken_and_matches_split_path(hip):
    """The single-kernel layer and the three-step path are two implementations of the
    same layer: same outputs to fp32 rounding, and the fused one really is selected
    for the C3 layer shape (with and without context)."""
    from vcnf_amd import fused as fz
    torch.manual_seed(21)
    for ctx_dim in (16, None):
        m = nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, num_context_channels=ctx_dim).cuda().eval()
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "unnormalized" in n:
                    p.normal_(0, 0.5)
                elif "final_layer.weight" in n:
                    p.normal_(0, 1.0 / np.sqrt(128))
        x = torch.randn(1000, 64, device="cuda")
        ctx = torch.randn(1000, 16, device="cuda") if ctx_dim else None
        assert fz.eligible(m.prqct, ctx)
        with torch.no_grad():
            for dirn in ("inverse", "forward"):
                set_fused(m, False)
                zs, ls = getattr(m, dirn)(x, context=ctx)
                for prec in ("fp32", "fp16x3"):
                    set_fused(m, prec)
                    zf, lf = getattr(m, dirn)(x, context=ctx)
                    assert_close(zf, zs.cpu(), rtol=1e-4, atol=1e-4, what="%s %s z" % (prec, dirn))
                    assert_close(lf, ls.cpu(), rtol=1e-4, atol=2e-3, what="%s %s ld" % (prec, dirn))
        # weights changed in place -> the packed copy is rebuilt
        set_fused(m, "fp16x3")
        with torch.no_grad():
            z0, _ = m.inverse(x, context=ctx)
            m.prqct.transform_net.final_layer.bias.add_(0.3)
            z1, _ = m.inverse(x, context=ctx)
        assert not torch.equal(z0, z1)
    # shapes outside the kernel's family fall back to the split path
    other = nf.flows.CoupledRationalQuadraticSpline(20, 2, 64, 8).cuda().eval()
    assert not fz.eligible(other.prqct, None)


def test_fp16x3_saturation_is_counted_and_fp32_restores_parity(hip):
    """Legacy form of the split-half matrix path (``range_safe = False``: one launch, no fp32 pass behind it): inputs,
    context and hidden activations are clamped at +-65504 (the reference is plain fp32, nets/resnet.py:92-106).  The
    kernel counts tiles that clamped; nf.check_saturation() raises, or - given the model - switches the couplings to
    the exact fp32 matrix path, after which the oracle's result is reproduced.  Hidden weights are scaled up until
    the counter trips."""
    torch.manual_seed(22)
    m = set_fused(nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, num_context_channels=16).cuda().eval(), "fp16x3")
    m.prqct.range_safe = False
    x = torch.randn(512, 64, device="cuda")
    ctx = torch.randn(512, 16, device="cuda")
    nf.check_saturation()                                   # clear
    with torch.no_grad():
        m.inverse(x, context=ctx)
    assert nf.check_saturation(model=m) == 0                # ordinary weights: nothing clamped, routing unchanged
    assert m.prqct.fused_precision == "fp16x3"
    tripped = False
    with torch.no_grad():
        for _ in range(12):
            m.prqct.transform_net.initial_layer.weight.mul_(8.0)
            m.prqct.transform_net.initial_layer.bias.mul_(8.0)
            z16, ld16 = m.inverse(x, context=ctx)
            assert torch.isfinite(z16).all() and torch.isfinite(ld16).all()     # saturates, never NaN / inf
            try:
                nf.check_saturation()
            except nf.VcnfError:
                tripped = True
                break
        assert tripped, "hidden activations never left the fp16 range"
        z16, ld16 = m.inverse(x, context=ctx)
        assert nf.check_saturation(model=m) > 0             # counted again; the model is switched to fp32
        assert m.prqct.fused_precision == "fp32"
        z32, ld32 = m.inverse(x, context=ctx)
        assert nf.check_saturation(model=m) == 0
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ora = oracle_rqs_coupling(sd, "prqct.", 8, 3.0, 128)
    with torch.no_grad():
        zo, ldo = ora.nsf_forward(x.cpu(), ctx.cpu())
        zo64, ldo64 = oracle_rqs_coupling({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()},
                                          "prqct.", 8, 3.0, 128).nsf_forward(
            x.cpu().double(), ctx.cpu().double())
    parity(z32, zo, zo64, what="fp32 route z")
    parity(ld32, ldo, ldo64, what="fp32 route log_det")
    # and the clamped run really differs (that is what the counter is for)
    assert float((ld16.cpu() - ldo).abs().max()) > 1e-3


@pytest.mark.parametrize("tile", [32, 128])
@pytest.mark.parametrize("blocks", [1, 2, 3])
@pytest.mark.parametrize("sampling", [False, True], ids=["density", "sampling"])
def test_fp16x3_default_is_range_safe_on_the_device(hip, blocks, sampling, tile):
    """The DEFAULT matrix path (fp16 split-half operands) never clamps: a tile (32 samples in the small-batch kernel,
    128 in the large-batch one) that holds a value the fp16
    halves cannot carry - a hidden activation beyond +-65504, a huge input, a NaN / Inf input - is left unwritten by the
    split-half kernel and evaluated by the exact fp32 kernel in the launch behind it, without a host round trip.
    Asserted: those tiles are BITWISE the exact fp32 path's results (non-finite patterns included), every other tile
    is BITWISE the split-half path's own result on clean inputs, the redo counter says which is which, and both
    log_det modes (store / accumulate into a running log_q) honour the skip."""
    torch.manual_seed(40 + blocks)
    mode = "fp16x3" if tile == 32 else "fp16x3-t128"
    m = set_fused(nf.flows.CoupledRationalQuadraticSpline(64, blocks, 128, 8, num_context_channels=16).cuda().eval(), mode)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "unnormalized_" in n:
                p.normal_(0.0, 0.5)
    B = 128 * 5 + 37                                       # six tiles of 128, the last one ragged
    x = torch.randn(B, 64, device="cuda")
    ctx = torch.randn(B, 16, device="cuda")
    call = (lambda mod, a, c: mod.forward(a, context=c)) if sampling else (lambda mod, a, c: mod.inverse(a, context=c))
    nf.range_redo_count()
    with torch.no_grad():
        z_clean, ld_clean = call(m, x, ctx)
    assert nf.range_redo_count() == 0
    idf, tff = m.prqct.identity_features.tolist(), m.prqct.transform_features.tolist()
    xb, cb = x.clone(), ctx.clone()
    xb[130, idf[3]] = 3.0e5               # tile 1: an identity feature (conditioner input) beyond the fp16 range
    cb[300, 5] = float("inf")             # tile 2: non-finite context
    xb[128 * 4 + 7, idf[0]] = float("nan")    # tile 4: NaN conditioner input
    xb[10, tff[2]] = 1.0e6                # tile 0: a TRANSFORMED feature may be anything (it never enters a GEMM)
    bad_tiles = [130 // tile, 300 // tile, (128 * 4 + 7) // tile]
    with torch.no_grad():
        z, ld = call(m, xb, cb)
        assert nf.range_redo_count() == len(bad_tiles)
        m32 = set_fused(m, "fp32")
        z32, ld32 = call(m32, xb, cb)
        set_fused(m, mode)
        # accumulate mode: the running log_q of NormalizingFlow.log_prob / sample
        logq = torch.full((B,), 0.25, device="cuda")
        z_acc = m.forward_into(xb, logq, context=cb) if sampling else m.inverse_into(xb, logq, context=cb)
    torch.cuda.synchronize()
    rows = torch.arange(B, device="cuda") // tile
    redo = torch.isin(rows, torch.tensor(bad_tiles, device="cuda"))
    eq = lambda a, b: torch.equal(a, b) or torch.equal(torch.nan_to_num(a, nan=1.25e30), torch.nan_to_num(b, nan=1.25e30))
    assert eq(z[redo], z32[redo]) and eq(ld[redo], ld32[redo]), "flagged tiles must carry the exact fp32 kernel's results"
    keep = ~redo
    keep[10] = False                       # row 10's own transformed input differs from the clean run
    assert torch.equal(z[keep], z_clean[keep]) and torch.equal(ld[keep], ld_clean[keep])
    assert torch.equal(z[10, idf], z_clean[10, idf])
    assert not torch.isfinite(z[300]).all() and not torch.isfinite(z[128 * 4 + 7]).all()      # the reference propagates them
    sign = -1.0 if sampling else 1.0
    assert eq(z_acc, z)
    assert eq(logq, 0.25 + sign * ld)
    nf.range_redo_count()
    _lib.bad_discriminant_counter("cuda").zero_()           # the NaN rows tripped the sampling direction's discriminant check


@pytest.mark.parametrize("blocks", [1, 2, 3])
@pytest.mark.parametrize("d,ctx_dim", [(64, 16), (64, 0), (32, 16), (32, 0)])
def test_small_batch_kernel_matches_large_batch_kernel(hip, d, ctx_dim, blocks):
    """csrc/fused_layer_v6s.hip (32-sample tiles, batches up to 16384) against csrc/fused_layer_v6.hip (128-sample
    tiles): same packed weights, same matrix instructions in the same order - z must be BITWISE equal, log_det equal
    up to the order of its per-sample sum; both directions, store and accumulate modes, ragged batches (1 row, a
    partial last tile, more tiles than resident workgroups), with and without the unconditional identity spline."""
    from vcnf_amd import fused as fz
    torch.manual_seed(300 + d + ctx_dim + blocks)
    m = nf.flows.CoupledRationalQuadraticSpline(d, blocks, 128, 8, num_context_channels=ctx_dim or None).cuda().eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "unnormalized_" in n:
                p.normal_(0.0, 0.5)
    assert fz.eligible(m.prqct, torch.zeros(1, ctx_dim, device="cuda") if ctx_dim else None)
    for B in (1, 31, 32, 33, 2048 + 17, 32 * 256 + 45):
        x = torch.randn(B, d, device="cuda") * 1.5
        ctx = torch.randn(B, ctx_dim, device="cuda") if ctx_dim else None
        kw = {"context": ctx} if ctx_dim else {}
        out = {}
        for mode in ("fp16x3", "fp16x3-t128"):
            set_fused(m, mode)
            with torch.no_grad():
                zf, ldf = m.forward(x, **kw)
                zi, ldi = m.inverse(x, **kw)
                lq = torch.full((B,), -0.5, device="cuda")
                za = m.inverse_into(x, lq, **kw)
            out[mode] = (zf, ldf, zi, ldi, za, lq)
        a, b = out["fp16x3"], out["fp16x3-t128"]
        assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[4], b[4]), (B, "z differs")
        for i in (1, 3, 5):
            tol = 4e-6 * (1.0 + b[i].abs())
            assert bool(((a[i] - b[i]).abs() <= tol).all()), (B, i, float((a[i] - b[i]).abs().max()))
    nf.check_discriminant()


@pytest.mark.parametrize("precision", ["fp16x3", "fp32"])
@pytest.mark.parametrize("layers,d,ctx_dim,blocks", [(12, 64, 16, 2), (5, 32, 0, 1), (16, 64, 16, 3), (19, 64, 16, 2)])
def test_rqs_stack_single_launch_matches_per_layer(hip, layers, d, ctx_dim, blocks, precision):
    """NormalizingFlow evaluates a run of one-kernel RQS layers at a small batch in ONE launch
    (vcnf_rqs_stack_fused_f32: the tile stays in LDS from the first layer to the last; 19 layers = a run of 16 and a
    run of 3).  Against one launch per layer (``fuse_rqs_stacks = False``): samples BITWISE equal, log-densities equal
    up to the rounding of one sum over the layers instead of one add per layer; both directions, both matrix paths,
    ragged batches."""
    torch.manual_seed(700 + layers + d)
    model = set_fused(_c3_model(layers=layers, d=d, c=ctx_dim or None, blocks=blocks).cuda().eval(), precision)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "unnormalized_" in n:
                p.normal_(0.0, 0.5)
    from vcnf_amd import fused as fz
    for B in (1, 33, 1024 + 7):
        x, eps = torch.randn(B, d, device="cuda") * 1.3, torch.randn(B, d, device="cuda")
        ctx = torch.randn(B, ctx_dim, device="cuda") if ctx_dim else None
        kw = {"context": ctx} if ctx_dim else {}
        with torch.no_grad():
            assert fz.plan_stack(list(model.flows), 0, x, ctx) is not None
        out = {}
        for stacks in (True, False):
            model.fuse_rqs_stacks = stacks
            with torch.no_grad():
                lp = model.log_prob(x, **kw)
                z, lq = model.sample_from(eps, **kw)
                # the density direction's latents: the run itself, outside log_prob
                zi, lqi = x, torch.zeros(B, device="cuda")
                if stacks:
                    order, i = list(reversed(model.flows)), 0
                    while i < len(order):
                        end, run, sig = fz.plan_stack(order, i, zi, ctx) or (i + 1, None, None)
                        zi = (fz.run_stack(run, sig, zi, ctx, False, lqi, 1.0)[0] if run else order[i].inverse_into(zi, lqi, **kw))
                        i = end
                else:
                    for f in reversed(model.flows):
                        zi = f.inverse_into(zi, lqi, **kw)
            out[stacks] = (lp, z, lq, zi, lqi)
        model.fuse_rqs_stacks = True
        a, b = out[True], out[False]
        assert torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]), (B, "latents differ")
        for i in (0, 2, 4):
            # the running log-density passes through magnitudes of ~1e2 (base density, D = 64): one rounding per layer
            # there (ulp 7.6e-6) against one for the whole run
            assert bool(((a[i] - b[i]).abs() <= 1e-4).all()), (B, i, float((a[i] - b[i]).abs().max()))
    nf.check_discriminant()


@pytest.mark.parametrize("sampling", [False, True], ids=["density", "sampling"])
def test_rqs_stack_is_range_safe_on_the_device(hip, sampling):
    """Range safety of the single-launch run on the split-half path: a 32-sample tile in which ANY layer met a value the
    fp16 halves cannot carry (here: a huge identity-feature input, a non-finite context entry, a NaN input) is left
    unwritten and re-evaluated through ALL layers by the exact fp32 kernel in the launch behind - those rows are BITWISE
    the fp32 model's latents (log-density: to the order of the sum), every other row BITWISE the clean run's, the redo
    counter counts the flagged tiles."""
    torch.manual_seed(77)
    model = set_fused(_c3_model(layers=6).cuda().eval(), "fp16x3")
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "unnormalized_" in n:
                p.normal_(0.0, 0.5)
    B = 32 * 9 + 5
    x, ctx = torch.randn(B, 64, device="cuda"), torch.randn(B, 16, device="cuda")

    def run(m, xa, ca):
        with torch.no_grad():
            if sampling:
                return m.sample_from(xa, context=ca)
            lq = torch.zeros(B, device="cuda")
            from vcnf_amd import fused as fz
            order = list(reversed(m.flows))
            plan = fz.plan_stack(order, 0, xa, ca)
            assert plan is not None and plan[0] == len(order)
            return fz.run_stack(plan[1], plan[2], xa, ca, False, lq, 1.0)[0], lq
    nf.range_redo_count()
    z_clean, lq_clean = run(model, x, ctx)
    assert nf.range_redo_count() == 0
    first = (model.flows[0] if sampling else model.flows[-1]).prqct       # the layer applied first
    idf = first.identity_features.tolist()
    xb, cb = x.clone(), ctx.clone()
    xb[40, idf[3]] = 3.0e5
    cb[100, 5] = float("inf")
    xb[32 * 9 + 2, idf[0]] = float("nan")
    bad_tiles = [40 // 32, 100 // 32, 9]
    z, lq = run(model, xb, cb)
    assert nf.range_redo_count() == len(bad_tiles)
    z32, lq32 = run(set_fused(model, "fp32"), xb, cb)
    set_fused(model, "fp16x3")
    torch.cuda.synchronize()
    rows = torch.arange(B, device="cuda") // 32
    redo = torch.isin(rows, torch.tensor(bad_tiles, device="cuda"))
    eq = lambda a, b: torch.equal(torch.nan_to_num(a, nan=1.25e30), torch.nan_to_num(b, nan=1.25e30))
    assert eq(z[redo], z32[redo]) and eq(lq[redo], lq32[redo]), "flagged tiles must carry the exact fp32 kernel's results"
    assert torch.equal(z[~redo], z_clean[~redo]) and torch.equal(lq[~redo], lq_clean[~redo])
    nf.range_redo_count()
    _lib.bad_discriminant_counter("cuda").zero_()


def _realnvp_like_reference_drivers(d, h, pairs, dtype, nets="st", leaky=0.0):
    """K x [MaskedAffineFlow(b | 1 - b, t, s), ActNorm] with s, t = MLP([d, h, d]) - /root/reference/run.py:58-68."""
    b = torch.tensor([1.0 if i % 2 == 0 else 0.0 for i in range(d)])
    flows = []
    for i in range(pairs):
        s = nf.nets.MLP([d, h, d], leaky=leaky, init_zeros=True) if "s" in nets else None
        t = nf.nets.MLP([d, h, d], leaky=leaky, init_zeros=True) if "t" in nets else None
        flows += [nf.flows.MaskedAffineFlow(b if i % 2 == 0 else 1 - b, t, s), nf.flows.ActNorm(d)]
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".net.2." in n:                        # the zero-initialised last layers: make the couplings do something
                p.normal_(0.0, 0.1)
    return model.to(dtype).cuda().eval()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["fp32", "fp64"])
@pytest.mark.parametrize("d,h,nets,leaky", [(2, 16, "st", 0.0), (15, 30, "st", 0.0), (2, 4, "s", 0.1), (6, 64, "t", 0.0), (16, 33, "st", 0.2)])
def test_masked_affine_stack_single_launch_matches_per_layer(hip, d, h, nets, leaky, dtype):
    """The models of the reference's own drivers - K x [MaskedAffineFlow with MLP conditioners, ActNorm], fp32 and
    .double() - evaluated by NormalizingFlow in ONE launch (csrc/masked_affine_stack.hip) against the per-layer path
    (torch GEMMs + vcnf_masked_affine + vcnf_affine_const, itself pinned to the reference by fixtures G7 / G9):
    log_prob, samples and their log-densities agree to rounding; ragged batches; a non-finite input gives the same NaN
    pattern; an ActNorm that has not seen its first batch ends the run (and is initialised by the per-layer path)."""
    from vcnf_amd import fused_masked
    torch.manual_seed(10 * d + h)
    model = _realnvp_like_reference_drivers(d, h, 6, dtype, nets, leaky)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1e-11, atol=1e-11)
    x0 = torch.randn(512, d, device="cuda", dtype=dtype)
    with torch.no_grad():
        assert fused_masked.plan(list(reversed(model.flows)), 0, x0) is None      # ActNorm not initialised: no run yet
        model.log_prob(x0)                                                        # first batch: data-dependent init
        for f in model.flows:
            if isinstance(f, nf.flows.ActNorm):
                f.s.add_(0.05 * torch.randn_like(f.s))
                f.t.add_(0.05 * torch.randn_like(f.t))
        plan = fused_masked.plan(list(reversed(model.flows)), 0, x0)
        assert plan is not None and plan[0] == len(model.flows)
    for B in (1, 17, 1024 + 3):
        x, eps = torch.randn(B, d, device="cuda", dtype=dtype), torch.randn(B, d, device="cuda", dtype=dtype)
        out = {}
        for stacks in (True, False):
            model.fuse_masked_stacks = stacks
            with torch.no_grad():
                out[stacks] = (model.log_prob(x),) + tuple(model.sample_from(eps))
        model.fuse_masked_stacks = True
        for a, b in zip(out[True], out[False]):
            assert torch.isfinite(b).all() and torch.allclose(a, b, **tol), (B, float((a - b).abs().max()))
    xb = torch.randn(40, d, device="cuda", dtype=dtype)
    xb[7, 0] = float("inf")
    res = []
    for stacks in (True, False):
        model.fuse_masked_stacks = stacks
        with torch.no_grad():
            res.append(model.log_prob(xb))
    model.fuse_masked_stacks = True
    assert torch.equal(torch.isnan(res[0]), torch.isnan(res[1])) and torch.equal(torch.isinf(res[0]), torch.isinf(res[1]))
    ok = torch.isfinite(res[1])
    assert torch.allclose(res[0][ok], res[1][ok], **tol)


def test_graphed_flow_over_masked_affine_stack_and_actnorm(hip):
    """The reference drivers' model family under nf.GraphedFlow: log_prob / sample_from captured into a HIP graph (the
    whole run is one kernel node; ActNorm's initialised-flag is read on the host once, so the capture does not hit the
    per-call device test of normalization.py:29 / :40) replay the eager results bit for bit in fp64, also after an
    in-place parameter update (the kernel reads the parameters in place)."""
    torch.manual_seed(12)
    model = _realnvp_like_reference_drivers(2, 16, 8, torch.float64)
    B = 1024
    x0 = torch.randn(B, 2, device="cuda", dtype=torch.float64)
    with torch.no_grad():
        model.log_prob(x0)                                   # data-dependent ActNorm initialisation (eager, first batch)
    g = nf.GraphedFlow(model, batch=B)
    for trial in range(3):
        x, eps = torch.randn(B, 2, device="cuda", dtype=torch.float64), torch.randn(B, 2, device="cuda", dtype=torch.float64)
        if trial == 2:
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.01 * torch.randn_like(p))
        with torch.no_grad():
            want_lp = model.log_prob(x)
            want_z, want_lq = model.sample_from(eps)
        lp = g.log_prob(x).clone()
        z, lq = g.sample_from(eps)
        assert torch.equal(lp, want_lp) and torch.equal(z, want_z) and torch.equal(lq, want_lq), trial


def test_data_mutation_needs_refresh_packed(hip):
    """the packed weight caches key on (data_ptr, _version); ``p.data`` edits do not bump _version.
    refresh_packed() (also run by train() / eval() / load_state_dict()) makes the fused kernel see them."""
    torch.manual_seed(5)
    model = set_fused(_c3_model(layers=2).cuda().eval(), "fp16x3")
    x, ctx = torch.randn(300, 64, device="cuda"), torch.randn(300, 16, device="cuda")
    with torch.no_grad():
        lp0 = model.log_prob(x, ctx)
        for p in model.flows[0].prqct.transform_net.final_layer.parameters():
            p.data.add_(0.25)                               # behind autograd's back
        stale = model.log_prob(x, ctx)
        assert torch.equal(stale, lp0)                      # the documented trap
        model.refresh_packed()
        lp1 = model.log_prob(x, ctx)
        split = set_fused(model, False).log_prob(x, ctx)
    assert not torch.equal(lp1, lp0)
    assert_close(lp1, split.cpu(), rtol=1e-5, atol=2e-4, what="fused after refresh vs split")
    # eval() / train() transitions refresh as well
    set_fused(model, "fp16x3")
    with torch.no_grad():
        for p in model.flows[1].prqct.transform_net.final_layer.parameters():
            p.data.add_(0.25)
        model.train(); model.eval()
        lp2 = model.log_prob(x, ctx)
        split2 = set_fused(model, False).log_prob(x, ctx)
    assert_close(lp2, split2.cpu(), rtol=1e-5, atol=2e-4, what="fused after train()/eval() vs split")


def test_g13_c5_layer_shape(hip):
    """Config C5's layer shape against the reference: D=1024, K=16 (P=47), conditioner
    512 -> 24064; the spline kernel streams each sample's 96 KB of params in feature chunks
    and the unconditional spline reads its logits from L2 instead of LDS tables."""
    fx = fixture("g13_c5_shape")
    sd, _ = state_for(fx, "c5", 1301, final_gain=1.0)
    flows = [nf.flows.CoupledRationalQuadraticSpline(1024, 2, 128, 16, reverse_mask=bool(i % 2)) for i in range(2)]
    model = load(nf.NormalizingFlow(nf.distributions.DiagGaussian(1024), flows), sd)
    with torch.no_grad():
        lp = model.log_prob(dev(T(fx["x"])))
        assert_close(lp, fx["c5/lp32"], what="log_prob (1e-5 rel)", **LP)
        parity(lp, fx["c5/lp32"], fx["c5/lp64"], rtol=1e-5, atol=2e-4, what="log_prob")
        z, lq = model.sample_from(dev(T(fx["eps"])))
        parity(z, fx["c5/s_z32"], fx["c5/s_z64"], what="sample z")
        parity(lq, fx["c5/s_logq32"], fx["c5/s_logq64"], rtol=1e-5, atol=2e-4, what="sample log_q")
    nf.check_discriminant()


def _glow_model(levels=2, blocks=2, hidden=16, input_shape=(3, 8, 8)):
    q0, merges, flows, L = [], [], [], levels
    for i in range(L):
        fl = [nf.flows.GlowBlock(input_shape[0] * 2 ** (L + 1 - i), hidden, split_mode="channel", scale=True)
              for _ in range(blocks)]
        fl += [nf.flows.Squeeze()]
        flows += [fl]
        if i > 0:
            merges += [nf.flows.Merge()]
            shape = (input_shape[0] * 2 ** (L - i), input_shape[1] // 2 ** (L - i), input_shape[2] // 2 ** (L - i))
        else:
            shape = (input_shape[0] * 2 ** (L + 1), input_shape[1] // 2 ** L, input_shape[2] // 2 ** L)
        q0 += [nf.distributions.DiagGaussian(shape)]
    return nf.MultiscaleFlow(q0, flows, merges, class_cond=False)


def test_g11_glow_multiscale(hip):
    """Config C4's family end to end against the reference: MultiscaleFlow.log_prob / sample
    through GlowBlocks whose affine coupling (4-D, sigmoid scale map, conv conditioner) and
    ActNorm run on the HIP kernels; 1x1 convolution and conditioner convs on PyTorch-ROCm."""
    fx = fixture("g11_glow_multiscale")
    model = load(_glow_model(), glow_state(fx, 1101))
    with torch.no_grad():
        lp = model.log_prob(dev(T(fx["x"])))
        parity(lp, fx["glow/lp32"], fx["glow/lp64"], rtol=1e-5, atol=1e-3, what="log_prob")
        z, lq = model.sample_from([dev(T(fx["eps0"])), dev(T(fx["eps1"]))])
        parity(z, fx["glow/s_z32"], fx["glow/s_z64"], rtol=1e-4, atol=1e-4, what="sample z")
        parity(lq, fx["glow/s_logq32"], fx["glow/s_logq64"], rtol=1e-5, atol=1e-3, what="sample log_q")
        z2, lq2 = model.sample(5)
        assert z2.shape == (5, 3, 8, 8) and lq2.shape == (5,)


def test_g20_c4_real_shape(hip):
    """Config C4 at its REAL shape against the reference (fixture G20; example/glow.ipynb cell 2): 3 x 32 x 32
    inputs, L = 3 levels, K = 16 GlowBlocks per level, 256 hidden channels - 48 affine couplings (4-D, sigmoid
    scale map) + ActNorms on the HIP kernels, 1x1 convolutions and conv conditioners on PyTorch-ROCm."""
    fx = fixture("g20_c4_real_shape")
    model = load(_glow_model(levels=3, blocks=16, hidden=256, input_shape=(3, 32, 32)), glow_state(fx, 2001, weight_gain=0.1, other_gain=0.02))
    with torch.no_grad():
        lp = model.log_prob(dev(T(fx["x"])))
        parity(lp, fx["glow/lp32"], fx["glow/lp64"], rtol=1e-5, atol=1e-2, what="log_prob")
        z, lq = model.sample_from([dev(T(fx["eps%d" % i])) for i in range(3)])
        parity(z, fx["glow/s_z32"], fx["glow/s_z64"], rtol=1e-4, atol=1e-4, what="sample z")
        parity(lq, fx["glow/s_logq32"], fx["glow/s_logq64"], rtol=1e-5, atol=1e-2, what="sample log_q")
        # a real batch: finite, independent of how the batch is cut, the fixture's rows reproduce their stand-alone
        # result; sample then log_prob reproduces log_q as well as the reference's own fp32 run does on the fixture
        torch.manual_seed(7)
        xb = torch.rand(2048, 3, 32, 32, device="cuda")
        xb[:4] = dev(T(fx["x"]))
        lpb = model.log_prob(xb)
        assert torch.isfinite(lpb).all()
        # (not bitwise: the conditioner convolutions are library calls whose algorithm depends on the batch size)
        assert_close(lpb[:4], lp.cpu(), rtol=2e-6, atol=1e-2, what="fixture rows inside a batch of 2048")
        assert_close(lpb[1024:1100], model.log_prob(xb[1024:1100]).cpu(), rtol=2e-6, atol=1e-2, what="batch cut")
        eps = [torch.randn(2048, *q.loc.shape[1:], device="cuda") for q in model.q0]
        zb, lqb = model.sample_from(eps)
        err = (model.log_prob(zb) - lqb).abs() / (1.0 + lqb.abs())
        ref_rt = np.abs(fx["glow/s_logq32"] - fx["glow/s_logq64"]) / (1.0 + np.abs(fx["glow/s_logq64"]))
        print("C4 real shape, B=2048 round trip log_q rel err: max %.3e mean %.3e (reference fp32-vs-fp64 on the "
              "fixture: max %.3e)" % (float(err.max()), float(err.mean()), float(ref_rt.max())))
        assert torch.isfinite(zb).all() and float(err.max()) <= 64.0 * float(ref_rt.max()) + 1e-5


@pytest.mark.parametrize("c,h,w", [(48, 4, 4), (24, 8, 8), (12, 16, 16), (4, 3, 5), (64, 2, 2), (20, 1, 1)])
def test_channel_mix_kernel(hip, c, h, w):
    """csrc/channel_mix.hip: y[b, o, p] = sum_c M[o, c] x[b, c, p] + v[o] (invertible 1x1 convolution + ActNorm of a
    GlowBlock in one pass, mixing.py:57-128 with normalization.py:8-38) against fp64, beside torch's own fp32 conv2d;
    ragged pixel counts (tiles of 16 pixels cross image boundaries when H W is not a multiple of 16)."""
    g = torch.Generator().manual_seed(c * 100 + h)
    mat = torch.randn(c, c, generator=g).cuda()
    vec = torch.randn(c, generator=g).cuda()
    for b in (1, 3, 257):
        x = torch.randn(b, c, h, w, generator=g).cuda()
        got = _lib.channel_mix(x, mat, vec)
        ref64 = torch.einsum("oc,bchw->bohw", mat.double(), x.double()) + vec.double().view(1, c, 1, 1)
        ref32 = torch.nn.functional.conv2d(x, mat.view(c, c, 1, 1), vec)
        scale = float(ref64.abs().max())
        e_got, e_ref = float((got.double() - ref64).abs().max()), float((ref32.double() - ref64).abs().max())
        assert got.shape == x.shape and e_got <= 2.0 * e_ref + 1e-6 * scale, (c, b, e_got, e_ref)
        if h * w == 1:       # rows of a [B, C] matrix (the LU linear layer): the 16-byte store variant
            got2 = _lib.channel_mix(x.view(b, c), mat, vec)
            assert torch.equal(got2, got.view(b, c))
    with pytest.raises(_lib.VcnfError):
        _lib.channel_mix(torch.randn(2, 6, 2, 2, device="cuda"), torch.eye(6, device="cuda"), torch.zeros(6, device="cuda"))


@pytest.mark.parametrize("c_in,c_out,h,w", [(256, 256, 16, 16), (256, 256, 4, 4), (64, 48, 3, 5), (16, 7, 1, 1), (96, 256, 2, 2)])
def test_conv1x1_fused_kernel(hip, c_in, c_out, h, w):
    """csrc/conv1x1.hip: y = leaky(W leaky(x + b_in) + b_out) in one pass on the fp16 split-half matrix path, against
    fp64 beside torch's own fp32 composition (conv2d + bias + LeakyReLU kernels, nets/cnn.py:20-52); ragged pixel counts
    (64-pixel passes cross image boundaries), partial row blocks (c_out not a multiple of 32), with and without the
    optional biases / activations."""
    from vcnf_amd.nets.cnn import pack_conv1x1
    g = torch.Generator().manual_seed(c_in + c_out + h)
    wgt = (torch.randn(c_out, c_in, generator=g) / c_in ** 0.5).cuda()
    b_in, b_out = torch.randn(c_in, generator=g).cuda(), torch.randn(c_out, generator=g).cuda()
    pack = pack_conv1x1(wgt)
    assert pack.numel() == int(_lib.lib().vcnf_conv1x1_pack_floats(c_in, c_out))
    lrelu = torch.nn.functional.leaky_relu
    for b in (1, 3, 130):
        x = torch.randn(b, c_in, h, w, generator=g).cuda()
        for bi, bo, si, so in ((b_in, b_out, 0.0, 0.0), (b_in, b_out, 0.1, 0.2), (None, None, None, None), (None, b_out, None, 0.0)):
            got = _lib.conv1x1_fused(x, pack, c_out, in_bias=bi, out_bias=bo, in_slope=si, out_slope=so)

            def ref(xx, ww, dt):
                t = xx.to(dt) + (bi.to(dt).view(1, -1, 1, 1) if bi is not None else 0)
                t = lrelu(t, si) if si is not None else t
                t = torch.einsum("oc,bchw->bohw", ww.to(dt), t) + (bo.to(dt).view(1, -1, 1, 1) if bo is not None else 0)
                return lrelu(t, so) if so is not None else t
            r64, r32 = ref(x, wgt, torch.float64), ref(x, wgt, torch.float32)
            scale = float(r64.abs().max())
            e_got, e_ref = float((got.double() - r64).abs().max()), float((r32.double() - r64).abs().max())
            assert got.shape == r64.shape and e_got <= 2.0 * e_ref + 2e-6 * scale, (c_in, c_out, b, e_got, e_ref)
    assert nf.check_saturation() == 0


@pytest.mark.parametrize("c_in,h,w", [(6, 16, 16), (12, 8, 8), (24, 4, 4), (3, 5, 7), (1, 2, 2)])
def test_conv3x3_1x1_fused_kernel(hip, c_in, h, w):
    """csrc/conv3x3_1x1.hip: y = leaky(W2 leaky(conv3x3(x, padding 1) + b1) + b2) in one launch (first two layers of the
    Glow conditioner, nets/cnn.py:20-52) on the fp16 split-half matrix path, against fp64 beside torch's own fp32
    composition; image borders (zero padding), passes crossing image boundaries, ragged batches."""
    import torch.nn.functional as F
    from vcnf_amd.nets.cnn import pack_conv1x1
    g = torch.Generator().manual_seed(31 * c_in + h)
    w1 = (torch.randn(256, c_in, 3, 3, generator=g) / (3.0 * c_in ** 0.5)).cuda()
    w2 = (torch.randn(256, 256, generator=g) / 16).cuda()
    b1, b2 = torch.randn(256, generator=g).cuda(), torch.randn(256, generator=g).cuda()
    k1 = 9 * c_in
    p1 = pack_conv1x1(F.pad(w1.reshape(256, k1), (0, (-k1) % 16)))
    p2 = pack_conv1x1(w2)
    assert p1.numel() == int(_lib.lib().vcnf_conv3x3_1x1_pack_floats(c_in))
    for b in (1, 5, 67):
        x = torch.randn(b, c_in, h, w, generator=g).cuda()
        got = _lib.conv3x3_1x1_fused(x, p1, p2, b1, b2, 0.1, 0.0)

        def ref(dt):
            t = F.leaky_relu(F.conv2d(x.to(dt), w1.to(dt), b1.to(dt), padding=1), 0.1)
            return F.leaky_relu(torch.einsum("oc,bchw->bohw", w2.to(dt), t) + b2.to(dt).view(1, -1, 1, 1), 0.0)
        r64, r32 = ref(torch.float64), ref(torch.float32)
        scale = float(r64.abs().max())
        e_got, e_ref = float((got.double() - r64).abs().max()), float((r32.double() - r64).abs().max())
        assert got.shape == r64.shape and e_got <= 2.0 * e_ref + 2e-6 * scale, (c_in, b, e_got, e_ref)
    assert nf.check_saturation() == 0


@pytest.mark.parametrize("c_in,c_out,h,w", [(6, 12, 16, 16), (12, 24, 8, 8), (24, 48, 4, 4), (3, 5, 5, 7), (2, 56, 3, 3)])
def test_convnet3_fused_taps_and_col2im(hip, c_in, c_out, h, w):
    """Whole Glow conditioner (Conv3x3, LeakyReLU, Conv1x1, LeakyReLU, Conv3x3; nets/cnn.py:20-52) in two launches
    (csrc/conv3x3_1x1.hip: fused kernel up to the last layer's nine tap results, then col2im) against fp64 beside torch's
    own fp32 composition; borders, passes crossing images, ragged batches, partial row blocks of the tap matrix."""
    import torch.nn.functional as F
    from vcnf_amd.nets.cnn import pack_conv1x1
    g = torch.Generator().manual_seed(17 * c_in + c_out)
    w1 = (torch.randn(256, c_in, 3, 3, generator=g) / (3.0 * c_in ** 0.5)).cuda()
    w2 = (torch.randn(256, 256, generator=g) / 16).cuda()
    w3 = (torch.randn(c_out, 256, 3, 3, generator=g) / 48).cuda()
    b1, b2, b3 = (torch.randn(n, generator=g).cuda() for n in (256, 256, c_out))
    k1 = 9 * c_in
    p1 = pack_conv1x1(F.pad(w1.reshape(256, k1), (0, (-k1) % 16)))
    p2 = pack_conv1x1(w2)
    p3 = pack_conv1x1(w3.permute(2, 3, 0, 1).reshape(9 * c_out, 256), row_blocks=(9 * c_out + 31) // 32)
    assert p3.numel() == int(_lib.lib().vcnf_convnet3_w3_pack_floats(c_out))
    for b in (1, 5, 67):
        x = torch.randn(b, c_in, h, w, generator=g).cuda()
        got = _lib.convnet3_fused(x, p1, p2, p3, b1, b2, b3, c_out, 0.1, 0.0)

        def ref(dt):
            t = F.leaky_relu(F.conv2d(x.to(dt), w1.to(dt), b1.to(dt), padding=1), 0.1)
            t = F.leaky_relu(F.conv2d(t, w2.to(dt).view(256, 256, 1, 1), b2.to(dt)), 0.0)
            return F.conv2d(t, w3.to(dt), b3.to(dt), padding=1)
        r64, r32 = ref(torch.float64), ref(torch.float32)
        scale = float(r64.abs().max())
        e_got, e_ref = float((got.double() - r64).abs().max()), float((r32.double() - r64).abs().max())
        assert got.shape == r64.shape and e_got <= 2.0 * e_ref + 2e-6 * scale, (c_in, c_out, b, e_got, e_ref)
    assert nf.check_saturation() == 0


def test_convnet2d_fused_middle_layer(hip):
    """ConvNet2d (Glow conditioner, nets/cnn.py:20-52) with its 1x1 convolution, both LeakyReLUs and two bias adds on
    csrc/conv1x1.hip against the same module evaluated layer by layer, and the .data / refresh_packed contract."""
    torch.manual_seed(9)
    net = nf.nets.ConvNet2d((6, 256, 256, 12), (3, 1, 3), leaky=0.1, init_zeros=False).cuda().eval()
    x = torch.randn(37, 6, 8, 8, device="cuda")
    with torch.no_grad():
        assert net._fusable(x)
        got = net(x)
        net.fused_conv1x1 = False
        ref = net(x)
        ref64 = net.double()(x.double())
        net.float()
        net.fused_conv1x1 = True
        e_got, e_ref = float((got.double() - ref64).abs().max()), float((ref.double() - ref64).abs().max())
        assert e_got <= 2.0 * e_ref + 2e-6 * float(ref64.abs().max()), (e_got, e_ref)
        # (not bitwise below: the library's 3x3 convolutions are not run-to-run deterministic)
        assert float((net(x) - got).abs().max()) <= 1e-5   # re-packed: .double() / .float() moved the parameters
        net.net[2].weight.data.mul_(0.5)
        assert float((net(x) - got).abs().max()) <= 1e-5   # packed copy keyed on (data_ptr, _version): stale
        nf.refresh_packed(net)
        assert float((net(x) - got).abs().max()) > 1e-3
    assert not net._fusable(x.requires_grad_(True)) or not torch.is_grad_enabled()


@pytest.mark.parametrize("use_lu", [True, False])
def test_glow_block_fused_mixers_match_layerwise(hip, use_lu):
    """GlowBlock with the 1x1 convolution and ActNorm composed into one channel map (vcnf_amd/flows/affine/glow.py)
    against the same block evaluated layer by layer (conv2d + ActNorm kernel; glow.py:59-73), both directions, and
    against the fp64 composition; then a parameter update through .data needs refresh_packed."""
    torch.manual_seed(5 + use_lu)
    blk = nf.flows.GlowBlock(24, 32, split_mode="channel", scale=True, use_lu=use_lu).cuda().eval()
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if "param_map" in n:
                p.normal_(0, 0.05)
        x = torch.randn(37, 24, 8, 8, device="cuda")
        blk.fused_mixers = False
        blk(x)                                             # initialises the ActNorm from this batch
        blk.flows[-1].s.add_(0.3 * torch.randn_like(blk.flows[-1].s))
        blk.flows[-1].t.add_(0.3 * torch.randn_like(blk.flows[-1].t))
        ref_f, ref_fl = blk(x)
        ref_i, ref_il = blk.inverse(x)
        blk.fused_mixers = True
        assert blk._mixer_eligible(x)
        got_f, got_fl = blk(x)
        got_i, got_il = blk.inverse(x)
        for got, ref in ((got_f, ref_f), (got_i, ref_i)):
            assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
        assert_close(got_fl, ref_fl.cpu(), rtol=1e-6, atol=1e-4, what="log_det sampling direction")
        assert_close(got_il, ref_il.cpu(), rtol=1e-6, atol=1e-4, what="log_det density direction")
        back, bl = blk.inverse(got_f)
        assert float((back - x).abs().max()) <= 1e-4 and float((bl + got_fl).abs().max()) <= 1e-3
        # cached composition: a .data update is not seen until refresh_packed
        blk.flows[-1].t.data.add_(1.0)
        stale, _ = blk(x)
        assert torch.equal(stale, got_f)
        nf.refresh_packed(blk)
        fresh, _ = blk(x)
        assert float((fresh - got_f - 1.0).abs().max()) <= 1e-4


@pytest.mark.parametrize("d_in,blocks", [(512, 2), (16, 1), (80, 3)])
def test_resnet_trunk_kernel_vs_torch(hip, d_in, blocks):
    """csrc/resnet_trunk.hip (ResidualNet trunk in one launch, exact fp32 matrix instructions) against the module's own
    PyTorch evaluation and the oracle's fp64 (nets/resnet.py:92-105), ragged batches."""
    from vcnf_amd import fused_final
    torch.manual_seed(100 + d_in)
    m = nf.flows.CoupledRationalQuadraticSpline(2 * d_in, blocks, 128, 16).cuda().eval()
    net = m.prqct.transform_net
    with torch.no_grad():
        for p in net.parameters():
            p.normal_(0, 0.2)
    sd64 = {k[len("prqct.transform_net."):]: v.detach().double().cpu() for k, v in m.state_dict().items()
            if k.startswith("prqct.transform_net.")}
    for b in (1, 17, 1024 + 5, 65536 + 21):          # the last one runs two 16-sample tiles per wave
        x = torch.randn(b, d_in, device="cuda")
        with torch.no_grad():
            assert fused_final.trunk_eligible(net, x, None)
            got = _lib.resnet_trunk(x, fused_final.packed_trunk(m.prqct), 128, blocks)
            ref32 = net.hidden(x)
            h64 = x.double().cpu() @ sd64["initial_layer.weight"].t() + sd64["initial_layer.bias"]
            for i in range(blocks):
                t = torch.relu(h64) @ sd64["blocks.%d.linear_layers.0.weight" % i].t() + sd64["blocks.%d.linear_layers.0.bias" % i]
                t = torch.relu(t) @ sd64["blocks.%d.linear_layers.1.weight" % i].t() + sd64["blocks.%d.linear_layers.1.bias" % i]
                h64 = h64 + t
        scale = float(h64.abs().max())
        e_got, e_ref = float((got.double().cpu() - h64).abs().max()), float((ref32.double().cpu() - h64).abs().max())
        assert e_got <= 2.0 * e_ref + 1e-6 * scale, (d_in, b, e_got, e_ref)
        # the split hand-over to the last-layer kernel: per row 128 fp16 hi halves | 128 fp16 lo halves of the same values
        with torch.no_grad():
            hs = _lib.resnet_trunk(x, fused_final.packed_trunk(m.prqct), 128, blocks, split=True).view(torch.float16).view(b, 256)
        c = got.clamp(-65504.0, 65504.0)
        ref_hi = c.half()
        assert torch.equal(hs[:, :128], ref_hi) and torch.equal(hs[:, 128:], ((c - ref_hi.float()) * 2048.0).half())


@pytest.mark.parametrize("k,tails,d,d_id", [(16, "linear", 1024, 512), (8, "linear", 37, 19), (10, "linear", 200, 70),
                                            (4, None, 9, 4), (32, "linear", 130, 65)])
def test_identity_half_kernel_bitwise(hip, k, tails, d, d_id):
    """csrc/rqs_kernels.hip::rqs_identity_half_kernel (gather + batch-shared unconditional spline + conditioner input +
    log-det partial rows in one launch, coupling.py:76-116) against the kernels it replaces on the gathered columns
    (vcnf_rqs_shared_f32: same table build, same bin evaluation -> bitwise), ragged feature chunks and batches, both
    directions, points outside the interval and exactly on its ends; and the plain gather without shared logits."""
    g = torch.Generator().manual_seed(7 * k + d)
    perm = torch.randperm(d, generator=g)[:d_id].sort().values
    idx = perm.to(torch.int32).cuda()
    nd = k - 1 if tails == "linear" else k + 1
    sw, sh, sd_ = (torch.randn(d_id, n, generator=g).cuda() for n in (k, k, nd))
    cfg = _lib.make_cfg(k, tails, tail_bound=3.0) if tails else _lib.make_cfg(k, None)
    for b in (1, 63, 64 * 5 + 3):
        if tails:
            x = (torch.randn(b, d, generator=g) * 2.5).cuda()
            x[0, perm[0]] = 3.0
            x[-1, perm[-1]] = -3.0
        else:
            x = torch.rand(b, d, generator=g).cuda()
        for inverse in (False, True):
            rows = _lib.identity_half_rows(d_id, (sw, sh, sd_))
            assert rows == (d_id + 63) // 64
            partial = torch.full((rows + 1, b), float("nan"), device="cuda")
            out = torch.full_like(x, float("nan"))
            ci = _lib.rqs_identity_half(x, out, idx, d_id, (sw, sh, sd_), cfg, inverse, partial=partial[1:])
            xi = x[:, perm.cuda()].contiguous()
            y_ref, lad_ref = _lib.rqs_elementwise_shared(xi, sw, sh, sd_, cfg, inverse)
            assert torch.equal(out[:, perm.cuda()], y_ref)
            assert torch.equal(ci, y_ref if inverse else xi)
            mask = torch.ones(d, dtype=torch.bool)
            mask[perm] = False
            assert torch.isnan(out[:, mask.cuda()]).all() and torch.isnan(partial[0]).all()   # nothing else is written
            want = lad_ref.double().sum(1)
            got = partial[1:].double().sum(0)
            assert torch.allclose(got, want, rtol=1e-6, atol=1e-6 * d_id)
            # rows are per 64-feature chunk
            for c in range(rows):
                assert torch.allclose(partial[1 + c].double(), lad_ref[:, 64 * c:64 * c + 64].double().sum(1), rtol=1e-6, atol=1e-5)
        out = torch.full_like(x, float("nan"))
        ci = _lib.rqs_identity_half(x, out, idx, d_id, None, None, False)
        assert torch.equal(ci, x[:, perm.cuda()]) and torch.equal(out[:, perm.cuda()], ci)
    assert nf.check_discriminant("cuda") is None


def test_g21_c5_real_depth(hip):
    """Config C5 at its REAL depth against the reference (fixture G21): 24 RQS couplings, D = 1024, K = 16
    (P = 47), conditioner 512 -> 24064; then the per-GPU shard's micro-batching on 32 768 samples: sample,
    log_prob through ShardedEvaluator(micro_batch = 8192) must reproduce log_q and equal the unchunked result."""
    fx = fixture("g21_c5_real_depth")
    sd, _ = state_for(fx, "c5", 2101, final_gain=1.0)
    flows = [nf.flows.CoupledRationalQuadraticSpline(1024, 2, 128, 16, reverse_mask=bool(i % 2)) for i in range(24)]
    model = load(nf.NormalizingFlow(nf.distributions.DiagGaussian(1024), flows), sd)
    with torch.no_grad():
        lp = model.log_prob(dev(T(fx["x"])))
        parity(lp, fx["c5/lp32"], fx["c5/lp64"], rtol=1e-5, atol=2e-3, what="log_prob")
        z, lq = model.sample_from(dev(T(fx["eps"])))
        parity(z, fx["c5/s_z32"], fx["c5/s_z64"], what="sample z")
        parity(lq, fx["c5/s_logq32"], fx["c5/s_logq64"], rtol=1e-5, atol=2e-3, what="sample log_q")
        assert survey71_violators(lp, fx["c5/lp32"], fx["c5/lp64"], "G21 log_prob") < 0.25
        torch.manual_seed(11)
        b = 32768
        eps = torch.randn(b, 1024, device="cuda")
        zb, lqb = model.sample_from(eps)
        ev = nf.ShardedEvaluator(model.log_prob, micro_batch=8192)
        lpb = ev.log_prob_shard(zb)
        assert torch.equal(lpb[:4096], model.log_prob(zb[:4096]))            # chunking changes nothing
        err = (lpb - lqb).abs() / (1.0 + lqb.abs())
        # yardstick: the oracle's own fp32 round trip on the first 32 samples (24 layers of a random-weight model
        # amplify rounding: the reference's round trip is ~3e-3 relative itself)
        sdc = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o_lq, _ = oracle_round_trip(oracle_crqs_stack(sdc, 24, 16, 3.0, 128), eps[:32].cpu(), None)
        anchored(err[:32], o_lq, "C5 real depth round trip log_q, first 32")
        print("C5 real depth, B=32768 round trip log_q rel err: max %.3e mean %.3e" % (float(err.max()), float(err.mean())))
        assert float(err.mean()) <= 2.0 * float(o_lq.mean()) + 1e-6 and float(err.max()) <= 64.0 * float(o_lq.max())
    nf.check_discriminant()


def test_c5_full_shard_one_pass(hip):
    """Config C5 at BASELINE.json's per-GPU size: the 524 288 x 1024 shard in ONE pass (no logits are materialised, so
    it fits: the three launches per layer of vcnf_amd/fused_final.py at their large-batch tile shapes - two tiles per
    wave in the trunk kernel).  Size-independent properties: the pass reproduces, bit for bit, the rows of a
    65 536-sample and of a 4 096-sample evaluation (per-sample arithmetic does not depend on the tile shape), and
    sample -> log_prob reproduces log_q as well as it does at 32 768 samples (test_g21_c5_real_depth)."""
    fx = fixture("g21_c5_real_depth")
    sd, _ = state_for(fx, "c5", 2101, final_gain=1.0)
    flows = [nf.flows.CoupledRationalQuadraticSpline(1024, 2, 128, 16, reverse_mask=bool(i % 2)) for i in range(24)]
    model = load(nf.NormalizingFlow(nf.distributions.DiagGaussian(1024), flows), sd)
    b = 524288
    gen = torch.Generator(device="cuda").manual_seed(12)
    with torch.no_grad():
        eps = torch.randn(b, 1024, device="cuda", generator=gen)
        z, lq = model.sample_from(eps)
        assert torch.isfinite(z).all() and torch.isfinite(lq).all()
        for lo, n in ((0, 4096), (262144 - 7, 65536), (b - 4096, 4096)):
            zs, lqs = model.sample_from(eps[lo:lo + n])
            assert torch.equal(zs, z[lo:lo + n]) and torch.equal(lqs, lq[lo:lo + n]), (lo, n)
        eps32 = eps[:32].cpu()
        del eps
        lp = model.log_prob(z)
        assert torch.equal(model.log_prob(z[1000:1000 + 8192]), lp[1000:1000 + 8192])
        err = (lp - lq).abs() / (1.0 + lq.abs())
        print("C5 full shard (524288 x 1024, one pass) round trip log_q rel err: max %.3e mean %.3e" % (
            float(err.max()), float(err.mean())))
        # yardstick as in test_g21_c5_real_depth: the oracle's own fp32 round trip on the first 32 samples (24 layers of
        # a random-weight model amplify rounding: the reference's round trip is ~3e-3 relative itself); the mean over the
        # shard must stay at that level, the max may reach further into the tail with 16 384x more samples
        sdc = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o_lq, _ = oracle_round_trip(oracle_crqs_stack(sdc, 24, 16, 3.0, 128), eps32, None)
        anchored(err[:32], o_lq, "C5 full shard round trip log_q, first 32")
        assert float(err.mean()) <= 2.0 * float(o_lq.mean()) + 1e-6 and float(err.max()) <= 64.0 * float(o_lq.max())
    nf.check_discriminant()
    assert nf.check_saturation(model=model) == 0


def test_actnorm_data_dependent_init(hip):
    """First batch standardises the output (normalization.py:20-27), then parameters stay."""
    torch.manual_seed(31)
    a = nf.flows.ActNorm((6, 1, 1)).cuda()
    x = 3.0 * torch.randn(64, 6, 5, 5, device="cuda") + 2.0
    with torch.no_grad():
        y, ld = a(x)
        assert float(a.data_dep_init_done) == 1.0
        assert_close(y.mean(dim=(0, 2, 3)), torch.zeros(6), rtol=0, atol=1e-4, what="mean")
        assert_close(y.std(dim=(0, 2, 3)), torch.ones(6), rtol=0, atol=1e-3, what="std")
        s0 = a.s.clone()
        a(x + 1.0)
        assert torch.equal(a.s, s0)
        xr, ldi = a.inverse(y)
        assert_close(xr, x.cpu(), rtol=1e-5, atol=1e-5, what="round trip")
        assert_close(ld + ldi, torch.zeros(()), rtol=0, atol=1e-4, what="log-det cancel")


# ---------------------------------------------------------------- next rows: LULinearPermute (G14), checkerboard (G15)
@pytest.mark.parametrize("d", [5, 64])
def test_g14_lu_linear_permute(hip, d):
    fx = fixture("g14_lu_linear_permute")
    tag = "d%d" % d
    sd, _ = state_for(fx, tag, 1401 + d, weight_gain=0.5)
    lay = load(nf.flows.LULinearPermute(d, identity_init=False), sd)
    x = dev(T(fx[tag + "/x"]))
    with torch.no_grad():
        for dirn, fn in (("fwd", lay.forward), ("inv", lay.inverse)):
            z, ld = fn(x)
            parity(z, fx["%s/%s_z32" % (tag, dirn)], fx["%s/%s_z64" % (tag, dirn)], rtol=1e-5, atol=1e-5, what=dirn + " z")
            assert_close(ld, fx["%s/%s_ld32" % (tag, dirn)], what=dirn + " ld", rtol=1e-6, atol=1e-6)
        # widths the channel-mix kernel covers (multiples of 4 up to 64) take it: one launch per direction
        events = []
        _lib.EVENT_SINK = events
        lay.forward(x)
        _lib.EVENT_SINK = None
        assert len([e for e in events if e[2] == "channel_mix"]) == (1 if d % 4 == 0 else 0)
        # the stand-alone permutation module is the HIP column gather
        p, _ = lay.permutation(x)
        assert torch.equal(p.cpu(), x.cpu()[:, lay.permutation._permutation.cpu()])
        back, _ = lay.permutation.inverse(p)
        assert torch.equal(back, x)


@pytest.mark.parametrize("mode", ["checkerboard", "checkerboard_inv"])
def test_g15_checkerboard_affine_block(hip, mode):
    fx = fixture("g15_checkerboard")
    sd, _ = state_for(fx, "blk/" + mode, 1501)
    blk = load(nf.flows.AffineCouplingBlock(nf.nets.MLP([6, 16, 12]), split_mode=mode), sd)
    x = dev(T(fx["blk/x"]))
    with torch.no_grad():
        for dirn, fn in (("fwd", blk.forward), ("inv", blk.inverse)):
            z, ld = fn(x.clone())
            assert_close(z, fx["blk/%s/%s_z32" % (mode, dirn)], what=dirn + " z", **TOL)
            assert_close(ld, fx["blk/%s/%s_ld32" % (mode, dirn)], what=dirn + " ld", **TOL)
        z4 = dev(T(fx["4d/z"]))
        (a, b), _ = nf.flows.Split(mode).forward(z4)
        assert np.array_equal(a.cpu().numpy(), fx["4d/%s/z1" % mode])
        assert np.array_equal(b.cpu().numpy(), fx["4d/%s/z2" % mode])


# ---------------------------------------------------------------- next row: circular tails (G16)
@pytest.mark.parametrize("case", ["K8_T3", "K5_T2.5"])
@pytest.mark.parametrize("inv", [False, True])
def test_g16_circular_tails_functional(hip, case, inv):
    fx = fixture("g16_circular")
    tb = float(case.split("_T")[1])
    tag = case + ("_inv" if inv else "_fwd")
    args = [dev(T(fx["%s/%s" % (case, n)])) for n in ("x", "uw", "uh", "ud")]
    with torch.no_grad():
        y, ld = nf.utils.splines.unconstrained_rational_quadratic_spline(*args, inverse=inv, tails="circular",
                                                                         tail_bound=tb)
    parity(y, fx[tag + "/y32"], fx[tag + "/y64"], what="y")
    parity(ld, fx[tag + "/ld32"], fx[tag + "/ld64"], what="ld")
    nf.check_discriminant()


def test_g16_circular_coupling_layer(hip):
    fx = fixture("g16_circular")
    sd, _ = state_for(fx, "layer", 1651, final_gain=2.0)
    net = lambda i, o: nf.nets.ResidualNet(in_features=i, out_features=o, hidden_features=32, context_features=4,
                                           num_blocks=1, activation=torch.nn.functional.relu,
                                           dropout_probability=0.0, use_batch_norm=False)
    m = nf.flows.neural_spline.coupling.PiecewiseRationalQuadraticCoupling(
        mask=nf.utils.masks.create_alternating_binary_mask(12, even=False), transform_net_create_fn=net,
        num_bins=6, tails="circular", tail_bound=2.0, apply_unconditional_transform=True)
    m = load(m, sd)
    x, ctx = dev(T(fx["layer/x"])), dev(T(fx["layer/ctx"]))
    with torch.no_grad():
        for dirn, fn in (("nsf_fwd", m.forward), ("nsf_inv", m.inverse)):
            z, ld = fn(x, ctx)
            parity(z, fx["layer/%s_z32" % dirn], fx["layer/%s_z64" % dirn], what=dirn + " z")
            parity(ld, fx["layer/%s_ld32" % dirn], fx["layer/%s_ld64" % dirn], what=dirn + " ld")
    nf.check_discriminant()


# ---------------------------------------------------------------- next row: image-shaped RQS coupling (G17)
def _image_coupling(cc):
    net = lambda i, o: nf.nets.ConvResidualNet(in_channels=i, out_channels=o, hidden_channels=16, context_channels=cc,
                                               num_blocks=1, activation=torch.nn.functional.relu,
                                               dropout_probability=0.0, use_batch_norm=False)
    return nf.flows.neural_spline.coupling.PiecewiseRationalQuadraticCoupling(
        mask=nf.utils.masks.create_alternating_binary_mask(6, even=True), transform_net_create_fn=net,
        num_bins=8, tails="linear", tail_bound=3.0, apply_unconditional_transform=True, img_shape=[4, 4])


@pytest.mark.parametrize("tag", ["noctx", "ctx"])
def test_g17_image_rqs_coupling(hip, tag):
    """[B, C, H, W] inputs, channel mask: the splines read the convolutional conditioner's
    [B, C_t*P, H, W] output in place (strided elementwise kernel), the per-pixel unconditional
    spline reads its [C_id, H, W, K] logits once for the whole batch."""
    fx = fixture("g17_image_rqs")
    sd, _ = state_for(fx, tag, 1701, final_gain=2.0)
    m = load(_image_coupling(2 if tag == "ctx" else None), sd)
    x = dev(T(fx["x"]))
    ctx = dev(T(fx["ctx"])) if tag == "ctx" else None
    with torch.no_grad():
        for dirn, fn in (("nsf_fwd", m.forward), ("nsf_inv", m.inverse)):
            z, ld = fn(x, ctx)
            parity(z, fx["%s/%s_z32" % (tag, dirn)], fx["%s/%s_z64" % (tag, dirn)], what=dirn + " z")
            parity(ld, fx["%s/%s_ld32" % (tag, dirn)], fx["%s/%s_ld64" % (tag, dirn)], rtol=1e-5, atol=1e-4,
                   what=dirn + " ld")
        y, ld1 = m.forward(x, ctx)
        back, ld2 = m.inverse(y, ctx)
        # fp32 round trip through steep bins: tight on average, bounded at the worst element.  The
        # log-det is a sum over C*H*W elements and the convolution algorithm MIOpen picks differs from
        # box to box (1.5e-3 seen once with every parity assertion above green), hence the margin.
        err = (back - x).abs()
        assert float(err.mean()) < 2e-5 and float(err.max()) < 2e-2 and float((ld1 + ld2).abs().mean()) < 5e-3
    nf.check_discriminant()


# ---------------------------------------------------------------- fused affine coupling layer
@pytest.mark.parametrize("d,hidden,mode,sm", [
    (2, 32, "channel", "exp"), (32, 64, "channel", "exp"), (32, 64, "channel_inv", "sigmoid"),
    (33, 32, "channel_inv", "sigmoid_inv"), (75, 128, "channel", "noscale"), (128, 128, "channel", "exp"),
    (9, 64, "channel_inv", "exp")])
@pytest.mark.parametrize("batch", [1, 77, 4096])
def test_fused_affine_layer_vs_three_step_path_and_oracle(hip, d, hidden, mode, sm, batch):
    """One-kernel AffineCouplingBlock (MLP conditioner on the fp32 matrix instruction) against the
    three-step path (torch GEMMs + affine kernel) and against the oracle in fp64."""
    from vcnf_amd import fused_affine
    torch.manual_seed(d * 7 + hidden)
    head = d - d // 2
    cin, cout = (head, d - head) if mode == "channel" else (d - head, head)
    scale = sm != "noscale"
    blk = nf.flows.AffineCouplingBlock(nf.nets.MLP([cin, hidden, hidden, (2 if scale else 1) * cout], leaky=0.1),
                                       scale=scale, scale_map=sm if scale else "exp", split_mode=mode)
    with torch.no_grad():
        blk.flows[1].param_map.net[4].weight.mul_(0.5)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    blk = blk.cuda()
    x = torch.randn(batch, d)
    assert fused_affine.eligible(blk, dev(x))
    sd64 = {k: v.double() for k, v in sd.items()}
    ora = OL.AffineCouplingBlock(lambda z: ON.mlp(sd64, "flows.1.param_map.", z, 0.1), scale=scale,
                                 scale_map=sm if scale else "exp", split_mode=mode)
    with torch.no_grad():
        for dirn in ("forward", "inverse"):
            blk.fused = True
            z1, ld1 = getattr(blk, dirn)(dev(x))
            blk.fused = False
            z2, ld2 = getattr(blk, dirn)(dev(x))
            want_z, want_ld = getattr(ora, dirn)(x.double())
            assert_close(z1, want_z, what="fused z " + dirn, rtol=2e-5, atol=2e-5)
            assert_close(z2, want_z, what="three-step z " + dirn, rtol=2e-5, atol=2e-5)
            if scale:
                assert_close(ld1, want_ld, what="fused ld " + dirn, rtol=2e-5, atol=2e-5 * max(1, cout))
            else:
                assert not ld1.any()
        # accumulate-into form used by NormalizingFlow
        blk.fused = True
        lq = torch.full((batch,), 0.25, device="cuda")
        z3 = blk.inverse_into(dev(x), lq)
        zi, ldi = blk.inverse(dev(x))
        assert torch.equal(z3, zi) and torch.allclose(lq, 0.25 + ldi, rtol=1e-6, atol=1e-6)


# ---------------------------------------------------------------- HIP-graph capture of the evaluation loops
@pytest.mark.parametrize("kind", ["c3", "c1"])
def test_graphed_flow_replays_match_eager(hip, kind):
    """log_prob / sample_from captured into a HIP graph (one launch per call) give the eager
    results bit for bit, for fresh inputs and after an in-place weight update."""
    torch.manual_seed(3)
    if kind == "c3":
        model, ctx_dim, d, b = _c3_model(layers=4).cuda(), 16, 64, 1000
    else:
        model, ctx_dim, d, b = _affine_model(4, 2, [1, 32, 32, 2]).cuda(), None, 2, 4096
    g = nf.GraphedFlow(model, batch=b, context_features=ctx_dim)
    for trial in range(3):
        x, eps = torch.randn(b, d, device="cuda"), torch.randn(b, d, device="cuda")
        ctx = torch.randn(b, ctx_dim, device="cuda") if ctx_dim else None
        kw = {"context": ctx} if ctx_dim else {}
        if trial == 2:
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(0.01 * torch.randn_like(p))
            g.refresh()
        with torch.no_grad():
            want_lp = model.log_prob(x, **kw)
            want_z, want_lq = model.sample_from(eps, **kw)
        lp = g.log_prob(x, ctx).clone()
        z, lq = g.sample_from(eps, ctx)
        assert torch.equal(lp, want_lp) and torch.equal(z, want_z) and torch.equal(lq, want_lq), trial
    nf.check_discriminant()


# ---------------------------------------------------------------- fused RQS layer, d_id = d_t = 16 family
@pytest.mark.parametrize("d,blocks,ctx_dim", [(64, 1, 16), (64, 3, 16), (32, 3, 0), (64, 1, 0)])
def test_fused_rqs_layer_one_and_three_blocks(hip, d, blocks, ctx_dim):
    """The one-kernel layer with 1 and 3 residual blocks, on both matrix paths (round 3: the exact fp32 kernel covers
    them too - it is the range fallback of the split-half kernel)."""
    from vcnf_amd import fused
    torch.manual_seed(7 * d + blocks)
    lay = nf.flows.CoupledRationalQuadraticSpline(d, blocks, 128, 8, num_context_channels=ctx_dim or None)
    with torch.no_grad():
        for n, p in lay.named_parameters():
            if "final_layer" in n or "unconditional" in n:
                p.normal_(0, 0.5)
    sd = {k: v.detach().clone() for k, v in lay.state_dict().items()}
    lay = lay.cuda()
    b = 515
    x = 1.5 * torch.randn(b, d)
    ctx = torch.randn(b, 16) if ctx_dim else None
    cg = dev(ctx) if ctx_dim else None
    assert fused.eligible(lay.prqct, cg)
    o32 = oracle_rqs_coupling(sd, "prqct.", 8, 3.0, 128)
    o64 = oracle_rqs_coupling({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, "prqct.", 8, 3.0, 128)
    for prec in ("fp32", "fp16x3"):
        lay.prqct.fused_precision = prec
        assert fused.eligible(lay.prqct, cg)
        with torch.no_grad():
            for dirn in ("forward", "inverse"):
                z, ld = getattr(lay, dirn)(dev(x), cg)
                w32 = getattr(o32, dirn)(x, ctx)
                w64 = getattr(o64, dirn)(x.double(), ctx.double() if ctx_dim else None)
                parity(z, w32[0], w64[0], what="%s z %s" % (prec, dirn))
                parity(ld, w32[1], w64[1], rtol=1e-5, atol=2e-5, what="%s ld %s" % (prec, dirn))
    nf.check_discriminant()


@pytest.mark.parametrize("ctx_dim", [0, 16])
@pytest.mark.parametrize("precision", ["fp16x3", "fp32"])
def test_fused_rqs_layer_d32_family(hip, ctx_dim, precision):
    """D = 32 (16 identity + 16 transformed features, H = 128, 2 blocks, 8 bins) takes the fused
    kernels too; compared with the three-step path and with the oracle (fp32 and fp64)."""
    torch.manual_seed(41 + ctx_dim)
    lay = nf.flows.CoupledRationalQuadraticSpline(32, 2, 128, 8, reverse_mask=True,
                                                  num_context_channels=ctx_dim or None)
    with torch.no_grad():
        for n, p in lay.named_parameters():
            if "final_layer" in n:
                p.normal_(0, 0.5)
            if "unconditional" in n:
                p.normal_(0, 0.5)
    sd = {k: v.detach().clone() for k, v in lay.state_dict().items()}
    lay = lay.cuda()
    lay.prqct.fused_precision = precision
    b = 777
    x = 1.5 * torch.randn(b, 32)
    ctx = torch.randn(b, 16) if ctx_dim else None
    from vcnf_amd import fused
    assert fused.eligible(lay.prqct, dev(ctx) if ctx_dim else None)
    o32 = oracle_rqs_coupling(sd, "prqct.", 8, 3.0, 128)
    o64 = oracle_rqs_coupling({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, "prqct.", 8, 3.0, 128)
    with torch.no_grad():
        for dirn in ("forward", "inverse"):
            lay.prqct.fused = True
            z, ld = getattr(lay, dirn)(dev(x), dev(ctx) if ctx_dim else None)
            lay.prqct.fused = False
            z2, ld2 = getattr(lay, dirn)(dev(x), dev(ctx) if ctx_dim else None)
            w32 = getattr(o32, dirn)(x, ctx)
            w64 = getattr(o64, dirn)(x.double(), ctx.double() if ctx_dim else None)
            parity(z, w32[0], w64[0], what="fused z " + dirn)
            parity(ld, w32[1], w64[1], rtol=1e-5, atol=2e-5, what="fused ld " + dirn)
            parity(z2, w32[0], w64[0], what="split z " + dirn)
    nf.check_discriminant()


def test_permute_folded_into_fused_affine_layer(hip):
    """NormalizingFlow folds a Permute next to a one-kernel affine layer into that kernel's load /
    store index; result identical to running the two layers separately (shuffle and swap)."""
    torch.manual_seed(8)
    d = 12
    for mode in ("shuffle", "swap"):
        flows = []
        for _ in range(3):
            flows += [nf.flows.AffineCouplingBlock(nf.nets.MLP([6, 32, 32, 12])), nf.flows.Permute(d, mode=mode)]
        with torch.no_grad():
            for f in flows[::2]:
                f.flows[1].param_map.net[4].weight.normal_(0, 0.2)
        model = nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows).cuda()
        x, eps = torch.randn(333, d, device="cuda"), torch.randn(333, d, device="cuda")
        with torch.no_grad():
            lp, (z, lq) = model.log_prob(x), model.sample_from(eps)
            zz, lq2 = model.q0.from_noise(eps)                 # the same stack, layer by layer
            for f in model.flows:
                zz, ld = f(zz)
                lq2 = lq2 - ld
            yy, lp2 = x, torch.zeros(333, device="cuda")
            for f in reversed(model.flows):
                yy, ld = f.inverse(yy)
                lp2 = lp2 + ld
            lp2 = lp2 + model.q0.log_prob(yy)
        assert torch.allclose(z, zz, rtol=1e-6, atol=1e-6) and torch.allclose(lq, lq2, rtol=1e-6, atol=1e-5)
        assert torch.allclose(lp, lp2, rtol=1e-6, atol=1e-5)


# ---------------------------------------------------------------- last layer + splines in one kernel (any d_t)
@pytest.mark.parametrize("d,k,blocks,ctx_dim,batch", [
    (48, 8, 3, 16, 1000),        # d_id = d_t = 24: outside the one-kernel families
    (1024, 16, 2, 0, 300),       # config C5's layer shape
    (42, 8, 1, 0, 77),           # d_t = 21: last feature group partly empty
    (10, 16, 2, 5, 4097),
    (40, 10, 2, 0, 513)])        # the reference's default bin count (coupling.py:250-258)
def test_final_layer_fused_with_splines(hip, d, k, blocks, ctx_dim, batch):
    """Conditioner trunk on PyTorch-ROCm, last Linear + splines in csrc/fused_final.hip (fp16 split-half
    matrix path, logits never materialised) against the three-step path and the oracle."""
    from vcnf_amd import fused, fused_final
    torch.manual_seed(d + k)
    lay = nf.flows.CoupledRationalQuadraticSpline(d, blocks, 128, k, reverse_mask=bool(d % 3),
                                                  num_context_channels=ctx_dim or None)
    with torch.no_grad():
        for n, p in lay.named_parameters():
            if "final_layer" in n or "unconditional" in n:
                p.normal_(0, 0.4)
    sd = {n: v.detach().clone() for n, v in lay.state_dict().items()}
    lay = lay.cuda()
    x = 1.5 * torch.randn(batch, d)
    ctx = torch.randn(batch, ctx_dim) if ctx_dim else None
    cg = dev(ctx) if ctx_dim else None
    assert not fused.eligible(lay.prqct, cg) and fused_final.eligible(lay.prqct, dev(x), cg)
    o32 = oracle_rqs_coupling(sd, "prqct.", k, 3.0, 128)
    o64 = oracle_rqs_coupling({n: (v.double() if v.is_floating_point() else v) for n, v in sd.items()}, "prqct.", k, 3.0, 128)
    with torch.no_grad():
        for dirn in ("forward", "inverse"):
            lay.prqct.fused = True
            z, ld = getattr(lay, dirn)(dev(x), cg)
            lay.prqct.fused = False
            z2, ld2 = getattr(lay, dirn)(dev(x), cg)
            w32 = getattr(o32, dirn)(x, ctx)
            w64 = getattr(o64, dirn)(x.double(), ctx.double() if ctx_dim else None)
            parity(z, w32[0], w64[0], what="final-fused z " + dirn)
            parity(ld, w32[1], w64[1], rtol=1e-5, atol=2e-5 * max(1, d // 64), what="final-fused ld " + dirn)
            parity(z2, w32[0], w64[0], what="three-step z " + dirn)
        lay.prqct.fused = True
        lq = torch.full((batch,), 0.5, device="cuda")
        z3 = lay.inverse_into(dev(x), lq, **({"context": cg} if ctx_dim else {}))
        zi, ldi = lay.inverse(dev(x), cg)
        assert torch.equal(z3, zi) and torch.allclose(lq, 0.5 + ldi, rtol=1e-6, atol=1e-5)
    nf.check_discriminant()


# ---------------------------------------------------------------- next row: per-feature tails / bounds (G18)
@pytest.mark.parametrize("inv", [False, True])
def test_g18_per_feature_tails_functional(hip, inv):
    """Tails per feature + a tensor of per-feature bounds (splines.py:50-66), evaluated group by group.
    Deliberate deviation: elements outside their bound pass through (identity, zero log-det) as with
    uniform tails; the reference leaves their OUTPUT at zero when tails is a list (no assignment in
    splines.py:50-57)."""
    fx = fixture("g18_per_feature_tails")
    tails = ["linear", "circular", "linear", "circular", "circular", "linear"]
    tag = "fn/" + ("inv" if inv else "fwd")
    x, uw, uh, ud = (dev(T(fx["fn/" + n])) for n in ("x", "uw", "uh", "ud"))
    bound = dev(T(fx["fn/bound"]))
    with torch.no_grad():
        y, ld = nf.utils.splines.unconstrained_rational_quadratic_spline(x, uw, uh, ud, inverse=inv, tails=tails,
                                                                         tail_bound=bound)
    inside = ((x >= -bound) & (x <= bound)).cpu()
    y32, y64, l32, l64 = (T(fx[tag + "/" + n]) for n in ("y32", "y64", "ld32", "ld64"))
    parity(y.cpu()[inside], y32[inside], y64[inside], what="y inside")
    parity(ld.cpu()[inside], l32[inside], l64[inside], what="ld inside")
    assert torch.equal(y.cpu()[~inside], x.cpu()[~inside]) and not ld.cpu()[~inside].any()
    assert not y32[~inside].any()                      # the reference's zeros, documented above
    nf.check_discriminant()


@pytest.mark.parametrize("kind", ["scalar", "tensor"])
def test_g18_circular_coupled_layer(hip, kind):
    fx = fixture("g18_per_feature_tails")
    sd, _ = state_for(fx, "layer/" + kind, 1801, final_gain=2.0)
    tb = 3.0 if kind == "scalar" else T(fx["layer/bound"])
    lay = nf.flows.CircularCoupledRationalQuadraticSpline(7, 1, 32, ind_circ=[0, 3, 4], num_bins=8, tail_bound=tb,
                                                          init_identity=False)
    lay.load_state_dict(sd, strict=False)              # tail-bound / scale buffers keep their constructor values
    missing = set(lay.state_dict()) - set(sd)
    assert all(("tail_bound" in m) or m.endswith("preprocessing.scale") for m in missing), missing
    lay = lay.cuda()
    x = dev(T(fx["layer/x"]))
    with torch.no_grad():
        for dirn, fn in (("fwd", lay.forward), ("inv", lay.inverse)):
            z, ld = fn(x)
            parity(z, fx["layer/%s/%s_z32" % (kind, dirn)], fx["layer/%s/%s_z64" % (kind, dirn)], what=dirn + " z")
            parity(ld, fx["layer/%s/%s_ld32" % (kind, dirn)], fx["layer/%s/%s_ld64" % (kind, dirn)], rtol=1e-5,
                   atol=2e-5, what=dirn + " ld")
    nf.check_discriminant()
    # training path through the per-feature layer
    loss = lay.inverse(x)[1].mean()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in lay.parameters())


# ---------------------------------------------------------------- next row: autoregressive RQS (G19)
def _ar_layer(fx, tag):
    if tag == "plain":
        lay = nf.flows.AutoregressiveRationalQuadraticSpline(6, 1, 32, num_bins=8, tail_bound=3.0, init_identity=False)
    else:
        lay = nf.flows.CircularAutoregressiveRationalQuadraticSpline(
            6, 1, 32, ind_circ=[1, 4], num_bins=8, tail_bound=torch.tensor([3.0, float(np.pi), 3.0, 2.5, float(np.pi), 3.0]),
            permute_mask=True, init_identity=False)
    sd, _ = state_for(fx, tag, 1901, final_gain=2.0)
    for key, v in fx.items():
        if key.startswith(tag + "/mask/"):
            sd[key[len(tag) + 6:]] = T(v)
    missing = lay.load_state_dict(sd, strict=False).missing_keys
    assert all(("tail_bound" in m) or m.endswith("preprocessing.scale") for m in missing), missing
    return lay.cuda()


@pytest.mark.parametrize("tag", ["plain", "circular"])
def test_g19_autoregressive_rqs(hip, tag):
    """MADE conditioner on PyTorch-ROCm, splines on the packed kernel reading its [B, D*P] output in place:
    density direction one pass, sampling direction D passes."""
    fx = fixture("g19_autoregressive")
    lay = _ar_layer(fx, tag)
    x = dev(T(fx["x"]))
    with torch.no_grad():
        for dirn, fn in (("fwd", lay.forward), ("inv", lay.inverse)):
            z, ld = fn(x)
            parity(z, fx["%s/%s_z32" % (tag, dirn)], fx["%s/%s_z64" % (tag, dirn)], what=dirn + " z")
            parity(ld, fx["%s/%s_ld32" % (tag, dirn)], fx["%s/%s_ld64" % (tag, dirn)], rtol=1e-5, atol=2e-5,
                   what=dirn + " ld")
        z, ld = lay.forward(x)
        back, ld2 = lay.inverse(z)
        assert float((back - x).abs().mean()) < 2e-5 and float((ld + ld2).abs().mean()) < 1e-4
    nf.check_discriminant()
    loss = lay.inverse(x)[1].mean()                      # training path (density direction)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in lay.parameters())

# ---------------------------------------------------------------- next row (f4 tail): masked affine autoregressive flow (G22)
@pytest.mark.parametrize("tag", ["plain", "ctx"])
def test_g22_masked_affine_autoregressive(hip, tag):
    """MaskedAffineAutoregressive (flows/affine/autoregressive.py:48-103) against the reference (fixture G22): MADE
    conditioner on PyTorch-ROCm, the elementwise map on csrc/affine_kernels.hip::maf_affine_kernel reading the MADE
    output in place; one-pass forward, D-pass inverse, round trip, ragged batch, and the differentiable path."""
    fx = fixture("g22_maf")
    lay = (nf.flows.MaskedAffineAutoregressive(7, 24, num_blocks=2) if tag == "plain"
           else nf.flows.MaskedAffineAutoregressive(7, 24, context_features=3, num_blocks=1))
    sd, _ = state_for(fx, tag, 2201, final_gain=1.0)
    for key, v in fx.items():
        if key.startswith(tag + "/mask/"):
            sd[key[len(tag) + 6:]] = T(v)
    lay.load_state_dict(sd)
    lay = lay.cuda()
    x = dev(T(fx["x"]))
    kw = {"context": dev(T(fx["ctx"]))} if tag == "ctx" else {}
    with torch.no_grad():
        for dirn, fn in (("fwd", lay.forward), ("inv", lay.inverse)):
            z, ld = fn(x, **kw)
            parity(z, fx["%s/%s_z32" % (tag, dirn)], fx["%s/%s_z64" % (tag, dirn)], what=dirn + " z")
            parity(ld, fx["%s/%s_ld32" % (tag, dirn)], fx["%s/%s_ld64" % (tag, dirn)], rtol=1e-5, atol=2e-5, what=dirn + " ld")
        z, ld = lay.forward(x, **kw)
        back, ld2 = lay.inverse(z, **kw)
        assert float((back - x).abs().max()) < 1e-4 and float((ld + ld2).abs().max()) < 1e-4
        one, ld1 = lay.forward(x[:1], **({"context": kw["context"][:1]} if kw else {}))
        # (not bitwise: the MADE's library GEMMs pick their algorithm by batch size)
        assert_close(one, z[:1].cpu(), rtol=1e-5, atol=1e-5, what="batch of one z")
        assert_close(ld1, ld[:1].cpu(), rtol=1e-5, atol=1e-5, what="batch of one ld")
        # the kernel against the reference's own composition (:75-81) on the same MADE output
        params = lay.autoregressive_net(x, kw.get("context"))
        p = params.view(-1, 7, 2)
        scale = torch.sigmoid(p[..., 0] + 2.) + 1e-3
        assert_close(z, (scale * x + p[..., 1]).cpu(), rtol=1e-6, atol=1e-6, what="kernel vs torch composition")
    xg = x.clone().requires_grad_(True)
    zg, ldg = lay.forward(xg, **kw)
    (zg.sum() + ldg.sum()).backward()
    assert torch.isfinite(xg.grad).all() and all(p.grad is not None for p in lay.parameters() if p.requires_grad)
    assert_close(zg.detach(), z.cpu(), rtol=1e-5, atol=1e-5, what="differentiable path")

