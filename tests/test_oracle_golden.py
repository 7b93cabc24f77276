"""Pin the CPU oracle to the reference: every oracle function is run on the
fixture inputs and compared with what the reference itself produced
(tests/golden/*.npz, written by tests/golden/make_golden.py).  fp32 results use
the same torch CPU ops in the same order, so they agree to a few ulp; fp64 runs
agree to ~1e-12.  Integer items are bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import (fixture, T, state_for, assert_close, oracle_rqs_coupling,
                     oracle_c3_stack, oracle_affine_stack, oracle_crqs_stack,
                     oracle_glow_multiscale, glow_state)
from oracle import rqs as OR, masks as OM, nets as ON, layers as OL

F32 = dict(rtol=2e-6, atol=2e-6)
F64 = dict(rtol=1e-11, atol=1e-11)


def _both(fx, key):
    return ((torch.float32, fx[key + "32"], F32), (torch.float64, fx[key + "64"], F64))


@pytest.mark.parametrize("k", [8, 10, 16])
@pytest.mark.parametrize("inv", [False, True])
def test_g1_rq_spline(k, inv):
    fx = fixture("g1_rqs")
    tag = "K%d_%s" % (k, "inv" if inv else "fwd")
    for dt, suf, tol in ((torch.float32, "32", F32), (torch.float64, "64", F64)):
        a = [T(fx["K%d/%s" % (k, n)], dt) for n in ("x", "uw", "uh", "ud")]
        y, ld = OR.rq_spline(*a, inverse=inv)
        assert_close(y, fx[tag + "/y" + suf], what=tag + " y", **tol)
        assert_close(ld, fx[tag + "/ld" + suf], what=tag + " ld", **tol)


@pytest.mark.parametrize("case", ["K8_T3", "K8_T1", "K16_T5", "K5_T2.5"])
@pytest.mark.parametrize("inv", [False, True])
def test_g2_rq_spline_tails(case, inv):
    fx = fixture("g2_rqs_tails")
    tb = float(case.split("_T")[1])
    tag = case + ("_inv" if inv else "_fwd")
    for dt, suf, tol in ((torch.float32, "32", F32), (torch.float64, "64", F64)):
        a = [T(fx["%s/%s" % (case, n)], dt) for n in ("x", "uw", "uh", "ud")]
        y, ld = OR.rq_spline_tails(*a, inverse=inv, tails="linear", tail_bound=tb)
        assert_close(y, fx[tag + "/y" + suf], what=tag + " y", **tol)
        assert_close(ld, fx[tag + "/ld" + suf], what=tag + " ld", **tol)


@pytest.mark.parametrize("rm", [0, 1])
def test_g3_crqs_layer(rm):
    fx = fixture("g3_crqs_layer")
    for dt, suf, tol in ((torch.float32, "32", F32), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, "rm%d" % rm, 301 + rm, dt, final_gain=2.0)
        lay = oracle_rqs_coupling(sd, "prqct.", 8, 3.0, 128)
        x = T(fx["x"], dt)
        z, ld = lay.inverse(x)
        assert_close(z, fx["rm%d/inv_z%s" % (rm, suf)], what="inv z", **tol)
        assert_close(ld, fx["rm%d/inv_ld%s" % (rm, suf)], what="inv ld", **tol)
        z, ld = lay.forward(x)
        assert_close(z, fx["rm%d/fwd_z%s" % (rm, suf)], what="fwd z", **tol)
        assert_close(ld, fx["rm%d/fwd_ld%s" % (rm, suf)], what="fwd ld", **tol)


@pytest.mark.parametrize("tag", ["d64", "d21", "d7k4"])
def test_g4_conditional_coupling(tag):
    fx = fixture("g4_cond_prqc")
    d, c, h, nb, k, tb, kind = fx[tag + "/cfg"]
    for dt, suf, tol in ((torch.float32, "32", F32), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, tag, 401 + int(d), dt, final_gain=2.0)
        lay = oracle_rqs_coupling(sd, "", int(k), float(tb), int(h))
        x, ctx = T(fx[tag + "/x"], dt), T(fx[tag + "/ctx"], dt)
        z, ld = lay.nsf_forward(x, ctx)
        assert_close(z, fx[tag + "/nsf_fwd_z" + suf], what="fwd z", **tol)
        assert_close(ld, fx[tag + "/nsf_fwd_ld" + suf], what="fwd ld", **tol)
        z, ld = lay.nsf_inverse(x, ctx)
        assert_close(z, fx[tag + "/nsf_inv_z" + suf], what="inv z", **tol)
        assert_close(ld, fx[tag + "/nsf_inv_ld" + suf], what="inv ld", **tol)


def test_g5_c3_stack_stress_weights():
    fx = fixture("g5_c3_stack")
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=1e-5, atol=2e-5)), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, "c3_stress", 501, dt, final_gain=6.0)
        sd.update({k[len("c3/int/"):]: T(v) for k, v in fx.items() if k.startswith("c3/int/")})
        st = oracle_c3_stack(sd)
        x, ctx, eps = (T(fx[n], dt) for n in ("x", "ctx", "eps"))
        assert_close(st.log_prob(x, ctx), fx["c3_stress/lp" + suf], what="log_prob", **tol)
        z, lq = st.sample_from(eps, ctx)
        assert_close(z, fx["c3_stress/s_z" + suf], what="sample z", **tol)
        assert_close(lq, fx["c3_stress/s_logq" + suf], what="sample log_q", **tol)


def test_g5_c3_stack():
    fx = fixture("g5_c3_stack")
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=1e-5, atol=2e-5)), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, "c3", 501, dt, final_gain=1.0)
        st = oracle_c3_stack(sd)
        x, ctx, eps = (T(fx[n], dt) for n in ("x", "ctx", "eps"))
        tr = []
        lp = st.log_prob(x, ctx, trace=tr)
        assert_close(lp, fx["c3/lp" + suf], what="log_prob", **tol)
        assert_close(tr[-1][0], fx["c3/lp_z" + suf], what="latent", **tol)
        assert_close(torch.stack([t[1] for t in tr]), fx["c3/lp_lds" + suf], what="layer log_dets", **tol)
        tr = []
        z, lq = st.sample_from(eps, ctx, trace=tr)
        assert_close(z, fx["c3/s_z" + suf], what="sample z", **tol)
        assert_close(lq, fx["c3/s_logq" + suf], what="sample log_q", **tol)
        assert_close(torch.stack([t[1] for t in tr]), fx["c3/s_lds" + suf], what="sample log_dets", **tol)


@pytest.mark.parametrize("d", [2, 32, 33])
@pytest.mark.parametrize("sm", ["exp", "sigmoid", "sigmoid_inv", "noscale"])
@pytest.mark.parametrize("mode", ["channel", "channel_inv"])
def test_g6_affine_block(d, sm, mode):
    fx = fixture("g6_affine")
    tag = "d%d/%s/%s" % (d, sm, mode)
    for dt, suf, tol in ((torch.float32, "32", F32), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, tag, 601 + d, dt)
        blk = OL.AffineCouplingBlock(lambda z: ON.mlp(sd, "flows.1.param_map.", z), scale=sm != "noscale",
                                     scale_map=sm if sm != "noscale" else "exp", split_mode=mode)
        x = T(fx["d%d/x" % d], dt)
        for dirn, fn in (("fwd", blk.forward), ("inv", blk.inverse)):
            z, ld = fn(x.clone())
            assert_close(z, fx["%s/%s_z%s" % (tag, dirn, suf)], what=dirn + " z", **tol)
            assert_close(ld, fx["%s/%s_ld%s" % (tag, dirn, suf)], what=dirn + " ld", **tol)


@pytest.mark.parametrize("d", [2, 9, 30])
@pytest.mark.parametrize("variant", ["st", "t_only", "s_only", "inf"])
def test_g7_masked_affine(d, variant):
    fx = fixture("g7_masked_affine")
    tag = "d%d/%s" % (d, variant)
    for dt, suf, tol in ((torch.float32, "32", F32), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, tag, 701 + d, dt)
        if variant == "inf":
            sd["s.net.2.bias"][1] = float("inf")
            sd["t.net.2.bias"][d - 1] = float("-inf")
        s_fn = (lambda z: ON.mlp(sd, "s.", z)) if variant != "t_only" else None
        t_fn = (lambda z: ON.mlp(sd, "t.", z)) if variant != "s_only" else None
        lay = OL.MaskedAffine(T(fx["d%d/b" % d], dt).view(1, -1), s_fn, t_fn)
        x = T(fx["d%d/x" % d], dt)
        for dirn, fn in (("fwd", lay.forward), ("inv", lay.inverse)):
            z, ld = fn(x)
            assert_close(z, fx["%s/%s_z%s" % (tag, dirn, suf)], what=dirn + " z", **tol)
            assert_close(ld, fx["%s/%s_ld%s" % (tag, dirn, suf)], what=dirn + " ld", **tol)


def test_g8_indices_bit_exact():
    fx = fixture("g8_indices")
    for d in (2, 5, 32, 64, 1024):
        for seed in (0, 7):
            torch.manual_seed(seed)
            perm, inv = OM.shuffle_perm(d)
            assert np.array_equal(perm.numpy(), fx["perm/d%d_s%d/perm" % (d, seed)])
            assert np.array_equal(inv.numpy(), fx["perm/d%d_s%d/inv_perm" % (d, seed)])
    for d in (2, 5, 32, 33):
        x = T(fx["swap/d%d/x" % d])
        p = OL.Permute(d, "swap")
        assert np.array_equal(p.forward(x)[0].numpy(), fx["swap/d%d/fwd" % d])
        assert np.array_equal(p.inverse(x)[0].numpy(), fx["swap/d%d/inv" % d])
    for d in (1, 2, 7, 64, 1024):
        for name, got in (("alt_even", OM.alternating_mask(d, True)), ("alt_odd", OM.alternating_mask(d, False)),
                          ("mid", OM.mid_split_mask(d))):
            want = fx["mask/%s/d%d" % (name, d)]
            assert got.dtype == torch.uint8 and np.array_equal(got.numpy(), want)
        for seed in (0, 3):
            assert np.array_equal(OM.random_mask(d, seed).numpy(), fx["mask/rand/d%d_s%d" % (d, seed)])
    torch.manual_seed(11)
    assert np.array_equal(OM.random_mask(64).numpy(), fx["mask/rand_global/d64_ms11"])
    for rm in (0, 1):
        idf, tf = OM.split_features(OM.alternating_mask(9, even=bool(rm)))
        assert np.array_equal(idf.numpy(), fx["crqs_idx/rm%d/identity" % rm])
        assert np.array_equal(tf.numpy(), fx["crqs_idx/rm%d/transform" % rm])


@pytest.mark.parametrize("tag", ["d2_Tnone", "d64_Tnone", "d64_T0.7", "d33_T1.9"])
def test_g9_diag_gaussian(tag):
    fx = fixture("g9_diag_gaussian")
    d = int(tag[1:].split("_")[0])
    temp = fx[tag + "/temp"][0]
    temp = None if np.isnan(temp) else float(temp)
    for dt, suf, tol in ((torch.float32, "32", F32), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, tag, 901 + d, dt)
        q = OL.DiagGaussian(sd["loc"], sd["log_scale"], temp)
        assert_close(q.log_prob(T(fx[tag + "/z"], dt)), fx[tag + "/logp" + suf], what="log_prob", **tol)
        z, lp = q.from_noise(T(fx[tag + "/eps" + suf], dt))
        assert_close(z, fx[tag + "/s_z" + suf], what="sample z", **tol)
        assert_close(lp, fx[tag + "/s_logp" + suf], what="sample logp", **tol)


@pytest.mark.parametrize("name,tag,layers,d,seed", [("g10_c1_two_moons", "c1", 4, 2, 1001),
                                                    ("g12_c2_tabular", "c2", 8, 32, 1201)])
def test_affine_stacks(name, tag, layers, d, seed):
    fx = fixture(name)
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=1e-5, atol=1e-5)), (torch.float64, "64", F64)):
        sd, ents = state_for(fx, tag, seed, dt, weight_gain=0.4 if tag == "c2" else 1.0)
        st = oracle_affine_stack(sd, layers, d)
        lp = st.log_prob(T(fx["x"], dt))
        assert_close(lp, fx[tag + "/lp" + suf], what="log_prob", **tol)
        z, lq = st.sample_from(T(fx["eps"], dt))
        assert_close(z, fx[tag + "/s_z" + suf], what="sample z", **tol)
        assert_close(lq, fx[tag + "/s_logq" + suf], what="sample log_q", **tol)


def test_g13_c5_layer_shape():
    """Config C5's layer shape (D=1024, K=16, ResidualNet 512 -> 24064), 2 layers."""
    fx = fixture("g13_c5_shape")
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=1e-5, atol=2e-4)), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, "c5", 1301, dt, final_gain=1.0)
        st = oracle_crqs_stack(sd, 2, 16, 3.0, 128)
        assert_close(st.log_prob(T(fx["x"], dt)), fx["c5/lp" + suf], what="log_prob", **tol)
        z, lq = st.sample_from(T(fx["eps"], dt))
        assert_close(z, fx["c5/s_z" + suf], what="sample z", **tol)
        assert_close(lq, fx["c5/s_logq" + suf], what="sample log_q", **tol)


def test_g11_glow_multiscale():
    """Config C4's family: MultiscaleFlow of GlowBlocks (affine coupling with a conv
    conditioner and sigmoid scale map, LU 1x1 convolution, ActNorm) + Squeeze / Merge."""
    fx = fixture("g11_glow_multiscale")
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=1e-5, atol=1e-3)), (torch.float64, "64", F64)):
        sd = glow_state(fx, 1101, dt)
        ms = oracle_glow_multiscale(sd)
        assert_close(ms.log_prob(T(fx["x"], dt)), fx["glow/lp" + suf], what="log_prob", **tol)
        z, lq = ms.sample_from([T(fx["eps0"], dt), T(fx["eps1"], dt)])
        assert_close(z, fx["glow/s_z" + suf], what="sample z", **tol)
        assert_close(lq, fx["glow/s_logq" + suf], what="sample log_q", **tol)


def test_g20_c4_real_shape():
    """Config C4 at its real shape (example/glow.ipynb cell 2: 3 x 32 x 32, L = 3, K = 16 GlowBlocks per level,
    256 hidden channels), 4 images: the oracle reproduces the reference's fp32 and fp64 outputs."""
    fx = fixture("g20_c4_real_shape")
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=2e-5, atol=2e-2)), (torch.float64, "64", F64)):
        sd = glow_state(fx, 2001, dt, weight_gain=0.1, other_gain=0.02)
        ms = oracle_glow_multiscale(sd, levels=3, blocks=16)
        assert_close(ms.log_prob(T(fx["x"], dt)), fx["glow/lp" + suf], what="log_prob", **tol)
        z, lq = ms.sample_from([T(fx["eps%d" % i], dt) for i in range(3)])
        assert_close(z, fx["glow/s_z" + suf], what="sample z", rtol=tol["rtol"], atol=min(tol["atol"], 1e-3))
        assert_close(lq, fx["glow/s_logq" + suf], what="sample log_q", **tol)


def test_g21_c5_real_depth():
    """Config C5 at its real depth: 24 RQS couplings, D = 1024, K = 16, ResidualNet 512 -> 24064 (hidden 128, 2 blocks)."""
    fx = fixture("g21_c5_real_depth")
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=1e-5, atol=2e-3)), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, "c5", 2101, dt, final_gain=1.0)
        st = oracle_crqs_stack(sd, 24, 16, 3.0, 128)
        assert_close(st.log_prob(T(fx["x"], dt)), fx["c5/lp" + suf], what="log_prob", **tol)
        z, lq = st.sample_from(T(fx["eps"], dt))
        assert_close(z, fx["c5/s_z" + suf], what="sample z", rtol=tol["rtol"], atol=min(tol["atol"], 2e-4))
        assert_close(lq, fx["c5/s_logq" + suf], what="sample log_q", **tol)


# ---------------------------------------------------------------- next rows
@pytest.mark.parametrize("d", [5, 64])
def test_g14_lu_linear_permute(d):
    fx = fixture("g14_lu_linear_permute")
    tag = "d%d" % d
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=2e-4, atol=2e-4)), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, tag, 1401 + d, dt, weight_gain=0.5)
        lay = OL.LULinearPermute(sd["permutation._permutation"], sd["linear.bias"], sd["linear.lower_entries"],
                                 sd["linear.upper_entries"], sd["linear.unconstrained_upper_diag"])
        x = T(fx[tag + "/x"], dt)
        for dirn, fn in (("fwd", lay.forward), ("inv", lay.inverse)):
            z, ld = fn(x)
            assert_close(z, fx["%s/%s_z%s" % (tag, dirn, suf)], what=dirn + " z", **tol)
            assert_close(ld, fx["%s/%s_ld%s" % (tag, dirn, suf)], what=dirn + " ld", **tol)
        back, _ = lay.inverse(lay.forward(x)[0])
        assert_close(back, x, what="round trip", rtol=1e-3 if dt == torch.float32 else 1e-9,
                     atol=1e-3 if dt == torch.float32 else 1e-9)


@pytest.mark.parametrize("mode", ["checkerboard", "checkerboard_inv"])
def test_g15_checkerboard(mode):
    fx = fixture("g15_checkerboard")
    inv = mode.endswith("inv")
    for name in ("2d", "3d", "4d"):
        z = T(fx[name + "/z"])
        z1, z2 = OL.checker_split(z, inv)
        assert np.array_equal(z1.numpy(), fx["%s/%s/z1" % (name, mode)])
        assert np.array_equal(z2.numpy(), fx["%s/%s/z2" % (name, mode)])
        assert torch.equal(OL.checker_merge(z1, z2, inv), z)
    for dt, suf, tol in ((torch.float32, "32", F32), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, "blk/" + mode, 1501, dt)
        blk = OL.AffineCouplingBlock(lambda z: ON.mlp(sd, "flows.1.param_map.", z), split_mode=mode)
        x = T(fx["blk/x"], dt)
        for dirn, fn in (("fwd", blk.forward), ("inv", blk.inverse)):
            z, ld = fn(x.clone())
            assert_close(z, fx["blk/%s/%s_z%s" % (mode, dirn, suf)], what=dirn + " z", **tol)
            assert_close(ld, fx["blk/%s/%s_ld%s" % (mode, dirn, suf)], what=dirn + " ld", **tol)


@pytest.mark.parametrize("case", ["K8_T3", "K5_T2.5"])
@pytest.mark.parametrize("inv", [False, True])
def test_g16_circular_tails(case, inv):
    fx = fixture("g16_circular")
    tb = float(case.split("_T")[1])
    tag = case + ("_inv" if inv else "_fwd")
    for dt, suf, tol in ((torch.float32, "32", F32), (torch.float64, "64", F64)):
        args = [T(fx["%s/%s" % (case, n)], dt) for n in ("x", "uw", "uh", "ud")]
        y, ld = OR.rq_spline_tails(*args, inverse=inv, tails="circular", tail_bound=tb)
        assert_close(y, fx[tag + "/y" + suf], what="y", **tol)
        assert_close(ld, fx[tag + "/ld" + suf], what="ld", **tol)


def test_g16_circular_coupling_layer():
    fx = fixture("g16_circular")
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=2e-5, atol=2e-5)), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, "layer", 1651, dt, final_gain=2.0)
        lay = oracle_rqs_coupling(sd, "", 6, 2.0, 32, tails="circular")
        x, ctx = T(fx["layer/x"], dt), T(fx["layer/ctx"], dt)
        for dirn, fn in (("nsf_fwd", lay.nsf_forward), ("nsf_inv", lay.nsf_inverse)):
            z, ld = fn(x, ctx)
            assert_close(z, fx["layer/%s_z%s" % (dirn, suf)], what=dirn + " z", **tol)
            assert_close(ld, fx["layer/%s_ld%s" % (dirn, suf)], what=dirn + " ld", **tol)


@pytest.mark.parametrize("tag", ["noctx", "ctx"])
def test_g17_image_rqs_coupling(tag):
    from helpers import oracle_image_rqs_coupling
    fx = fixture("g17_image_rqs")
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=2e-5, atol=2e-5)), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, tag, 1701, dt, final_gain=2.0)
        lay = oracle_image_rqs_coupling(sd)
        x = T(fx["x"], dt)
        ctx = T(fx["ctx"], dt) if tag == "ctx" else None
        for dirn, fn in (("nsf_fwd", lay.nsf_forward), ("nsf_inv", lay.nsf_inverse)):
            z, ld = fn(x, ctx)
            assert_close(z, fx["%s/%s_z%s" % (tag, dirn, suf)], what=dirn + " z", **tol)
            assert_close(ld, fx["%s/%s_ld%s" % (tag, dirn, suf)], what=dirn + " ld", **tol)


@pytest.mark.parametrize("inv", [False, True])
def test_g18_per_feature_tails_functional(inv):
    fx = fixture("g18_per_feature_tails")
    tails = ["linear", "circular", "linear", "circular", "circular", "linear"]
    tag = "fn/" + ("inv" if inv else "fwd")
    for dt, suf, tol in ((torch.float32, "32", F32), (torch.float64, "64", F64)):
        args = [T(fx["fn/" + n], dt) for n in ("x", "uw", "uh", "ud")]
        y, ld = OR.rq_spline_tails(*args, inverse=inv, tails=tails, tail_bound=T(fx["fn/bound"], dt))
        assert_close(y, fx[tag + "/y" + suf], what="y", **tol)
        assert_close(ld, fx[tag + "/ld" + suf], what="ld", **tol)


@pytest.mark.parametrize("kind", ["scalar", "tensor"])
def test_g18_circular_coupled_layer(kind):
    from helpers import oracle_circular_layer
    fx = fixture("g18_per_feature_tails")
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=2e-5, atol=2e-5)), (torch.float64, "64", F64)):
        sd, _ = state_for(fx, "layer/" + kind, 1801, dt, final_gain=2.0)
        tb = 3.0 if kind == "scalar" else T(fx["layer/bound"])
        lay = oracle_circular_layer(sd, 7, [0, 3, 4], tb)
        x = T(fx["layer/x"], dt)
        for dirn, fn in (("fwd", lay.forward), ("inv", lay.inverse)):
            z, ld = fn(x)
            assert_close(z, fx["layer/%s/%s_z%s" % (kind, dirn, suf)], what=dirn + " z", **tol)
            assert_close(ld, fx["layer/%s/%s_ld%s" % (kind, dirn, suf)], what=dirn + " ld", **tol)


def _ar_state(fx, tag, dt):
    sd, _ = state_for(fx, tag, 1901, dt, final_gain=2.0)
    for key, v in fx.items():
        if key.startswith(tag + "/mask/"):
            sd[key[len(tag) + 6:]] = T(v, dt)
    return sd


def _oracle_ar(sd, tag, dt):
    pre = "mprqat.autoregressive_net."
    if tag == "plain":
        return OL.AutoregressiveRQS(lambda x: ON.made(sd, pre, x), 6, 8, "linear", 3.0)
    bound = torch.tensor([3.0, float(np.pi), 3.0, 2.5, float(np.pi), 3.0])
    tails = ["circular" if i in (1, 4) else "linear" for i in range(6)]
    scale = (np.pi / bound[[1, 4]]).to(dt)
    pp = lambda x: ON.periodic_features(sd, pre + "preprocessing.", x, scale)
    return OL.AutoregressiveRQS(lambda x: ON.made(sd, pre, x, preprocess=pp), 6, 8, tails, bound.to(dt))


@pytest.mark.parametrize("tag", ["plain", "circular"])
def test_g19_autoregressive_rqs(tag):
    fx = fixture("g19_autoregressive")
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=2e-5, atol=2e-5)), (torch.float64, "64", F64)):
        lay = _oracle_ar(_ar_state(fx, tag, dt), tag, dt)
        x = T(fx["x"], dt)
        for dirn, fn in (("fwd", lay.forward), ("inv", lay.inverse)):
            z, ld = fn(x)
            assert_close(z, fx["%s/%s_z%s" % (tag, dirn, suf)], what=dirn + " z", **tol)
            assert_close(ld, fx["%s/%s_ld%s" % (tag, dirn, suf)], what=dirn + " ld", **tol)

def _maf_state(fx, tag, dt):
    sd, _ = state_for(fx, tag, 2201, dt, final_gain=1.0)
    for key, v in fx.items():
        if key.startswith(tag + "/mask/"):
            sd[key[len(tag) + 6:]] = T(v, dt)
    return sd


@pytest.mark.parametrize("tag", ["plain", "ctx"])
def test_g22_masked_affine_autoregressive(tag):
    """Oracle restatement of the MAF layer against the reference's outputs (fixture G22), fp32 and fp64, both
    directions; the context case runs the MADE's GLU context layers."""
    fx = fixture("g22_maf")
    for dt, suf, tol in ((torch.float32, "32", dict(rtol=2e-5, atol=2e-5)), (torch.float64, "64", F64)):
        sd = _maf_state(fx, tag, dt)
        ctx = T(fx["ctx"], dt) if tag == "ctx" else None
        lay = OL.MaskedAffineAutoregressive(lambda x: ON.made(sd, "autoregressive_net.", x, context=ctx), 7)
        x = T(fx["x"], dt)
        for dirn, fn in (("fwd", lay.forward), ("inv", lay.inverse)):
            z, ld = fn(x)
            assert_close(z, fx["%s/%s_z%s" % (tag, dirn, suf)], what=dirn + " z", **tol)
            assert_close(ld, fx["%s/%s_ld%s" % (tag, dirn, suf)], what=dirn + " ld", **tol)


# ---------------------------------------------------------------- G23: the reference's own gradients
def _g23_oracle_grads(fx, case, dt):
    from helpers import g23_reference
    tag, seed, gains, in_names, gz_name, make, loss = case
    sd, _ = state_for(fx, tag, seed, dt, **gains)
    if tag.startswith("masked"):
        sd["b"] = T(fx["masked/b"], dt).view(1, -1)
    names, ref = g23_reference(fx, tag, len(in_names))
    leaves = {k: (v.clone().requires_grad_() if (v.is_floating_point() and k in names) else v) for k, v in sd.items()}
    xs = [T(fx[n], dt).requires_grad_() for n in in_names]
    gz = T(fx[gz_name], dt) if gz_name else None
    val = loss(make(leaves), *xs, gz)
    g = torch.autograd.grad(val, xs + [leaves[n] for n in names])
    return names, ref, val.detach(), g[:len(xs)], g[len(xs):]


def _g23_ids():
    from helpers import g23_cases
    return [c[0] for c in g23_cases()]


@pytest.mark.parametrize("idx", range(8), ids=_g23_ids())
def test_g23_oracle_autograd_reproduces_reference_gradients(idx):
    """Gradient parity pinned to the reference: torch autograd over the ORACLE reproduces the
    gradients the reference's own autograd produced (tests/golden/make_golden.py::g23_gradients) - fp64 to 1e-9
    relative to the gradient's rms, fp32 within 4x the reference's own fp32-vs-fp64 gradient error + 2e-5 of the rms
    (different but equivalent fp32 operation orders inside torch's backward formulas)."""
    from helpers import g23_cases
    fx = fixture("g23_gradients")
    case = g23_cases()[idx]
    for dt, prec in ((torch.float64, "64"), (torch.float32, "32")):
        names, ref, val, gin, gpar = _g23_oracle_grads(fx, case, dt)
        loss_ref, gin_ref, gpar_ref = ref[prec]
        assert abs(float(val) - loss_ref) <= (1e-10 if prec == "64" else 2e-5) * max(1.0, abs(loss_ref))
        pairs = [("input%d" % i, a, b, ref["64"][1][i]) for i, (a, b) in enumerate(zip(gin, gin_ref))]
        pairs += [(n, a, gpar_ref[n], ref["64"][2][n]) for n, a in zip(names, gpar)]
        for lab, got, want, want64 in pairs:
            rms = float(want64.double().pow(2).mean().sqrt()) + 1e-30
            err = (got.double() - want.double()).abs()
            if prec == "64":
                assert float(err.max()) <= 1e-9 * rms, "%s %s fp64: %.3e (rms %.3e)" % (case[0], lab, float(err.max()), rms)
            else:
                noise = (want.double() - want64.double()).abs()
                bound = 2e-5 * rms + 4.0 * noise + 4.0 * float(noise.mean())
                bad = float((err > bound).double().mean())
                assert bad <= 1e-3, "%s %s fp32: %.2f %% outside, max err %.3e, rms %.3e" % (case[0], lab, 100 * bad, float(err.max()), rms)
