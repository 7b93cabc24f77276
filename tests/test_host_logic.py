"""Host-side logic that needs no GPU: integer parity items (masks, permutation
buffers, feature index buffers) against the reference's golden values, the
state_dict layout of every module against the reference's entry lists, error
conventions, and the refusal to compute on CPU tensors (no fallback path)."""
import numpy as np
import pytest
import torch

import synth
import vcnf_amd as nf
from helpers import fixture


def test_masks_bit_exact():
    fx = fixture("g8_indices")
    for d in (1, 2, 7, 64, 1024):
        for name, got in (("alt_even", nf.utils.create_alternating_binary_mask(d, True)),
                          ("alt_odd", nf.utils.create_alternating_binary_mask(d, False)),
                          ("mid", nf.utils.create_mid_split_binary_mask(d))):
            assert got.dtype == torch.uint8
            assert np.array_equal(got.numpy(), fx["mask/%s/d%d" % (name, d)])
        for seed in (0, 3):
            assert np.array_equal(nf.utils.create_random_binary_mask(d, seed=seed).numpy(),
                                  fx["mask/rand/d%d_s%d" % (d, seed)])
    torch.manual_seed(11)
    assert np.array_equal(nf.utils.create_random_binary_mask(64).numpy(), fx["mask/rand_global/d64_ms11"])


def test_permute_buffers_bit_exact():
    fx = fixture("g8_indices")
    for d in (2, 5, 32, 64, 1024):
        for seed in (0, 7):
            torch.manual_seed(seed)
            p = nf.flows.Permute(d, mode="shuffle")
            assert p.perm.dtype == torch.int64
            assert np.array_equal(p.perm.numpy(), fx["perm/d%d_s%d/perm" % (d, seed)])
            assert np.array_equal(p.inv_perm.numpy(), fx["perm/d%d_s%d/inv_perm" % (d, seed)])
            assert sorted(p.state_dict()) == ["inv_perm", "perm"]
    # swap as a gather index reproduces the reference's outputs exactly
    for d in (2, 5, 32, 33):
        x = fx["swap/d%d/x" % d]
        p = nf.flows.Permute(d, mode="swap")
        assert np.array_equal(x[:, p.gather_index(False).numpy()], fx["swap/d%d/fwd" % d])
        assert np.array_equal(x[:, p.gather_index(True).numpy()], fx["swap/d%d/inv" % d])
        assert len(p.state_dict()) == 0


def test_feature_index_buffers_bit_exact():
    fx = fixture("g8_indices")
    for rm in (0, 1):
        m = nf.flows.CoupledRationalQuadraticSpline(9, 1, 8, 4, reverse_mask=bool(rm))
        assert np.array_equal(m.prqct.identity_features.numpy(), fx["crqs_idx/rm%d/identity" % rm])
        assert np.array_equal(m.prqct.transform_features.numpy(), fx["crqs_idx/rm%d/transform" % rm])
        assert m.prqct.identity_features.dtype == torch.int64


def _entries(module):
    return [(k, tuple(v.shape)) for k, v in module.state_dict().items() if v.is_floating_point()]


def _ref_entries(fx, tag):
    return synth.decode_entries(fx[tag + "/entries"])


def test_state_dict_layout_matches_reference():
    """Names, order and shapes of all float entries equal the reference's, and the
    reference's integer buffers load by name."""
    fx = fixture("g3_crqs_layer")
    m = nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8)
    assert _entries(m) == _ref_entries(fx, "rm0")
    ints = {k[len("rm0/int/"):] for k in fx if k.startswith("rm0/int/")}
    assert ints == {k for k, v in m.state_dict().items() if not v.is_floating_point()}

    fx = fixture("g5_c3_stack")
    flows = [nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, reverse_mask=bool(i % 2),
                                                     num_context_channels=16) for i in range(12)]
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(64), flows)
    assert _entries(model) == _ref_entries(fx, "c3")

    fx = fixture("g4_cond_prqc")
    from vcnf_amd.nets import ResidualNet
    net = lambda i, o: ResidualNet(i, o, hidden_features=48, context_features=5, num_blocks=1)
    m = nf.flows.PiecewiseRationalQuadraticCoupling(nf.utils.create_mid_split_binary_mask(21), net, num_bins=10,
                                                    tails="linear", tail_bound=2.0,
                                                    apply_unconditional_transform=True)
    assert _entries(m) == _ref_entries(fx, "d21")

    fx = fixture("g6_affine")
    m = nf.flows.AffineCouplingBlock(nf.nets.MLP([17, 24, 24, 32]))
    assert _entries(m) == _ref_entries(fx, "d33/exp/channel")

    fx = fixture("g7_masked_affine")
    b = torch.tensor([1.0 if i % 2 == 0 else 0.0 for i in range(9)])
    m = nf.flows.MaskedAffineFlow(b, nf.nets.MLP([9, 16, 9]), nf.nets.MLP([9, 16, 9]))
    assert [e for e in _entries(m) if e[0] != "b"] == _ref_entries(fx, "d9/st")
    assert m.state_dict()["b"].shape == (1, 9)

    fx = fixture("g10_c1_two_moons")
    flows = []
    for _ in range(4):
        flows += [nf.flows.AffineCouplingBlock(nf.nets.MLP([1, 32, 32, 2], init_zeros=True)),
                  nf.flows.Permute(2, mode="swap")]
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(2), flows)
    assert _entries(model) == _ref_entries(fx, "c1")
    assert model.categoricals is None

    fx = fixture("g9_diag_gaussian")
    assert _entries(nf.distributions.DiagGaussian(64)) == _ref_entries(fx, "d64_Tnone")

    # Glow block: keys and order of the reference (conv conditioner, LU 1x1 conv, ActNorm)
    fx = fixture("g11_glow_multiscale")
    blk = nf.flows.GlowBlock(12, 16)
    ref = [e for e in _ref_entries(fx, "glow") if e[0].startswith("flows.1.0.")]
    fixed = ("P", "sign_S", "eye", "data_dep_init_done")
    got = [("flows.1.0." + k, s) for k, s in _entries(blk) if k.split(".")[-1] not in fixed]
    assert got == ref
    assert {k.split(".")[-1] for k in blk.state_dict()} >= set(fixed)


def test_initialisation_conventions():
    m = nf.flows.CoupledRationalQuadraticSpline(8, 2, 16, 8)
    u = m.prqct.unconditional_transform
    assert torch.all(u.unnormalized_widths == 0) and torch.all(u.unnormalized_heights == 0)
    edge = np.log(np.exp(1 - 1e-3) - 1)
    assert torch.allclose(u.unnormalized_derivatives, torch.tensor(edge, dtype=torch.float32))
    assert u.unnormalized_derivatives.shape == (4, 7)
    last = m.prqct.transform_net.blocks[1].linear_layers[1]
    assert last.weight.abs().max() <= 1e-3 and last.bias.abs().max() <= 1e-3    # resnet.py:34-36
    assert m.prqct.transform_net.final_layer.out_features == 4 * 23
    z = nf.nets.MLP([3, 8, 4], init_zeros=True)
    assert torch.all(z.net[2].weight == 0) and torch.all(z.net[2].bias == 0)


def test_error_conventions():
    net = lambda i, o: nf.nets.ResidualNet(i, o, hidden_features=8)
    with pytest.raises(ValueError):
        nf.flows.PiecewiseRationalQuadraticCoupling(torch.ones(2, 2), net)
    with pytest.raises(ValueError):
        nf.flows.PiecewiseRationalQuadraticCoupling(torch.ones(0), net)
    with pytest.raises(RuntimeError):
        nf.flows.PiecewiseRationalQuadraticCoupling(torch.tensor([1, 0]), net, tails="cubic")
    with pytest.raises(RuntimeError):
        nf.utils.splines.unconstrained_rational_quadratic_spline(
            torch.zeros(2), torch.zeros(2, 4), torch.zeros(2, 4), torch.zeros(2, 3), tails="cubic")
    with pytest.raises(ValueError):
        nf.utils.splines.rational_quadratic_spline(
            torch.zeros(2), torch.zeros(2, 8), torch.zeros(2, 8), torch.zeros(2, 9), min_bin_width=0.2)
    m = nf.flows.PiecewiseRationalQuadraticCoupling(torch.tensor([1, 0, 1, 0]), net, tails="linear")
    with pytest.raises(ValueError):
        m(torch.zeros(3, 4, 2))           # rank not in {2, 4}
    with pytest.raises(ValueError):
        m(torch.zeros(3, 5))              # wrong feature count
    with pytest.raises(NotImplementedError):
        nf.flows.AffineCouplingBlock(nf.nets.MLP([2, 4, 4]), scale_map="tanh")._run(torch.zeros(2, 4), False)
    with pytest.raises(NotImplementedError):
        nf.flows.Permute(4, mode="reverse").gather_index(False)
    with pytest.raises(NotImplementedError):
        nf.flows.Split("rows").forward(torch.zeros(2, 4))


def test_no_cpu_fallback():
    """CPU tensors are refused; nothing silently computes off-device."""
    with torch.no_grad():
        with pytest.raises(nf.VcnfError):
            nf.flows.CoupledRationalQuadraticSpline(4, 1, 8).inverse(torch.zeros(2, 4))
        with pytest.raises(nf.VcnfError):
            nf.flows.AffineCouplingBlock(nf.nets.MLP([2, 4, 4])).forward(torch.zeros(2, 4))
        with pytest.raises(nf.VcnfError):
            nf.flows.Permute(4, "swap").forward(torch.zeros(2, 4))
        with pytest.raises(nf.VcnfError):
            nf.distributions.DiagGaussian(4).log_prob(torch.zeros(2, 4))
        with pytest.raises(nf.VcnfError):
            nf.utils.splines.unconstrained_rational_quadratic_spline(
                torch.zeros(2), torch.zeros(2, 4), torch.zeros(2, 4), torch.zeros(2, 3))


def test_product_does_not_import_oracle():
    """The package and the timing tools never import the oracle, the test helpers or a test module
    (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may)."""
    import os, re
    pkg = os.path.dirname(os.path.abspath(nf.__file__))
    repo = os.path.dirname(pkg)
    pat = re.compile(r"^\s*(from|import)\s+(oracle|helpers|test_\w+)\b", flags=re.M)
    for root in (pkg, os.path.join(repo, "profiles", "tools")):
        for dp, _, files in os.walk(root):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dp, f)).read()
                    assert not pat.search(src), os.path.join(dp, f)
    bench = open(os.path.join(repo, "bench.py")).read()
    body = bench.split("def cpu_baseline", 1)
    assert not pat.search(body[0]), "bench.py reaches the oracle outside cpu_baseline"


def test_shard_bounds_partition():
    for total in (0, 1, 7, 1 << 20, 1000003):
        for w in (1, 2, 3, 8):
            spans = [nf.shard_bounds(total, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        nf.shard_bounds(8, 2, 2)


# ---------------------------------------------------------------- next rows built from tensor ops
def test_checkerboard_split_merge_matches_reference_fixture():
    import vcnf_amd as nf
    from helpers import fixture, T
    fx = fixture("g15_checkerboard")
    for name in ("2d", "3d", "4d"):
        z = T(fx[name + "/z"])
        for mode in ("checkerboard", "checkerboard_inv"):
            (z1, z2), ld = nf.flows.Split(mode).forward(z)
            assert ld == 0
            assert np.array_equal(z1.numpy(), fx["%s/%s/z1" % (name, mode)])
            assert np.array_equal(z2.numpy(), fx["%s/%s/z2" % (name, mode)])
            back, _ = nf.flows.Merge(mode).forward([z1, z2])
            assert torch.equal(back, z)
            again, _ = nf.flows.Split(mode).inverse([z1, z2])
            assert torch.equal(again, z)
    with pytest.raises(ValueError):
        nf.flows.Split("checkerboard").forward(torch.zeros(2, 5))


@pytest.mark.parametrize("d", [5, 64])
def test_lu_linear_permute_matches_reference_fixture(d):
    """One GEMM per direction (permutation folded into L U / its fp64 inverse) against the
    reference's two triangular products / solves + index_select."""
    import vcnf_amd as nf
    from helpers import fixture, T, state_for, assert_close
    fx = fixture("g14_lu_linear_permute")
    tag = "d%d" % d
    sd, _ = state_for(fx, tag, 1401 + d, weight_gain=0.5)
    lay = nf.flows.LULinearPermute(d, identity_init=False)
    lay.load_state_dict(sd)
    x = T(fx[tag + "/x"])
    with torch.no_grad():
        for dirn, fn in (("fwd", lay.forward), ("inv", lay.inverse)):
            z, ld = fn(x)
            # noise of the reference's own fp32 run on this fixture bounds the comparison
            noise = float(np.abs(fx["%s/%s_z32" % (tag, dirn)] - fx["%s/%s_z64" % (tag, dirn)]).max())
            assert_close(z, fx["%s/%s_z64" % (tag, dirn)], what=dirn + " z", rtol=1e-5, atol=1e-5 + 4 * noise)
            assert_close(ld, fx["%s/%s_ld32" % (tag, dirn)], what=dirn + " ld", rtol=1e-6, atol=1e-6)
        assert lay.linear._mats, "matrices are cached when no gradient is required"
    z, ld = lay.inverse(x.clone().requires_grad_())
    (z.sum() + ld.sum()).backward()
    assert all(p.grad is not None for p in lay.parameters())


def test_sharded_evaluator_micro_batches():
    from vcnf_amd.sharded import ShardedEvaluator
    calls = []

    def fake_log_prob(x, ctx=None):
        calls.append(len(x))
        return x.sum(1) + (0 if ctx is None else ctx.sum(1))
    x, c = torch.randn(1000, 3), torch.randn(1000, 2)
    ev = ShardedEvaluator(fake_log_prob, micro_batch=256)
    got = ev.log_prob_shard(x, c)
    assert calls == [256, 256, 256, 232]
    assert torch.equal(got, x.sum(1) + c.sum(1))
    assert abs(float(ev.mean_log_prob(x)) - float(x.sum(1).double().mean())) < 1e-9


@pytest.mark.parametrize("tag", ["plain", "circular"])
def test_made_masks_and_state_keys_match_reference_fixture(tag):
    """MADE's masks and degrees (deterministic for permute_mask=False) and the state-dict key set of
    the autoregressive spline layers against the reference's (fixture G19)."""
    from helpers import fixture
    fx = fixture("g19_autoregressive")
    torch.manual_seed(0)
    if tag == "plain":
        lay = nf.flows.AutoregressiveRationalQuadraticSpline(6, 1, 32, num_bins=8, tail_bound=3.0)
    else:
        lay = nf.flows.CircularAutoregressiveRationalQuadraticSpline(
            6, 1, 32, ind_circ=[1, 4], num_bins=8, tail_bound=torch.tensor([3.0, np.pi, 3.0, 2.5, np.pi, 3.0]))
    ours = lay.state_dict()
    ref_float = {n for n, _ in synth.decode_entries(fx[tag + "/entries"])}
    ref_keys = ref_float | {k[len(tag) + 5:] for k in fx if k.startswith(tag + "/int/")} \
        | {k[len(tag) + 6:] for k in fx if k.startswith(tag + "/mask/")}
    extra = {k for k in ours if k.endswith("tail_bound") or k.endswith("preprocessing.scale")}
    assert set(ours) - extra == ref_keys
    if tag == "plain":                                   # no random permutation of the degrees
        for k in fx:
            if k.startswith("plain/mask/"):
                assert np.array_equal(ours[k[len("plain/mask/"):]].numpy(), fx[k]), k
            if k.startswith("plain/int/"):
                assert np.array_equal(ours[k[len("plain/int/"):]].numpy(), fx[k]), k


def test_counter_device_resolution(monkeypatch):
    """'cuda' without an index means the CURRENT device, not device 0 (a rank whose current device
    is not 0 must read its own counters)."""
    import torch
    from vcnf_amd import _lib
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    assert _lib.device_index("cuda") == 3
    assert _lib.device_index(torch.device("cuda")) == 3
    assert _lib.device_index("cuda:1") == 1
    assert _lib.device_index(torch.device("cuda", 0)) == 0
    import pytest
    with pytest.raises(_lib.VcnfError):
        _lib.device_index("cpu")


def test_refresh_packed_invalidates_every_cache():
    import torch
    import vcnf_amd as nf
    lay = nf.flows.CoupledRationalQuadraticSpline(8, 1, 16, 4)
    lay.prqct.__dict__['_fused_pack'] = {0: (("key",), torch.zeros(3)), 1: (("key",), torch.zeros(3))}   # one per matrix path
    lay.prqct.__dict__['_fused_final_pack'] = {'key': 1, 'buf': torch.zeros(2)}
    lu = nf.flows.LULinearPermute(8)
    lu.linear._mats[("k",)] = torch.zeros(1)
    blk = nf.flows.AffineCouplingBlock(nf.nets.MLP([4, 8, 8, 8]))
    blk.__dict__['_fused_affine_pack'] = {'key': 2, 'buf': torch.zeros(2)}
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(8), [lay, lu, blk])
    model.eval()                                            # train(False) runs refresh_packed
    for prec in (0, 1):
        assert lay.prqct.__dict__['_fused_pack'][prec][0] is None and lay.prqct.__dict__['_fused_pack'][prec][1] is not None
    assert lay.prqct.__dict__['_fused_final_pack']['key'] is None
    assert blk.__dict__['_fused_affine_pack']['key'] is None
    assert len(lu.linear._mats) == 0
    lay.prqct.__dict__['_fused_pack'] = {1: (("key",), torch.zeros(3))}
    model.load_state_dict(model.state_dict())
    assert lay.prqct.__dict__['_fused_pack'][1][0] is None
    # caches added in round 2: trunk pack, affine stack buffer + descriptors, memoised stack plans, the GlowBlock's
    # composed mixer, the conditioner's packed 1x1 convolution
    lay.prqct.__dict__['_fused_trunk_pack'] = {'key': 3, 'buf': torch.zeros(2)}
    blk.__dict__['_fused_affine_stack'] = {'key': 4, 'wpack': torch.zeros(2), 'desc': {True: ("k", 1)}}
    model.__dict__['_stack_plans'] = {("k",): None}
    glow = nf.flows.GlowBlock(8, 16)
    keep = (torch.zeros(2), torch.zeros(2), torch.zeros(()))
    glow.__dict__['_mix_cache'] = {True: {'key': 5, 'out': keep}, 'norm_ready': True}
    glow.flows[0].flows[1].param_map.__dict__['_fused_conv_pack'] = {'key': 6, 'buf': torch.zeros(2)}
    nf.refresh_packed(model)
    nf.refresh_packed(glow)
    assert lay.prqct.__dict__['_fused_trunk_pack']['key'] is None
    assert blk.__dict__['_fused_affine_stack']['key'] is None and 'desc' not in blk.__dict__['_fused_affine_stack']
    assert blk.__dict__['_fused_affine_stack']['wpack'] is not None           # buffers stay (rewritten in place later)
    assert model.__dict__['_stack_plans'] == {}
    assert glow.__dict__['_mix_cache'][True]['key'] is None and glow.__dict__['_mix_cache'][True]['out'] is keep
    assert glow.__dict__['_mix_cache']['norm_ready'] is False
    assert glow.flows[0].flows[1].param_map.__dict__['_fused_conv_pack']['key'] is None
