"""GEMM-level error of the two matrix arithmetics of the fused RQS layer kernels.

The reference's conditioner is plain fp32 nn.Linear (normflow/nets/resnet.py:78-106).  The fused layer kernel has an
exact-fp32 matrix path (v_mfma_f32_16x16x4_f32 chains) and the fp16 split-half path ("fp16x3": hi + lo fp16 halves
of both operands, hi*hi + (hi*lo + lo*hi) 2^-11 on v_mfma_f32_32x32x16_f16, fp32 accumulation).  The stack-level
parity tests are green on both, but that is parity on fixtures, not a statement about the arithmetic.  Here each dense
layer shape of config C3 - 48 -> 128 (first layer), 128 -> 128 (hidden layers, ReLU'd input), 16 -> 128 (context
gates), 128 -> 736 (last layer) - is evaluated by vcnf_linear_probe_f32, which runs exactly the instruction sequences
of the two kernels (same split function, same accumulation order), and judged against an fp64 product:

    "ulp of the result":  |y_path - y_64| / ulp32(|y_64|)             (median / mean / p99.9 / max)
    "ulp of the scale":   |y_path - y_64| / ulp32(sum_k |x_k w_k| + |b|)  (mean / p99.9 / max)

The first is the figure VERDICT r2 asks for; it is heavy-tailed (an output that cancels to ~0 has a tiny ulp: the mean is
carried by a few hundred of the ~1e7 outputs, the max is one draw), so the asserted comparison uses the second, the
standard normalisation of a dot product's rounding error, plus the median of the first.

Criterion (asserted): the split-half path's error is NO LARGER than the exact-fp32 MFMA path's - mean and 99.9th percentile
in ulp of the scale; median (+0.01) in ulp of the result; max in ulp of the scale within 1.25x (an extreme-value statistic)
- on the first (48-deep, lo*lo kept), hidden and last (128-deep) layers for both input families, and on the 16-deep gate
layer for p99.9 and max.  torch's own fp32 GEMM on the same device (what the reference's nn.Linear would run) is printed
beside them.  Measured on MI355X: on the 128-deep layers the split-half path has 0.6-0.75x the fp32 path's mean error (it
rounds the accumulator 8 times per output instead of 128 times), on the first layer 0.73x.

The gate layer (16 -> 128) is the exception that is stated rather than hidden: a 16-deep fp32 chain rounds only 16 times,
and the f16 matrix instruction aligns its 17 addends (16 products + the accumulator input) to the largest one and truncates
(the signed mean error of that layer is negative as long as the bias sits in the accumulator).  With the lo*lo term kept
(as the kernel does since this round) its mean error is 0.79x the fp32 path's on randn-scale inputs and 1.09x on
fixture-scale inputs (0.290 -> 0.254 against 0.233 ulp of the scale), its p99.9 0.79x / 0.68x, its max 0.85x / 0.92x.
Asserted for that layer: mean <= 1.15x, p99.9 and max <= 1.0x.  What the gate error does to the conditioner's OUTPUT is
asserted by test_conditioner_logits_error_split_vs_fp32: the 736 logits of a whole ResidualNet evaluated GEMM by GEMM
with each path's arithmetic (gate -> sigmoid -> product with the branch -> residual -> last layer) are closer to the fp64
logits on the split-half path than on the exact-fp32 path.
"""
import numpy as np
import pytest
import torch

import vcnf_amd as nf
from vcnf_amd import _lib

pytestmark = pytest.mark.gpu

B = 16384
SHAPES = [  # name, K, N, relu'd input, probe mode of the split path, input family
    ("first 48->128", 48, 128, False, _lib.PROBE_F16X3_LL),
    ("hidden 128->128", 128, 128, True, _lib.PROBE_F16X3),
    ("gate 16->128", 16, 128, False, _lib.PROBE_F16X3_LL),
    ("last 128->736", 128, 736, False, _lib.PROBE_F16X3),
]


def ulp32(v):
    """fp32 spacing at |v| (v fp64), floored at the spacing of the smallest normal."""
    a = np.maximum(np.abs(v), np.finfo(np.float32).tiny).astype(np.float32)
    return np.spacing(a).astype(np.float64)


def stats(y, ref64, scale64):
    err = np.abs(y.double().cpu().numpy() - ref64)
    e = err / ulp32(ref64)
    g = err / ulp32(scale64)
    return dict(r_med=float(np.median(e)), r_mean=float(e.mean()), r_p999=float(np.quantile(e, 0.999)), r_max=float(e.max()),
                s_mean=float(g.mean()), s_p999=float(np.quantile(g, 0.999)), s_max=float(g.max()))


def inputs(kind, k, n, relu, gen):
    """'fixture': the synthetic weights of the golden fixtures (tests/golden/synth.py: N(0, 1/fan_in) matrices, 0.1 N(0,1)
    biases) and unit-scale activations; 'randn': nn.Linear's own init range U(+-1/sqrt(K)) with heavier-tailed
    activations (N(0, 3^2), a few at 1e3) - the scale trained conditioners reach."""
    if kind == "fixture":
        w = torch.randn(n, k, generator=gen) / k ** 0.5
        b = 0.1 * torch.randn(n, generator=gen)
        x = torch.randn(B, k, generator=gen)
    else:
        w = (torch.rand(n, k, generator=gen) * 2 - 1) / k ** 0.5
        b = (torch.rand(n, generator=gen) * 2 - 1) / k ** 0.5
        x = 3.0 * torch.randn(B, k, generator=gen)
        x[::97, ::5] *= 300.0
    if relu:
        x = x + 0.3          # the hidden layers see relu(h): about two thirds of the units active
    return x, w, b


@pytest.mark.parametrize("kind", ["fixture", "randn"])
@pytest.mark.parametrize("name,k,n,relu,mode", SHAPES, ids=[s[0].split()[0] for s in SHAPES])
def test_split_half_gemm_error_not_above_fp32_mfma(hip, kind, name, k, n, relu, mode):
    gen = torch.Generator().manual_seed(7 * k + n + (kind == "randn"))
    x, w, b = inputs(kind, k, n, relu, gen)
    xr = x.clamp_min(0) if relu else x
    ref64 = (xr.double() @ w.double().t() + b.double()).numpy()
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    y32 = _lib.linear_probe(xd, wd, bd, _lib.PROBE_F32, relu)
    ysp = _lib.linear_probe(xd, wd, bd, mode, relu)
    ylib = torch.nn.functional.linear(xd.clamp_min(0) if relu else xd, wd, bd)
    torch.cuda.synchronize()
    assert _lib.check_saturation(hip, model=torch.nn.Identity()) == 0
    scale64 = (xr.double().abs() @ w.double().abs().t() + b.double().abs()).numpy()
    s32, ssp, slib = stats(y32, ref64, scale64), stats(ysp, ref64, scale64), stats(ylib, ref64, scale64)
    fmt = "%(r_med).3f / %(r_mean).2f / %(r_p999).1f / %(r_max).0f | %(s_mean).4f / %(s_p999).3f / %(s_max).3f"
    print("\n%-16s %-8s  ulp of the result (median / mean / p99.9 / max) | ulp of the scale (mean / p99.9 / max)\n"
          "    exact-fp32 MFMA  %s\n    split-half       %s\n    torch fp32 GEMM  %s" % (name, kind, fmt % s32, fmt % ssp, fmt % slib))
    what = "%s %s: " % (name, kind)
    gate = k == 16
    assert ssp["s_mean"] <= (1.15 if gate else 1.0) * s32["s_mean"], what + "mean error (ulp of the scale) of the split-half path above the fp32 MFMA path's"
    assert ssp["s_p999"] <= s32["s_p999"], what + "p99.9 error (ulp of the scale) of the split-half path above the fp32 MFMA path's"
    assert ssp["s_max"] <= (1.0 if gate else 1.25) * s32["s_max"], what + "max error (ulp of the scale)"
    assert ssp["r_med"] <= s32["r_med"] + (0.15 if gate else 0.01), what + "median error (ulp of the result)"


def _conditioner(x, ctx, sd, linear):
    """ResidualNet of config C3 (resnet.py:92-106, block :38-57) with every nn.Linear evaluated by ``linear``."""
    h = linear("initial_layer", torch.cat((x, ctx), dim=1), False)
    for i in range(2):
        t = linear("blocks.%d.linear_layers.0" % i, h, True)
        t = linear("blocks.%d.linear_layers.1" % i, t, True)
        g = linear("blocks.%d.context_layer" % i, ctx, False)
        h = h + t * torch.sigmoid(g)
    return linear("final_layer", h, False)


@pytest.mark.parametrize("gain", [1.0, 6.0], ids=["well-conditioned", "stress"])
def test_conditioner_logits_error_split_vs_fp32(hip, gain):
    """End to end through one conditioner: the logits [B, 32 * 23] that parameterise the splines.  Weights as in the golden
    fixtures (tests/golden/synth.py; final-layer gain 1 and 6), inputs N(0,1).  Every dense layer runs through the probe
    with the arithmetic of one matrix path (split-half: lo*lo kept in the first layer and the gates, as in the kernel);
    ReLU / sigmoid / product / residual in fp32 on the device for both.  Reference: the same network in fp64 on the host.
    Asserted: mean and p99.9 of |logit - logit64| are no larger on the split-half path than on the exact-fp32 path."""
    import synth
    shapes = [("initial_layer", 128, 48)] + [("blocks.%d.%s" % (i, n), 128, kk) for i in range(2)
                                             for n, kk in (("linear_layers.0", 128), ("linear_layers.1", 128), ("context_layer", 16))]
    shapes += [("final_layer", 736, 128)]
    ents = [(n + "." + leaf, (o, i) if leaf == "weight" else (o,)) for n, o, i in shapes for leaf in ("weight", "bias")]
    sd = synth.synth_state(ents, 77, final_gain=gain)
    g = torch.Generator().manual_seed(5)
    x, ctx = torch.randn(B, 32, generator=g), torch.randn(B, 16, generator=g)
    sd64 = {k: v.double() for k, v in sd.items()}
    ref = _conditioner(x.double(), ctx.double(), sd64,
                       lambda n, a, relu: torch.nn.functional.linear(a.clamp_min(0) if relu else a, sd64[n + ".weight"], sd64[n + ".bias"]))
    sdd = {k: v.cuda() for k, v in sd.items()}

    def path(split):
        def lin(n, a, relu):
            mode = _lib.PROBE_F32
            if split:
                mode = _lib.PROBE_F16X3_LL if ("initial" in n or "context" in n) else _lib.PROBE_F16X3
            return _lib.linear_probe(a, sdd[n + ".weight"], sdd[n + ".bias"], mode, relu)
        return _conditioner(x.cuda(), ctx.cuda(), sdd, lin)
    e32 = (path(False).double().cpu() - ref).abs()
    esp = (path(True).double().cpu() - ref).abs()
    rms = float(ref.pow(2).mean().sqrt())
    q = lambda e: (float(e.mean()) / rms, float(e.flatten()[::7].quantile(0.999)) / rms, float(e.max()) / rms)
    print("\nconditioner logits (gain %g, rms %.3f): |error| / rms  mean / p99.9 / max:  exact-fp32 path %.3e / %.3e / %.3e   "
          "split-half path %.3e / %.3e / %.3e" % ((gain, rms) + q(e32) + q(esp)))
    assert q(esp)[0] <= q(e32)[0] and q(esp)[1] <= q(e32)[1]


def test_probe_matches_fused_kernel_arithmetic(hip):
    """Sanity of the probe's fp32 mode: it stays within a few ulp of a plain fp32 fma chain (k ascending, accumulator
    started at the bias - the order in which fused_layer.hip::dense_block accumulates); the fraction of bit-identical
    entries is printed (how the matrix instruction rounds inside its four-deep k-step is not documented)."""
    gen = torch.Generator().manual_seed(3)
    x, w, b = inputs("fixture", 48, 128, False, gen)
    y = _lib.linear_probe(x.cuda(), w.cuda(), b.cuda(), _lib.PROBE_F32, False).cpu()
    acc = b.clone().unsqueeze(0).repeat(B, 1).numpy().astype(np.float32)
    xn, wn = x.numpy(), w.numpy()
    for kk in range(48):
        # fp32 fma, emulated exactly in fp64 (a product of two fp32 values is exact in fp64; the sum rounds once)
        acc = (xn[:, kk:kk + 1].astype(np.float64) * wn[None, :, kk].astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
    diff = np.abs(y.numpy() - acc)
    print("\nprobe fp32 mode vs emulated fma chain: max abs diff %.3g, exact in %.2f %% of entries" % (diff.max(), 100.0 * (diff == 0).mean()))
    assert diff.max() <= 4 * np.spacing(np.abs(acc).max())


@pytest.mark.parametrize("k,n", [(48, 128), (128, 128), (16, 128), (128, 736), (128, 48), (736, 128), (128, 20), (64, 128), (128, 96), (64, 96)])
def test_training_linear_f16x3_error_not_above_library_fp32(hip, k, n):
    """csrc/linear_f16x3.hip - nn.Linear's forward and input gradient on the training path at large batches (weights in
    their natural layout, split in registers) - against an fp64 product: mean and p99.9 error not above those of the
    library's fp32 GEMM on the same operands (the split-half path rounds the accumulator k / 16 times instead of k
    times); ragged batches (1 row, a partial tile), output widths that are not whole 32-row blocks (48, 20), a reduction
    longer than one staged chunk (736), and no clamping on these inputs."""
    gen = torch.Generator().manual_seed(100 + k + n)
    for Bt in (1, 63, 64 * 40 + 17):
        x = torch.randn(Bt, k, generator=gen).cuda()
        w = (torch.randn(n, k, generator=gen) / k ** 0.5).cuda()
        b = torch.randn(n, generator=gen).cuda()
        g = torch.randn(Bt, n, generator=gen).cuda()
        nf.check_saturation()
        cases = [("forward", _lib.linear_f16x3(x, w, b), x.double() @ w.double().t() + b.double(), torch.addmm(b, x, w.t()),
                  (x.abs().double() @ w.abs().double().t() + b.abs().double()))]
        if _lib.lib().vcnf_linear_f16x3_supported(n, k):          # the input gradient reduces over n
            cases.append(("dgrad", _lib.linear_f16x3(g, w, None, input_grad=True), g.double() @ w.double(), g @ w,
                          g.abs().double() @ w.abs().double()))
        for what, got, ref, lib, scale in cases:
            e_sp, e_lib = (got.double() - ref).abs(), (lib.double() - ref).abs()
            # every entry within a few fp32 roundings of the dot product's scale sum |x w| (any batch size) ...
            assert bool((e_sp <= 4 * 2.0 ** -24 * scale + 1e-30).all()), (Bt, what, float((e_sp / scale).max()))
            if Bt > 1000:                                          # ... and, with enough entries for a statistic, not above the library
                # (16-deep reductions: the fp32 chain rounds only 16 times - the split-half path is level with it, not
                # below: 0.8-1.1x, the gate-layer finding of test_split_half_gemm_error_not_above_fp32_mfma)
                depth = k if what == "forward" else n
                assert float(e_sp.mean()) <= (1.15 if depth <= 16 else 1.05) * float(e_lib.mean()), \
                    (what, float(e_sp.mean()), float(e_lib.mean()))
                q = lambda e: float(e.flatten()[::3].quantile(0.999))
                assert q(e_sp) <= 1.1 * q(e_lib), (what, q(e_sp), q(e_lib))
        assert nf.check_saturation() == 0
    # ReLU on the input while it is read / on the result before it is stored (the residual block's activations ride on the
    # kernel): exactly the composition with torch.relu
    xr = torch.randn(200, k, generator=gen).cuda()
    assert torch.equal(_lib.linear_f16x3(xr, w, b, relu_in=True), _lib.linear_f16x3(torch.relu(xr), w, b))
    assert torch.equal(_lib.linear_f16x3(xr, w, b, relu_out=True), torch.relu(_lib.linear_f16x3(xr, w, b)))
    if _lib.lib().vcnf_linear_f16x3_supported(n, k):
        # ... and on an input gradient: result * (mask > 0) + addend as it is stored
        gr, mk, ad = (torch.randn(200, m, generator=gen).cuda() for m in (n, k, k))
        want = _lib.linear_f16x3(gr, w, None, input_grad=True) * (mk > 0) + ad
        assert torch.equal(_lib.linear_f16x3(gr, w, None, input_grad=True, mask=mk, addend=ad), want)
    nf.check_saturation()
    # a value beyond the fp16 range is clamped and counted, never silently
    x = torch.randn(64, k, generator=gen).cuda()
    x[5, 3] = 1.0e6
    _lib.linear_f16x3(x, w, b)
    with pytest.raises(nf.VcnfError):
        nf.check_saturation()
