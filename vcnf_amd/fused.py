"""Host side of the fused RQS-coupling-layer kernel (csrc/fused_layer.hip): decides
whether a layer qualifies, packs the ResidualNet weights into matrix-core fragment
order (once per weight version, cached on the module) and launches the kernel.

Fragment order (v_mfma_f32_16x16x4_f32, weights as the A operand): for output row
block nb and k-step s the 64 lanes hold W[16 nb + (lane & 15)][k(s, lane >> 4)].
Layers fed from the input tile use natural k = 4 s + q; layers fed from the previous
layer's accumulators use k = 16 (s >> 2) + 4 q + (s & 3), the order in which those
accumulators sit in registers.  Four consecutive k-steps are interleaved per lane so
that one 16-byte load feeds four matrix instructions.  The last layer's rows are
permuted so that lane group q receives the 3K-1 logits of feature 4 g + q.
"""
import os

import torch
from torch import nn
from torch.nn import functional as F

from . import _lib


def _frag_index(n_rows_blocks, n_s4, chained, device):
    nb = torch.arange(n_rows_blocks, device=device).view(-1, 1, 1, 1)
    s4 = torch.arange(n_s4, device=device).view(1, -1, 1, 1)
    lane = torch.arange(64, device=device).view(1, 1, -1, 1)
    c = torch.arange(4, device=device).view(1, 1, 1, -1)
    rows = 16 * nb + (lane & 15) + 0 * s4 + 0 * c
    q = lane >> 4
    cols = (16 * s4 + 4 * q + c) if chained else (4 * (4 * s4 + c) + q)
    return rows.expand(n_rows_blocks, n_s4, 64, 4), (cols + 0 * nb).expand(n_rows_blocks, n_s4, 64, 4)


def _pack_dense(weight, chained):
    n, k = weight.shape
    rows, cols = _frag_index(n // 16, k // 16, chained, weight.device)
    return weight[rows, cols].reshape(-1)


def _pack_final(weight, bias, d_t, p):
    """weight [d_t * p, H] -> fragments [g][b][s4][lane][c] with the row permutation
    (g, b, lane) -> feature 4 g + ((lane & 15) >> 2), logit 4 b + (lane & 3); rows that
    pad p up to a multiple of 4 are zero."""
    h = weight.shape[1]
    p4 = (p + 3) // 4
    dev = weight.device
    g = torch.arange(d_t // 4, device=dev).view(-1, 1, 1, 1, 1)
    b = torch.arange(p4, device=dev).view(1, -1, 1, 1, 1)
    s4 = torch.arange(h // 16, device=dev).view(1, 1, -1, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 1, 1, -1, 1)
    c = torch.arange(4, device=dev).view(1, 1, 1, 1, -1)
    i = lane & 15
    feat = 4 * g + (i >> 2)
    t = 4 * b + (i & 3)
    rows = feat * p + t
    cols = 16 * s4 + 4 * (lane >> 4) + c
    shape = (d_t // 4, p4, h // 16, 64, 4)
    valid = (t < p).expand(shape)
    rows = torch.where(t < p, rows, torch.zeros_like(rows)).expand(shape)
    wf = torch.where(valid, weight[rows, cols.expand(shape)], torch.zeros((), device=dev, dtype=weight.dtype))
    # bias [g][q][4 * p4]
    g2 = torch.arange(d_t // 4, device=dev).view(-1, 1, 1)
    q2 = torch.arange(4, device=dev).view(1, -1, 1)
    t2 = torch.arange(4 * p4, device=dev).view(1, 1, -1)
    idx = (4 * g2 + q2) * p + t2
    ok = (t2 < p).expand(d_t // 4, 4, 4 * p4)
    bf = torch.where(ok, bias[torch.where(ok, idx, torch.zeros_like(idx))], torch.zeros((), device=dev, dtype=bias.dtype))
    return wf.reshape(-1), bf.reshape(-1)


PREC_F32, PREC_F16X3 = 0, 1
LO_SCALE = 2048.0
# Matrix path used when a module does not set ``fused_precision`` itself.  'fp16x3' (fp16 split-half operands, fp32
# accumulation) is the default because (a) GEMM by GEMM its error against an fp64 product is at or below the exact
# fp32 matrix path's on every layer shape (tests/test_gpu_gemm_error.py) and (b) the fused RQS layer is range-safe
# on the device: what the fp16 halves cannot carry is evaluated by the exact fp32 kernel, never clamped (run()).
# VCNF_FUSED_PRECISION=fp32 / ``fused_precision = 'fp32'`` select exact fp32 matrix instructions everywhere
# (also for the Glow conv conditioner, whose split-half kernels clamp and count instead: nf.check_saturation()).
DEFAULT_PRECISION = os.environ.get('VCNF_FUSED_PRECISION', 'fp16x3')


def _split_halves(w):
    """fp32 -> (hi, lo) fp16 with w ~= hi + lo / 2048 (22 significant bits)."""
    hi = w.half()
    lo = ((w - hi.float()) * LO_SCALE).half()
    return hi, lo


def _as_floats(h):
    return h.contiguous().view(torch.float32).reshape(-1)


LOG2E = 1.4426950408889634


def _hi_lo_fragments(w):
    """[..., 64, 8] fp32 fragment values -> floats of [..., hi | lo, 64, 8] fp16 (two halves per float)."""
    hi, lo = _split_halves(w)
    return _as_floats(torch.stack([hi, lo], dim=-3))


def _afrag_cols(k_total, chained, device):
    """k index of slot (lane half kg, i) of k-step t for v_mfma_f32_32x32x16_f16 operands: [t, 64, 8].
    chained: the operand is the previous layer's accumulator (register r of lane half kg is row
    8 (r / 4) + 4 kg + r % 4 of its 32-row block, registers 8 hh .. 8 hh + 7 feed k-step 2 nb + hh);
    otherwise the natural order 16 t + 8 kg + i of an input vector."""
    t = torch.arange(k_total // 16, device=device).view(-1, 1, 1)
    kg = (torch.arange(64, device=device) >> 5).view(1, -1, 1)
    i = torch.arange(8, device=device).view(1, 1, -1)
    cols = (16 * t + 8 * (i >> 2) + 4 * kg + (i & 3)) if chained else (16 * t + 8 * kg + i)
    return cols.expand(k_total // 16, 64, 8)


def _pack_dense6(weight, chained):
    """[N, K] -> A fragments [N / 32][K / 16][hi | lo][lane][8]: lane l holds row 32 nb + l % 32."""
    n, k = weight.shape
    dev = weight.device
    cols = _afrag_cols(k, chained, dev)
    rows = (32 * torch.arange(n // 32, device=dev).view(-1, 1, 1, 1) +
            (torch.arange(64, device=dev) & 31).view(1, 1, -1, 1)).expand(n // 32, k // 16, 64, 8)
    return _hi_lo_fragments(weight[rows, cols.unsqueeze(0).expand(n // 32, -1, -1, -1)])


def _pack_bias6(bias):
    """[N] -> accumulator order [N / 32][lane half][16]: register r of lane half kg is row 8 (r / 4) + 4 kg + r % 4."""
    dev = bias.device
    nb = torch.arange(bias.numel() // 32, device=dev).view(-1, 1, 1)
    kg = torch.arange(2, device=dev).view(1, -1, 1)
    r = torch.arange(16, device=dev).view(1, 1, -1)
    return bias[32 * nb + 8 * (r >> 2) + 4 * kg + (r & 3)].reshape(-1)


def _pack_final6(weight, bias, d_t, p, wh_scale, num_bins):
    """Last layer for the fp16 split-half kernel.  Rows are permuted so that after three 32-row blocks the lane
    half kg of a column holds the p logits of features 4 g + 2 kg + {0, 1} in accumulator entries 24 f2 + tl
    (entry v = register v % 16 of block v / 16; tl = p is a zero row), and scaled: width / height logits by
    wh_scale * log2(e), derivative logits by log2(e) (csrc/rqs_lean.hpp evaluates 2^x)."""
    h = weight.shape[1]
    dev = weight.device
    scale = torch.where(torch.arange(p, device=dev) < 2 * num_bins, wh_scale * LOG2E, LOG2E).double()
    w = (weight.double().view(d_t, p, h) * scale.view(1, p, 1)).float()
    bsc = (bias.double().view(d_t, p) * scale.view(1, p)).float()
    w = torch.cat([w, torch.zeros(d_t, 1, h, device=dev)], dim=1)            # [d_t, 24, h], logit 23 = 0
    bsc = torch.cat([bsc, torch.zeros(d_t, 1, device=dev)], dim=1)
    g = torch.arange(d_t // 4, device=dev).view(-1, 1, 1)
    b = torch.arange(3, device=dev).view(1, -1, 1)
    i = torch.arange(32, device=dev).view(1, 1, -1)                          # row inside the 32-row block
    v = 16 * b + 4 * (i >> 3) + (i & 3)                                      # accumulator entry of that row
    feat = 4 * g + 2 * ((i >> 2) & 1) + v // 24
    wperm = w[feat, (v % 24).expand_as(feat)].reshape(-1, h)                 # [(d_t / 4) * 96, h]
    kg = torch.arange(2, device=dev).view(1, -1, 1)
    v2 = torch.arange(48, device=dev).view(1, 1, -1)
    bf = bsc[4 * g + 2 * kg + v2 // 24, (v2 % 24).expand(d_t // 4, 2, 48)]
    return _pack_dense6(wperm, True), bf.reshape(-1)


def pack_layer_h3(net, d_t, p, wh_scale, num_bins):
    """Flat fp32 buffer in the layout of csrc/fused_common.hpp::PackLayout6 (fp16 split-half kernel): every
    matrix as hi | lo fp16 A fragments of v_mfma_f32_32x32x16_f16, every bias in accumulator order.  Folded in:
    the logit scale and log2(e) of the spline's exponentials (last layer, see _pack_final6) and log2(e) of the
    gate sigmoid (context layers)."""
    parts = [_pack_dense6(net.initial_layer.weight, False), _pack_bias6(net.initial_layer.bias)]
    for blk in net.blocks:
        parts += [_pack_dense6(blk.linear_layers[0].weight, True), _pack_bias6(blk.linear_layers[0].bias),
                  _pack_dense6(blk.linear_layers[1].weight, True), _pack_bias6(blk.linear_layers[1].bias)]
        if net.context_features:
            wc = (blk.context_layer.weight.double() * LOG2E).float()
            bc = (blk.context_layer.bias.double() * LOG2E).float()
            parts += [_pack_dense6(wc, False), _pack_bias6(bc)]
    wf, bf = _pack_final6(net.final_layer.weight, net.final_layer.bias, d_t, p, wh_scale, num_bins)
    parts += [wf, bf]
    return torch.cat([t.detach().reshape(-1).float() for t in parts]).contiguous()


def pack_layer(net, d_t, p):
    """Flat fp32 buffer in the layout of csrc/fused_common.hpp::PackLayout (exact fp32 matrix path)."""
    parts = [_pack_dense(net.initial_layer.weight, False), net.initial_layer.bias]
    for blk in net.blocks:
        parts += [_pack_dense(blk.linear_layers[0].weight, True), blk.linear_layers[0].bias,
                  _pack_dense(blk.linear_layers[1].weight, True), blk.linear_layers[1].bias]
        if net.context_features:
            parts += [_pack_dense(blk.context_layer.weight, False), blk.context_layer.bias]
    wf, bf = _pack_final(net.final_layer.weight, net.final_layer.bias, d_t, p)
    parts += [wf, bf]
    return torch.cat([t.detach().reshape(-1).float() for t in parts]).contiguous()


def _is_relu(act):
    return act is F.relu or act is torch.relu or isinstance(act, nn.ReLU)


def eligible(coupling, context):
    """Layer shapes and conditioner structure the fused kernel implements."""
    net = coupling.transform_net
    from .nets.resnet import ResidualNet
    if type(net) is not ResidualNet or net.preprocessing is not None:
        return False
    if coupling.tails != 'linear':
        return False
    blocks = list(net.blocks)
    if any(b.use_batch_norm or not _is_relu(b.activation) or (b.dropout.p > 0 and b.training) for b in blocks):
        return False
    ctx_dim = net.context_features or 0
    if (context is None) != (ctx_dim == 0):
        return False
    if context is not None and (context.dim() != 2 or context.shape[1] != ctx_dim):
        return False
    return bool(_lib.lib().vcnf_rqs_layer_fused_supported(
        coupling.num_identity_features, coupling.num_transform_features, ctx_dim,
        net.hidden_features, len(blocks), coupling.num_bins, _lib.TAILS_LINEAR))


def precision_of(coupling):
    mode = getattr(coupling, 'fused_precision', None) or DEFAULT_PRECISION
    if mode not in ('fp32', 'fp16x3'):
        raise ValueError("fused_precision must be 'fp32' or 'fp16x3'")
    return PREC_F16X3 if mode == 'fp16x3' else PREC_F32


def _param_key(coupling):
    """(data_ptr, _version) of every conditioner parameter.  ``net.parameters()`` walks the module tree on every call
    (half of the host time of an eager log_prob at small batches); the walk is done once and remembered as
    (owner module, name) slots, which are read through ``_parameters`` - a re-assigned Parameter is still seen;
    refresh_packed() forgets the slots (needed only after sub-MODULES were replaced)."""
    slots = coupling.__dict__.get('_fused_slots')
    if slots is None:
        slots = [(mod, name) for mod in coupling.transform_net.modules() for name in mod._parameters
                 if mod._parameters[name] is not None]
        coupling.__dict__['_fused_slots'] = slots
    key = []
    for mod, name in slots:
        p = mod._parameters[name]
        key.append((p.data_ptr(), p._version))
    return tuple(key)


def packed_weights(coupling, prec=None, key=None):
    """Cached packed buffer of one matrix path; refreshed when any conditioner parameter changed.  A refresh of the
    same size rewrites the existing device buffer in place, so a captured HIP graph that holds its address
    (vcnf_amd.graphs) sees the new weights."""
    net = coupling.transform_net
    prec = precision_of(coupling) if prec is None else prec
    if key is None:
        key = _param_key(coupling)
    packs = coupling.__dict__.setdefault('_fused_pack', {})
    cache = packs.get(prec)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            if prec == PREC_F16X3:
                buf = pack_layer_h3(net, coupling.num_transform_features, coupling._transform_dim_multiplier(),
                                    coupling._cfg(True).wh_scale, coupling.num_bins)
            else:
                buf = pack_layer(net, coupling.num_transform_features, coupling._transform_dim_multiplier())
            if cache is not None and cache[1].shape == buf.shape and cache[1].device == buf.device:
                cache[1].copy_(buf)
                buf = cache[1]
        cache = (key, buf)
        packs[prec] = cache
    return cache[1]


def run(coupling, inputs, context, sampling, log_q=None, sign=1.0):
    """One launch of the fused layer kernel on the coupling's matrix path.  'fp16x3' is range-safe: tiles holding a
    non-finite input or a value beyond the fp16 range are evaluated by the exact fp32 kernel in a second launch that
    otherwise returns at once (_lib.rqs_layer_fused); ``coupling.range_safe = False`` drops that launch (the
    split-half kernel then clamps at +-65504 and counts, nf.check_saturation())."""
    net = coupling.transform_net
    shared = coupling.unconditional_transform.logits() if coupling.unconditional_transform is not None else None
    prec = precision_of(coupling)
    safe = prec == PREC_F16X3 and getattr(coupling, 'range_safe', True)
    key = _param_key(coupling)
    return _lib.rqs_layer_fused(inputs, context, coupling._index32('tf'), coupling._index32('id'),
                                net.context_features or 0, net.hidden_features, len(net.blocks),
                                prec, packed_weights(coupling, prec, key), shared, coupling._cfg(True), sampling,
                                logdet=log_q, sign=sign,
                                wpack_f32=packed_weights(coupling, PREC_F32, key) if safe else None)


# ---------------------------------------------------------------- runs of layers in one launch (small batches)
def _stack_sig(coupling, context):
    """What must agree between the layers of one launch: shape, conditioner structure, spline configuration, matrix
    path, range policy; None for a layer the fused kernels do not cover."""
    if not (coupling.fused and eligible(coupling, context)):
        return None
    net = coupling.transform_net
    cfg = coupling._cfg(True)
    return (coupling.num_identity_features, coupling.num_transform_features, net.context_features or 0,
            net.hidden_features, len(net.blocks), precision_of(coupling), bool(getattr(coupling, 'range_safe', True)),
            coupling.unconditional_transform is not None,
            tuple(getattr(cfg, f) for f, _ in cfg._fields_))


def plan_stack(order, start, z, context):
    """Longest run order[start:end] (at least two, at most the kernel's limit) of CoupledRationalQuadraticSpline layers
    that one launch of vcnf_rqs_stack_fused_f32 evaluates, or None: 2-D fp32 inputs on the device, no gradient
    required, a batch small enough for the 32-sample-tile kernel, every layer fused-eligible with ONE shape, spline
    configuration and matrix path.  Returns (end, couplings, signature)."""
    from .flows.neural_spline.wrapper import CoupledRationalQuadraticSpline
    if z.dim() != 2 or z.dtype != torch.float32 or not z.is_cuda or z.shape[0] > _lib.small_batch_rows():
        return None
    lim = int(_lib.lib().vcnf_rqs_stack_fused_max_layers())
    run, sig = [], None
    for flow in order[start:start + lim]:
        if type(flow) is not CoupledRationalQuadraticSpline:
            break
        cp = flow.prqct
        if cp.per_feature or cp._needs_grad(z, context):
            break
        s = _stack_sig(cp, context)
        if s is None or (sig is not None and s != sig):
            break
        sig = s
        run.append(cp)
    if len(run) < 2:
        return None
    return start + len(run), run, sig


def _stack_stamp(cp):
    return (cp.fused, cp.fused_precision, getattr(cp, 'range_safe', True), cp.tails, cp.num_bins, cp.per_feature,
            cp.unconditional_transform is None, id(cp.transform_net), cp.tail_bound if not torch.is_tensor(cp.tail_bound) else id(cp.tail_bound))


def cached_plan_stack(owner, order, start, z, context):
    """plan_stack memoised on the calling model for evaluations without autograd (the plan costs a library call and a
    spline configuration per layer - more than the launch it saves at these batch sizes).  A cached plan is reused
    while the run's modules and their routing switches are what they were; shape, dtype, device, the batch-size
    threshold and the context width are part of the key."""
    if torch.is_grad_enabled():
        return plan_stack(order, start, z, context)
    if z.dim() != 2 or z.dtype != torch.float32 or not z.is_cuda:
        return None
    small = z.shape[0] <= _lib.small_batch_rows()
    key = (start, len(order), id(order[start]), z.shape[1], small, None if context is None else tuple(context.shape[1:]),
           DEFAULT_PRECISION)
    plans = owner.__dict__.setdefault('_rqs_stack_plans', {})
    hit = plans.get(key)
    if hit is not None:
        plan, mods, stamp = hit
        if all(a is b for a, b in zip(order[start:start + len(mods)], mods)) and \
                stamp == tuple(_stack_stamp(f.prqct) for f in mods if hasattr(f, 'prqct')):
            return plan
    plan = plan_stack(order, start, z, context)
    from .flows.neural_spline.wrapper import CoupledRationalQuadraticSpline
    lim = int(_lib.lib().vcnf_rqs_stack_fused_max_layers())
    # what the plan looked at: the run and the flow that ended it
    n = (plan[0] - start if plan is not None else 0) + 1
    mods = [f for f in order[start:start + min(n, lim)] if type(f) is CoupledRationalQuadraticSpline]
    if len(plans) > 64:
        plans.clear()
    plans[key] = (plan, mods, tuple(_stack_stamp(f.prqct) for f in mods))
    return plan


def run_stack(run, sig, z, context, sampling, log_q, sign):
    """Execute a planned run in one launch (plus the fp32 re-evaluation launch of the range-safe split-half path, which
    returns at once unless a tile was flagged).  The ctypes layer table is cached on the first coupling of the run and
    refilled with the current addresses on every call (packed buffers are rewritten in place, so they rarely change)."""
    d_id, d_t, ctx_dim, hidden, blocks, prec, safe = sig[:7]
    first = run[0]
    cache = first.__dict__.setdefault('_fused_rqs_stack', {})
    ids = (tuple(id(c) for c in run), bool(sampling))
    if cache.get('ids') != ids:
        cache.clear()
        cache['ids'] = ids
        cache['tab'] = (_lib.RqsStackLayer * len(run))()
        cache['tab32'] = (_lib.RqsStackLayer * len(run))()
    want32 = prec == PREC_F16X3 and safe
    tab, tab32 = cache['tab'], cache['tab32']
    keep = []                                   # the tensors behind the table's addresses, alive through the launch
    for i, cp in enumerate(run):
        key = _param_key(cp)
        wp = packed_weights(cp, prec, key)
        tf, idx = cp._index32('tf'), cp._index32('id')
        shared = cp.unconditional_transform.logits() if cp.unconditional_transform is not None else (None, None, None)
        keep.append((wp, tf, idx, shared))
        for t, w in ((tab, wp),) + (((tab32, packed_weights(cp, PREC_F32, key)),) if want32 else ()):
            e = t[i]
            e.transform_idx, e.identity_idx, e.wpack = tf.data_ptr(), idx.data_ptr(), w.data_ptr()
            e.shared_w = shared[0].data_ptr() if shared[0] is not None else None
            e.shared_h = shared[1].data_ptr() if shared[1] is not None else None
            e.shared_d = shared[2].data_ptr() if shared[2] is not None else None
            keep.append(w)
    out = _lib.rqs_stack_fused(z, context, tab, tab32 if want32 else None, keep[0][0].numel(), d_t, d_id, ctx_dim, hidden,
                               blocks, prec, first._cfg(True), sampling, logdet=log_q, sign=sign)
    del keep
    return out


def refresh_packed(module):
    """Invalidate every packed / derived weight cache under ``module``: the fused kernels read re-ordered copies of
    the conditioner weights (this file, fused_affine.py, fused_final.py) and ``_LULinear`` a cached L.U product,
    all keyed on (data_ptr, _version) of the parameters (likewise a GlowBlock's composed 1x1 convolution + ActNorm map).  An in-place update through ``.data`` (``p.data.copy_(ema)``,
    initialisation on ``.data`` - an idiom of the reference API) does NOT bump ``_version``: call this afterwards.
    ``NormalizingFlow`` / ``MultiscaleFlow`` call it on train() / eval() transitions and after load_state_dict().
    Buffers are rewritten in place at the next use, so a captured HIP graph keeps its addresses."""
    for m in module.modules():
        d = m.__dict__
        d.pop('_fused_slots', None)
        d.pop('_init_done_host', None)              # ActNorm: re-read the device flag (a loaded state dict may reset it)
        d.pop('_masked_stack', None)
        d.pop('_masked_stack_plans', None)
        d.pop('_fused_rqs_stack', None)
        d.pop('_rqs_stack_plans', None)
        if isinstance(d.get('_fused_pack'), dict):
            for prec, (key, buf) in list(d['_fused_pack'].items()):
                d['_fused_pack'][prec] = (None, buf)
        for name in ('_fused_affine_pack', '_fused_affine_stack', '_fused_final_pack', '_fused_trunk_pack', '_fused_conv_pack',
                     '_fused_conv3_pack', '_fused_taps_pack'):
            if isinstance(d.get(name), dict):
                d[name]['key'] = None
                d[name].pop('desc', None)                   # stack launch descriptors (permutation rows)
        if isinstance(d.get('_stack_plans'), dict):         # NormalizingFlow: memoised stack plans
            d['_stack_plans'].clear()
        if isinstance(d.get('_mats'), dict):
            d['_mats'].clear()
        if isinstance(d.get('_mix_cache'), dict):           # GlowBlock: composed 1x1 convolution + ActNorm per direction
            for k, v in d['_mix_cache'].items():
                if isinstance(v, dict):
                    v['key'] = None
            d['_mix_cache']['norm_ready'] = False
    return module
