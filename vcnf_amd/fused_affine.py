"""Host side of the fused affine-coupling-layer kernel (csrc/fused_affine.hip): eligibility,
weight packing (matrix-core fragment order, cached per parameter version) and launch.

Fragment order (v_mfma_f32_16x16x4_f32, weights as the A operand): for output row block nb and a
group of four k-steps the 64 lanes hold float4 W[16 nb + (lane & 15)][k(step, lane >> 4)].  The
first layer reads the input tile, k = 4 step + q; the other two consume the previous layer's
accumulators, k = 16 pb + 4 q + r for step 4 pb + r.  Biases are stored per row block as
[q][r] -> bias[16 nb + 4 q + r], the accumulator row order.
"""
import torch
from torch import nn

from . import _lib


def _linears(mlp):
    """(linears, slope) when ``mlp`` is Linear, LeakyReLU, Linear, LeakyReLU, Linear; else None.  Cached on the module
    (keyed on the identity of its layers): this runs for every layer of every evaluation."""
    net = getattr(mlp, 'net', None)
    key = tuple(map(id, net)) if isinstance(net, nn.Sequential) else None
    hit = mlp.__dict__.get('_fa_linears')
    if hit is not None and hit[0] == key:
        return hit[1]
    got = _linears_uncached(mlp)
    mlp.__dict__['_fa_linears'] = (key, got)
    return got


def _linears_uncached(mlp):
    from .nets.mlp import MLP
    if type(mlp) is not MLP:
        return None
    mods = list(mlp.net)
    if len(mods) != 5 or not all(isinstance(mods[i], nn.Linear) for i in (0, 2, 4)):
        return None
    if not all(isinstance(mods[i], nn.LeakyReLU) for i in (1, 3)) or mods[1].negative_slope != mods[3].negative_slope:
        return None
    return [mods[0], mods[2], mods[4]], float(mods[1].negative_slope)


def eligible(block, z):
    if z.dim() != 2 or z.dtype != torch.float32 or block.split_mode not in ('channel', 'channel_inv'):
        return False
    core = block.flows[1]
    got = _linears(core.param_map)
    if got is None:
        return False
    (l1, l2, l3), _ = got
    if l1.out_features != l2.in_features or l2.out_features != l1.out_features or l3.in_features != l2.out_features:
        return False
    return bool(_lib.lib().vcnf_affine_layer_fused_supported(l1.in_features, l1.out_features, l3.out_features,
                                                             z.shape[1]))


def _pack_first(w, hb, kig):
    dev = w.device
    nb = torch.arange(hb, device=dev).view(-1, 1, 1, 1)
    g = torch.arange(kig, device=dev).view(1, -1, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 1, -1, 1)
    i = torch.arange(4, device=dev).view(1, 1, 1, -1)
    rows = (16 * nb + (lane & 15)).expand(hb, kig, 64, 4)
    cols = (4 * (4 * g + i) + (lane >> 4)).expand(hb, kig, 64, 4)
    ok = cols < w.shape[1]
    vals = w[rows, torch.where(ok, cols, torch.zeros_like(cols))]
    return torch.where(ok, vals, torch.zeros((), device=dev, dtype=w.dtype)).reshape(-1)


def _pack_chained(w, nb_out, hb):
    """w [n_out, 16 hb] (rows beyond n_out are zero) -> [nb][pb][lane][r]."""
    dev = w.device
    nb = torch.arange(nb_out, device=dev).view(-1, 1, 1, 1)
    pb = torch.arange(hb, device=dev).view(1, -1, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 1, -1, 1)
    r = torch.arange(4, device=dev).view(1, 1, 1, -1)
    rows = (16 * nb + (lane & 15)).expand(nb_out, hb, 64, 4)
    cols = (16 * pb + 4 * (lane >> 4) + r).expand(nb_out, hb, 64, 4)
    ok = rows < w.shape[0]
    vals = w[torch.where(ok, rows, torch.zeros_like(rows)), cols]
    return torch.where(ok, vals, torch.zeros((), device=dev, dtype=w.dtype)).reshape(-1)


def _pack_bias(b, blocks):
    out = torch.zeros(16 * blocks, dtype=b.dtype, device=b.device)
    out[:b.numel()] = b
    return out


def pack(l1, l2, l3):
    hb = l1.out_features // 16
    kig = 1 if l1.in_features <= 16 else 4
    ob = (l3.out_features + 15) // 16
    parts = [_pack_first(l1.weight.detach(), hb, kig), _pack_bias(l1.bias.detach(), hb),
             _pack_chained(l2.weight.detach(), hb, hb), _pack_bias(l2.bias.detach(), hb),
             _pack_chained(l3.weight.detach(), ob, hb), _pack_bias(l3.bias.detach(), ob)]
    buf = torch.cat(parts).contiguous()
    want = int(_lib.lib().vcnf_affine_layer_fused_pack_floats(l1.in_features, l1.out_features, l3.out_features))
    assert buf.numel() == want, (buf.numel(), want)
    return buf


def _pack_h3(w, nb_out, hb):
    """w [n_out, 16 hb] -> hi | lo fp16 A fragments of v_mfma_f32_16x16x32_f16, [nb][t][hi | lo][lane][8] as floats: lane l
    holds row 16 nb + l % 16 and, for q = l / 16, the eight k-slots of k-step t = units 32 t + 4 q + i (i < 4) and
    32 t + 16 + 4 q + (i - 4): rows 4 q .. 4 q + 3 of the previous layer's row blocks 2 t and 2 t + 1, i.e. that layer's
    accumulators as they sit in the lane (csrc/fused_affine.hip::affine_layer_on_strip_h3).  Rows beyond n_out are zero."""
    from .fused import _split_halves, _as_floats
    dev = w.device
    nt = hb // 2
    nb = torch.arange(nb_out, device=dev).view(-1, 1, 1, 1)
    t = torch.arange(nt, device=dev).view(1, -1, 1, 1)
    lane = torch.arange(64, device=dev).view(1, 1, -1, 1)
    i = torch.arange(8, device=dev).view(1, 1, 1, -1)
    shape = (nb_out, nt, 64, 8)
    rows = (16 * nb + (lane & 15)).expand(shape)
    cols = (32 * t + 16 * (i >> 2) + 4 * (lane >> 4) + (i & 3)).expand(shape)
    ok = rows < w.shape[0]
    vals = torch.where(ok, w[torch.where(ok, rows, torch.zeros_like(rows)), cols], torch.zeros((), device=dev, dtype=w.dtype))
    hi, lo = _split_halves(vals)
    return _as_floats(torch.stack([hi, lo], dim=2))           # [nb, t, 2, 64, 8] halves


def pack_h3(l2, l3):
    hb = l2.out_features // 16
    ob = (l3.out_features + 15) // 16
    return torch.cat([_pack_h3(l2.weight.detach(), hb, hb), _pack_h3(l3.weight.detach(), ob, hb)]).contiguous()


def h3_ok(block):
    """The split-half form of the stack kernel: hidden width a multiple of 32 and the block's matrix path 'fp16x3'
    (vcnf_amd.fused.DEFAULT_PRECISION / ``block.fused_precision``)."""
    from . import fused
    (l1, l2, l3), _ = _linears(block.flows[1].param_map)
    return l1.out_features % 32 == 0 and fused.precision_of(block) == fused.PREC_F16X3


def packed_weights(block):
    (l1, l2, l3), slope = _linears(block.flows[1].param_map)
    params = (l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias)
    key = tuple((p.data_ptr(), p._version, str(p.device)) for p in params)
    cache = block.__dict__.setdefault('_fused_affine_pack', {})
    if cache.get('key') != key:
        cache['key'] = key
        buf = pack(l1, l2, l3)
        old = cache.get('buf')
        if old is not None and old.shape == buf.shape and old.device == buf.device:
            old.copy_(buf)           # in place: a captured HIP graph keeps reading this address
        else:
            cache['buf'] = buf
    return cache['buf'], (l1, l2, l3), slope


def run(block, z, code, inverse, log_q=None, sign=1.0, in_gather=None, out_gather=None):
    """Whole AffineCouplingBlock on [B, D] in one launch; same (out, log_det) contract as the
    three-step path of AffineCouplingBlock._run."""
    buf, (l1, _, l3), slope = packed_weights(block)
    c = z.shape[1]
    head = c - c // 2
    if block.split_mode == 'channel':
        cond_off, t_off, d_t = 0, head, c - head
    else:
        cond_off, t_off, d_t = head, 0, head
    return _lib.affine_layer_fused(z, buf, cond_off, l1.in_features, t_off, d_t, l1.out_features, slope, code,
                                   inverse, logdet=log_q, sign=sign, in_gather=in_gather, out_gather=out_gather)


STACK_MAX = 16


def _geometry(block, c):
    head = c - c // 2
    return (0, head, c - head) if block.split_mode == 'channel' else (head, 0, head)


def plan_stack(order, start, z, inverse):
    """Longest run order[start:end] of one-kernel-eligible AffineCouplingBlocks of ONE conditioner shape, with at most
    one Permute between neighbours (and optionally before the first / after the last), that
    ``vcnf_affine_stack_fused_f32`` can execute in a single launch.  Returns (end, steps, trailing_permute) with
    steps = [(block, permute_before or None)], or None when the run has fewer than two blocks."""
    from .flows.affine.coupling import AffineCouplingBlock
    from .flows.mixing import Permute
    if z.dim() != 2:
        return None
    steps, pending, j, shape = [], None, start, None
    while j < len(order) and len(steps) < STACK_MAX:
        f = order[j]
        if isinstance(f, Permute) and pending is None:
            pending = f
            j += 1
            continue
        if isinstance(f, AffineCouplingBlock) and f.fusable(z):
            (l1, l2, l3), slope = _linears(f.flows[1].param_map)
            if not _lib.lib().vcnf_affine_stack_fused_supported(l1.in_features, l1.out_features, l3.out_features, z.shape[1]):
                break                      # e.g. more than 128 features: the layers run one launch each
            core = f.flows[1]
            sh = (l1.in_features, l1.out_features, l3.out_features, slope, core.scale, core.scale_map,
                  _geometry(f, z.shape[1])[2])
            if shape is None:
                shape = sh
            if sh == shape:
                steps.append((f, pending))
                pending = None
                j += 1
                continue
        break
    # a permutation still pending here executes right after the run's last block: the kernel applies it to its result
    if len(steps) < 2:
        return None
    return j, steps, pending


def _struct_key(f):
    """What a plan / launch descriptor depends on besides module identity (structural edits in place on an
    existing module - swapping param_map, changing scale_map or split_mode, re-seeding a Permute - must not leave a stale
    plan in use): split mode, scale flags and the conditioner's identity of a block; the index buffers' storage and
    version of a Permute."""
    core = getattr(f, 'flows', [None, None])[1] if hasattr(f, 'split_mode') else None
    if core is not None:
        return (f.split_mode, core.scale, core.scale_map, id(core.param_map))
    perm = getattr(f, 'perm', None)
    if torch.is_tensor(perm):
        inv = getattr(f, 'inv_perm', perm)
        return (perm.data_ptr(), perm._version, inv.data_ptr(), inv._version)
    return None


def cached_plan(owner, order, start, z, inverse):
    """plan_stack memoised on the calling model for evaluations without autograd: the plan depends only on the
    modules in ``order`` (identity, their ``fused`` switches), the position, the direction and the feature count."""
    if torch.is_grad_enabled():
        return plan_stack(order, start, z, inverse)
    key = (start, bool(inverse), z.dim(), z.shape[-1], z.dtype, tuple(map(id, order)),
           tuple(getattr(f, 'fused', None) for f in order), tuple(_struct_key(f) for f in order))
    plans = owner.__dict__.setdefault('_stack_plans', {})
    if key not in plans:
        if len(plans) > 32:
            plans.clear()
        plans[key] = plan_stack(order, start, z, inverse)
    return plans[key]


def run_stack(steps, trailing, z, code, inverse, log_q, sign):
    """Execute a planned run (see plan_stack) in one launch."""
    first = steps[0][0]
    cache = first.__dict__.setdefault('_fused_affine_stack', {})
    # validity of everything cached below: the run's blocks and the (address, version) of their conditioner parameters
    ids = tuple(id(b) for b, _ in steps)
    if cache.get('ids') != ids:
        cache.clear()
        cache['ids'] = ids
        cache['params'] = [p for b, _ in steps for lin in _linears(b.flows[1].param_map)[0] for p in (lin.weight, lin.bias)]
    key = tuple((p.data_ptr(), p._version) for p in cache['params'])
    if cache.get('key') != key:
        # one weight buffer for the run, shared by both directions and rewritten IN PLACE when a layer's pack changes: a
        # captured HIP graph keeps reading the same address (GraphedFlow.refresh)
        new = torch.cat([packed_weights(b)[0] for b, _ in steps]).contiguous()
        old = cache.get('wpack')
        if old is not None and old.shape == new.shape and old.device == new.device:
            old.copy_(new)
        else:
            cache['wpack'] = new
        # ... and the split-half fragments of the second / third dense layers (same in-place rule)
        with torch.no_grad():
            new3 = torch.cat([pack_h3(*_linears(b.flows[1].param_map)[0][1:]) for b, _ in steps]).contiguous() \
                if _linears(first.flows[1].param_map)[0][0].out_features % 32 == 0 else None
        old3 = cache.get('wpack_h3')
        if new3 is not None and old3 is not None and old3.shape == new3.shape and old3.device == new3.device:
            old3.copy_(new3)
        else:
            cache['wpack_h3'] = new3
        cache['key'] = key
    dkey = (bool(inverse), z.shape[1], str(z.device), None if trailing is None else (id(trailing), _struct_key(trailing)),
            tuple((_struct_key(b), None if p is None else (id(p), _struct_key(p))) for b, p in steps))
    desc = cache.setdefault('desc', {}).get(bool(inverse))
    if desc is None or desc[0] != dkey:
        (l1, _, l3), slope = _linears(first.flows[1].param_map)
        rows, layers = [], []
        for blk, perm in steps:
            gb = -1
            if perm is not None:
                gb = len(rows)
                rows.append(perm._idx32(inverse, z.device))
            cond_off, t_off, d_t = _geometry(blk, z.shape[1])
            layers.append((cond_off, t_off, d_t, gb))
        ga = -1
        if trailing is not None:
            ga = len(rows)
            rows.append(trailing._idx32(inverse, z.device))
        gathers = torch.stack(rows).contiguous() if rows else None
        desc = (dkey, layers, ga, gathers, l1.in_features, l1.out_features, slope)
        cache['desc'][bool(inverse)] = desc
    _, layers, ga, gathers, c_in, hidden, slope = desc
    # (the split-half form keeps 32-row strips: two per wave = 1 KB x D of LDS per workgroup, D <= 64)
    use_h3 = cache.get('wpack_h3') is not None and z.shape[1] <= 64 and all(h3_ok(b) for b, _ in steps)
    return _lib.affine_stack_fused(z, cache['wpack'], layers, ga, gathers, c_in, hidden, slope, code,
                                   inverse, logdet=log_q, sign=sign, wpack_h3=cache['wpack_h3'] if use_h3 else None)
