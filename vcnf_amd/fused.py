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


def pack_layer(net, d_t, p):
    """Flat fp32 buffer in the layout of csrc/fused_layer.hip::PackLayout."""
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
    if coupling.tails != 'linear' or coupling.unconditional_transform is None and False:
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


def packed_weights(coupling):
    """Cached packed buffer; rebuilt when any conditioner parameter changed."""
    net = coupling.transform_net
    key = tuple((p.data_ptr(), p._version) for p in net.parameters())
    cache = coupling.__dict__.get('_fused_pack')
    if cache is None or cache[0] != key:
        with torch.no_grad():
            buf = pack_layer(net, coupling.num_transform_features, coupling._transform_dim_multiplier())
        cache = (key, buf)
        coupling.__dict__['_fused_pack'] = cache
    return cache[1]


def run(coupling, inputs, context, sampling, log_q=None, sign=1.0):
    net = coupling.transform_net
    shared = coupling.unconditional_transform.logits() if coupling.unconditional_transform is not None else None
    return _lib.rqs_layer_fused(inputs, context, coupling._index32('tf'), coupling._index32('id'),
                                net.context_features or 0, net.hidden_features, len(net.blocks),
                                packed_weights(coupling), shared, coupling._cfg(True), sampling,
                                logdet=log_q, sign=sign)
