"""The fused layer kernel consumes conditioner weights re-ordered into matrix-core
fragments (vcnf_amd/fused.py).  This CPU test emulates the v_mfma_f32_16x16x4_f32
dataflow the kernel relies on - lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15],
the result D[4 (l >> 4) + r][l & 15] lands in register r of lane l - and checks that
chaining the packed layers reproduces the ResidualNet output, including the
accumulator-as-next-operand k order and the row permutation of the last layer."""
import numpy as np
import torch

import vcnf_amd as nf
from vcnf_amd import fused
from oracle import nets as ON

LANE = np.arange(64)
M16, Q = LANE & 15, LANE >> 4


def mfma(a, b, acc):
    """a, b: [64] per-lane operands; acc: [64, 4]."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[M16, Q] = a
    B[Q, M16] = b
    D = A @ B
    return acc + np.stack([D[4 * Q + r, M16] for r in range(4)], axis=1)


def dense_block(frags, bop, acc):
    """frags: [NS4, 64, 4] for one row block."""
    for s4 in range(frags.shape[0]):
        for c in range(4):
            acc = mfma(frags[s4, :, c], bop(4 * s4 + c), acc)
    return acc


def run_emulated(buf, x_id, ctx, DI=32, DT=32, C=16, H=128, NBLK=2, K=8):
    """One 16-sample column block through the packed conditioner -> logits [16, DT, P]."""
    P = 3 * K - 1
    P4 = (P + 3) // 4
    NB, NS0, NSH, NSC = H // 16, (DI + C) // 4, H // 4, C // 4
    pos = [0]

    def take(n):
        out = buf[pos[0]:pos[0] + n]
        pos[0] += n
        return out
    hin = np.zeros((64, NS0))
    for s in range(DI // 4):
        hin[:, s] = x_id[M16, 4 * s + Q]
    for s in range(NSC):
        hin[:, DI // 4 + s] = ctx[M16, 4 * s + Q]

    def bias(vec, nb):
        return np.stack([vec[16 * nb + 4 * Q + r] for r in range(4)], axis=1)
    w0 = take(NB * NS0 * 64).reshape(NB, NS0 // 4, 64, 4)
    b0 = take(H)
    h = [dense_block(w0[nb], lambda s: hin[:, s], bias(b0, nb)) for nb in range(NB)]
    for _ in range(NBLK):
        wa = take(NB * NSH * 64).reshape(NB, NSH // 4, 64, 4); ba = take(H)
        wb = take(NB * NSH * 64).reshape(NB, NSH // 4, 64, 4); bb = take(H)
        if C:
            wc = take(NB * NSC * 64).reshape(NB, NSC // 4, 64, 4); bc = take(H)
        t = [dense_block(wa[nb], lambda s: np.maximum(h[s >> 2][:, s & 3], 0), bias(ba, nb)) for nb in range(NB)]
        for nb in range(NB):
            acc = dense_block(wb[nb], lambda s: np.maximum(t[s >> 2][:, s & 3], 0), bias(bb, nb))
            if C:
                gate = dense_block(wc[nb], lambda s: hin[:, DI // 4 + s], bias(bc, nb))
                acc = acc / (1 + np.exp(-gate))
            h[nb] = h[nb] + acc
    wf = take((DT // 4) * P4 * NSH * 64).reshape(DT // 4, P4, NSH // 4, 64, 4)
    bf = take((DT // 4) * 4 * 4 * P4).reshape(DT // 4, 4, 4 * P4)
    assert pos[0] == len(buf)
    logits = np.zeros((16, DT, P))
    for g in range(DT // 4):
        pa = [dense_block(wf[g, b], lambda s: h[s >> 2][:, s & 3],
                          np.stack([bf[g, Q, 4 * b + r] for r in range(4)], axis=1)) for b in range(P4)]
        for lane in range(64):
            for tpar in range(P):
                logits[M16[lane], 4 * g + Q[lane], tpar] = pa[tpar >> 2][lane, tpar & 3]
    return logits


def _check(ctx_dim):
    torch.manual_seed(4 + ctx_dim)
    m = nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, num_context_channels=ctx_dim or None)
    net = m.prqct.transform_net
    with torch.no_grad():
        for p in net.parameters():
            p.normal_(0, 0.3)
    buf = fused.pack_layer(net, 32, 23).double().numpy()
    assert len(buf) == nf.lib().vcnf_rqs_layer_fused_pack_floats(32, 32, ctx_dim, 2)
    x_id = torch.randn(16, 32, dtype=torch.float64)
    ctx = torch.randn(16, ctx_dim, dtype=torch.float64) if ctx_dim else None
    sd = {k: v.detach().double() for k, v in net.state_dict().items()}
    want = ON.residual_net(sd, "", x_id, ctx).reshape(16, 32, 23).numpy()
    got = run_emulated(buf, x_id.numpy(), ctx.numpy() if ctx_dim else np.zeros((16, 0)), C=ctx_dim)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5), np.abs(got - want).max()


def test_pack_layout_conditional():
    _check(16)


def test_pack_layout_unconditional():
    _check(0)
