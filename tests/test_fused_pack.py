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


# ---------------------------------------------------------------------------------------------------
# fp16 split-half kernel (csrc/fused_layer_v6.hip): v_mfma_f32_32x32x16_f16 dataflow as measured on gfx950
# (profiles/r02_mfma_32x32x16_layout.txt): A lane l = row l % 32, k-slots (l / 32, 0..7); B lane l = column
# l % 32, same slots; D register r of lane l = row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32.
C32, KG = LANE & 31, LANE >> 5


def mfma32(a, b, acc):
    """a, b: [64, 8] per-lane operands; acc: [64, 16]."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for i in range(8):
        A[C32, 8 * KG + i] = a[:, i]
        B[8 * KG + i, C32] = b[:, i]
    Dm = A @ B
    return acc + np.stack([Dm[8 * (r >> 2) + 4 * KG + (r & 3), C32] for r in range(16)], axis=1)


def _halves(floats):
    """float32 words of the packed buffer -> the fp16 values they carry (two per word)."""
    return np.ascontiguousarray(floats.astype(np.float32)).view(np.float16).astype(np.float64)


def run_emulated6(buf, x_id, ctx, wh_scale, DI=32, DT=32, C=16, H=128, NBLK=2, K=8):
    """One 32-sample column block through the packed conditioner -> logits [32, DT, P] (scales undone)."""
    P = 3 * K - 1
    NB, NT0, NTH, NTC, NG = H // 32, (DI + C) // 16, H // 16, C // 16, DT // 4
    pos = [0]

    def take(n):
        out = buf[pos[0]:pos[0] + n]
        pos[0] += n
        return out

    def frags(nb_count, nt):
        hl = _halves(take(nb_count * nt * 512)).reshape(nb_count, nt, 2, 64, 8)
        return hl[:, :, 0] + hl[:, :, 1] / 2048.0

    def bias16(n):
        return take(n).astype(np.float64).reshape(n // 32, 2, 16)

    def operand_in(vec, t):          # natural order: slot (kg, i) of k-step t = vec[:, 16 t + 8 kg + i]
        return np.stack([vec[C32, 16 * t + 8 * KG + i] for i in range(8)], axis=1)

    def operand_acc(accs, t, relu):  # k-step t = registers 8 hh .. 8 hh + 7 of row block t // 2
        v = accs[t >> 1][:, 8 * (t & 1):8 * (t & 1) + 8]
        return np.maximum(v, 0) if relu else v

    def layer(w, b, nt, bop):
        out = []
        for nb in range(w.shape[0]):
            acc = b[nb][KG]
            for t in range(nt):
                acc = mfma32(w[nb, t], bop(t), acc)
            out.append(acc)
        return out
    xin = np.concatenate([x_id, ctx], axis=1)
    w0 = frags(NB, NT0); b0 = bias16(H)
    h = layer(w0, b0, NT0, lambda t: operand_in(xin, t))
    for _ in range(NBLK):
        wa = frags(NB, NTH); ba = bias16(H)
        wb = frags(NB, NTH); bb = bias16(H)
        if C:
            wc = frags(NB, NTC); bc = bias16(H)
        t1 = layer(wa, ba, NTH, lambda t: operand_acc(h, t, True))
        t2 = layer(wb, bb, NTH, lambda t: operand_acc(t1, t, True))
        if C:
            gate = layer(wc, bc, NTC, lambda t: operand_in(ctx, t))
            t2 = [a / (1 + np.exp2(-g)) for a, g in zip(t2, gate)]
        h = [a + b for a, b in zip(h, t2)]
    wf = frags(NG * 3, NTH).reshape(NG, 3, NTH, 64, 8)
    bf = take(NG * 96).astype(np.float64).reshape(NG, 2, 48)
    assert pos[0] == len(buf)
    logits = np.zeros((32, DT, P))
    log2e = 1.4426950408889634
    for g in range(NG):
        pa = []
        for b in range(3):
            acc = bf[g][KG][:, 16 * b:16 * b + 16]
            for t in range(NTH):
                acc = mfma32(wf[g, b, t], operand_acc(h, t, False), acc)
            pa.append(acc)
        for lane in range(64):
            for v in range(48):
                f2, tl = divmod(v, 24)
                if tl < P:
                    sc = wh_scale * log2e if tl < 2 * K else log2e
                    logits[C32[lane], 4 * g + 2 * KG[lane] + f2, tl] = pa[v >> 4][lane, v & 15] / sc
                else:
                    assert pa[v >> 4][lane, v & 15] == 0.0            # padding row
    return logits


def _check6(ctx_dim, d=64):
    torch.manual_seed(14 + ctx_dim + d)
    m = nf.flows.CoupledRationalQuadraticSpline(d, 2, 128, 8, num_context_channels=ctx_dim or None)
    net = m.prqct.transform_net
    with torch.no_grad():
        for p in net.parameters():
            p.normal_(0, 0.3)
    di = d // 2
    wh_scale = 1.0 / np.sqrt(128.0)
    buf = fused.pack_layer_h3(net, di, 23, wh_scale, 8).numpy()
    assert len(buf) == nf.lib().vcnf_rqs_layer_fused_pack_floats(di, di, ctx_dim, 2)
    x_id = torch.randn(32, di, dtype=torch.float64)
    ctx = torch.randn(32, ctx_dim, dtype=torch.float64) if ctx_dim else None
    sd = {k: v.detach().double() for k, v in net.state_dict().items()}
    want = ON.residual_net(sd, "", x_id, ctx).reshape(32, di, 23).numpy()
    got = run_emulated6(buf, x_id.numpy(), ctx.numpy() if ctx_dim else np.zeros((32, 0)), wh_scale,
                        DI=di, DT=di, C=ctx_dim)
    # weights carry 22 significant bits (hi + lo / 2048), activations are exact here
    assert np.allclose(got, want, rtol=2e-5, atol=2e-5 * np.abs(want).max()), np.abs(got - want).max()


def test_pack_layout_h3_conditional():
    _check6(16)


def test_pack_layout_h3_unconditional():
    _check6(0)


def test_pack_layout_h3_d32():
    _check6(16, d=32)
    _check6(0, d=32)
