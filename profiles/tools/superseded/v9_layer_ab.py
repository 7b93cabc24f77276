"""A / B of the fused RQS layer kernel structures on one C3 layer (D = 64, context 16, hidden 128 x 2 blocks, 8 bins), 1 M
samples per launch: bitwise comparison of the results and launch time (HIP events, 20 launches back to back) of
precision 1 (current structure) against precision 2 (round-2 structure) and 0 (exact fp32).
    python profiles/tools/layer_ab.py [blocks]"""
import os
import sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import vcnf_amd as nf
from vcnf_amd import _lib, fused

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(0)
lay = nf.flows.CoupledRationalQuadraticSpline(64, blocks, 128, 8, num_context_channels=16).cuda().eval()
with torch.no_grad():
    for n, p in lay.named_parameters():
        if "unnormalized_" in n:
            p.normal_(0.0, 0.5)
c = lay.prqct
net = c.transform_net
shared = c.unconditional_transform.logits()


def run(prec, x, ctx, sampling, safe=False):
    pack = fused.packed_weights(c, 1 if prec == 2 else prec)
    return _lib.rqs_layer_fused(x, ctx, c._index32('tf'), c._index32('id'), 16, 128, blocks, prec, pack, shared,
                                c._cfg(True), sampling, wpack_f32=fused.packed_weights(c, 0) if safe else None)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for B in (1 << 20, 128 * 5 + 37):
        xb, cb = torch.randn(B, 64, device='cuda'), torch.randn(B, 16, device='cuda')
        for sampling in (False, True):
            y1, l1 = run(1, xb, cb, sampling)
            y2, l2 = run(2, xb, cb, sampling)
            torch.cuda.synchronize()
            print("B %8d %-8s  new == round-2 structure bitwise: y %s  log_det %s   max|dy| %.3g max|dld| %.3g" % (
                B, "sampling" if sampling else "density", torch.equal(y1, y2), torch.equal(l1, l2),
                float((y1 - y2).abs().max()), float((l1 - l2).abs().max())))
    xb, cb = torch.randn(1 << 20, 64, device='cuda'), torch.randn(1 << 20, 16, device='cuda')
    for rep in range(2):
        for prec, name in ((1, "new structure"), (2, "round-2 structure"), (0, "exact fp32")):
            print("%-18s density %.4f ms  sampling %.4f ms" % (name, timeit(lambda: run(prec, xb, cb, False)),
                                                             timeit(lambda: run(prec, xb, cb, True))))
    print("%-18s density %.4f ms  sampling %.4f ms" % ("new + fp32 redo launch", timeit(lambda: run(1, xb, cb, False, True)),
                                                     timeit(lambda: run(1, xb, cb, True, True))))
