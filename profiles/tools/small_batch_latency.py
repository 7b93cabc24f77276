"""C3 / C2 log_prob latency at the reference drivers' batch sizes (/root/reference/run.py:45-47 uses 1024-2048),
eager and under GraphedFlow, per matrix path of the RQS coupling.  VERDICT r2 item 8."""
import sys, time
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import vcnf_amd as nf

torch.manual_seed(0)


def c3(**attrs):
    flows = [nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, reverse_mask=bool(i % 2), num_context_channels=16)
             for i in range(12)]
    m = nf.NormalizingFlow(nf.distributions.DiagGaussian(64), flows).cuda()
    for f in flows:
        for k, v in attrs.items():
            setattr(f.prqct, k, v)
    return m


def c2():
    flows = []
    for _ in range(8):
        flows += [nf.flows.AffineCouplingBlock(nf.nets.MLP([16, 64, 64, 32])), nf.flows.Permute(32, mode="swap")]
    return nf.NormalizingFlow(nf.distributions.DiagGaussian(32), flows).cuda()


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


variants = [("C3 fp16x3", lambda: c3(), 64, 16), ("C3 fp32", lambda: c3(fused_precision='fp32'), 64, 16),
            ("C3 split (three-step)", lambda: c3(fused=False), 64, 16), ("C2", c2, 32, None)]
if len(sys.argv) > 1:
    variants = [v for v in variants if any(a in v[0] for a in sys.argv[1:])]
for name, mk, d, c in variants:
    m = mk()
    for B in (1024, 2048, 4096, 16384):
        x = torch.randn(B, d, device='cuda')
        ctx = torch.randn(B, c, device='cuda') if c else None
        kw = {"context": ctx} if c else {}
        g = nf.GraphedFlow(m, B, c)
        with torch.no_grad():
            te = timeit(lambda: m.log_prob(x, **kw))
        tg = timeit(lambda: g.log_prob(x, ctx))
        print("%s B=%d log_prob: eager %.3f ms, graph %.3f ms" % (name, B, te * 1e3, tg * 1e3), flush=True)
