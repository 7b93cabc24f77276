import sys, time
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import vcnf_amd as nf
torch.manual_seed(0)
def affine(layers, d, widths):
    flows = []
    for _ in range(layers):
        flows += [nf.flows.AffineCouplingBlock(nf.nets.MLP(widths)), nf.flows.Permute(d, mode="swap")]
    return nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows).cuda()
def c3():
    flows = [nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, reverse_mask=bool(i % 2), num_context_channels=16) for i in range(12)]
    return nf.NormalizingFlow(nf.distributions.DiagGaussian(64), flows).cuda()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for name, m, d, c, B in (("C1", affine(4, 2, [1, 32, 32, 2]), 2, None, 4096), ("C2", affine(8, 32, [16, 64, 64, 32]), 32, None, 4096), ("C3", c3(), 64, 16, 4096), ("C3", c3(), 64, 16, 65536)):
    x = torch.randn(B, d, device='cuda'); ctx = torch.randn(B, c, device='cuda') if c else None
    kw = {"context": ctx} if c else {}
    g = nf.GraphedFlow(m, B, c)
    with torch.no_grad():
        te = timeit(lambda: m.log_prob(x, **kw))
    tg = timeit(lambda: g.log_prob(x, ctx))
    print("%s B=%d log_prob: eager %.3f ms, graph %.3f ms (%.2fx)" % (name, B, te * 1e3, tg * 1e3, te / tg))
