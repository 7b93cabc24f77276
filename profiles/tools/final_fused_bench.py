"""Last layer + splines in one kernel (csrc/fused_final.hip) against the three-step path."""
import sys, time, os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import vcnf_amd as nf
torch.manual_seed(0)

def timeit(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

with torch.no_grad():
    for name, d, k, blocks, c, layers, B in (("C5 layer stack", 1024, 16, 2, 0, 24, 16384),
                                             ("D=48 K=8 3-block (d_id = 24: outside the one-kernel families)", 48, 8, 3, 16, 12, 1 << 18),
                                             ("D=128 K=16", 128, 16, 2, 0, 12, 1 << 17)):
        flows = [nf.flows.CoupledRationalQuadraticSpline(d, blocks, 128, k, reverse_mask=bool(i % 2), num_context_channels=c or None) for i in range(layers)]
        m = nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows).cuda()
        x, e = torch.randn(B, d, device='cuda'), torch.randn(B, d, device='cuda')
        ctx = torch.randn(B, c, device='cuda') if c else None
        kw = {"context": ctx} if c else {}
        for fused in (True, False):
            for f in m.flows: f.prqct.fused = fused
            dt = timeit(lambda: (m.log_prob(x, **kw), m.sample_from(e, **kw)), 3)
            print("%-50s B=%7d %s: %8.2f ms/step %8.2f M transforms/s" % (name, B, "final-fused" if fused else "three-step ", dt * 1e3, 2 * B / dt / 1e6), flush=True)
