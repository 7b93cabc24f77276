import sys, time
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import os
from vcnf_amd import build as _B
if os.environ.get('LIBV'): _B.LIB = os.environ['LIBV']
import vcnf_amd as nf
torch.manual_seed(0)
def model(layers, d, widths):
    flows = []
    for _ in range(layers):
        flows += [nf.flows.AffineCouplingBlock(nf.nets.MLP(widths)), nf.flows.Permute(d, mode="swap")]
    return nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows).cuda()
for name, (layers, d, widths, B) in {"C1": (4, 2, [1, 32, 32, 2], 1 << 20), "C2": (8, 32, [16, 64, 64, 32], 1 << 20)}.items():
    m = model(layers, d, widths)
    x, eps = torch.randn(B, d, device='cuda'), torch.randn(B, d, device='cuda')
    for fused in (True, False):
        for f in m.flows:
            if hasattr(f, 'fused'): f.fused = fused
        def step():
            with torch.no_grad():
                m.log_prob(x); m.sample_from(eps)
        for _ in range(3): step()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print("%s fused=%d: %.2f ms/step  %.1f M transforms/s (log_prob + sample, B=%d, %d layers)" % (name, fused, dt * 1e3, 2 * B / dt / 1e6, B, layers))
