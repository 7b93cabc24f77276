import sys, os
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import vcnf_amd as nf
torch.manual_seed(0)
lay = nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, num_context_channels=16).cuda().eval()
lay.prqct.fused_precision = "fp16x3"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
with torch.no_grad():
    xb, cb = torch.randn(1 << 20, 64, device='cuda'), torch.randn(1 << 20, 16, device='cuda')
    print("%s  inverse %.3f ms forward %.3f ms" % (os.environ.get("VCNF_FUSED_KERNEL", "v3"), timeit(lambda: lay.inverse(xb, context=cb)), timeit(lambda: lay.forward(xb, context=cb))))
