import sys, time
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import vcnf_amd as nf
torch.manual_seed(0)
flows = [nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, reverse_mask=bool(i % 2), num_context_channels=16) for i in range(12)]
model = nf.NormalizingFlow(nf.distributions.DiagGaussian(64), flows).cuda()
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
for B in (16384, 131072):
    x, c = torch.randn(B, 64, device='cuda'), torch.randn(B, 16, device='cuda')
    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.forward_kld(x, context=c)
        loss.backward()
        opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 10
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print("train step C3 B=%d: %.2f ms  %.2f M samples/s  peak mem %.1f GB" % (B, dt * 1e3, B / dt / 1e6, torch.cuda.max_memory_allocated() / 2**30))
