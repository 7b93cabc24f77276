"""C3 Adam step on forward_kld at the reference drivers batch sizes: eager loop against nf.GraphedTrainStep (one HIP graph per step)."""
import sys, time
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import vcnf_amd as nf
torch.manual_seed(0)
def mk():
    flows = [nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, reverse_mask=bool(i % 2), num_context_channels=16) for i in range(12)]
    return nf.NormalizingFlow(nf.distributions.DiagGaussian(64), flows).cuda()
for B in (1024, 2048, 16384):
    x, c = torch.randn(B, 64, device='cuda'), torch.randn(B, 16, device='cuda')
    model = mk()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
    def eager():
        opt.zero_grad(set_to_none=True)
        loss = model.forward_kld(x, context=c); loss.backward(); opt.step()
    for _ in range(3): eager()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): eager()
    torch.cuda.synchronize(); te = (time.perf_counter() - t) / 10
    step = nf.GraphedTrainStep(model, opt, batch=B, context_features=16)
    for _ in range(3): step(x, c)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): step(x, c)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t) / 20
    print("C3 Adam step B=%d: eager %.2f ms (%.2f M samples/s), one HIP graph %.2f ms (%.2f M samples/s)" % (B, te * 1e3, B / te / 1e6, tg * 1e3, B / tg / 1e6), flush=True)


# the reference drivers' own model family (/root/reference/run.py:58-68), fp64, maximum-likelihood step
def realnvp(d, h, pairs):
    b = torch.tensor([1.0 if i % 2 == 0 else 0.0 for i in range(d)])
    flows = []
    for i in range(pairs):
        s, t = nf.nets.MLP([d, h, d], init_zeros=True), nf.nets.MLP([d, h, d], init_zeros=True)
        flows += [nf.flows.MaskedAffineFlow(b if i % 2 == 0 else 1 - b, t, s), nf.flows.ActNorm(d)]
    return nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows).double().cuda()


for d, h, pairs, B in ((2, 16, 32, 1024), (15, 30, 16, 2048)):
    x = torch.randn(B, d, device='cuda', dtype=torch.float64)
    model = realnvp(d, h, pairs)
    with torch.no_grad():
        model.log_prob(x)                                  # ActNorm initialisation
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)
    def eager():
        opt.zero_grad(set_to_none=True)
        loss = model.forward_kld(x); loss.backward(); opt.step()
    for _ in range(3): eager()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): eager()
    torch.cuda.synchronize(); te = (time.perf_counter() - t) / 10
    step = nf.GraphedTrainStep(model, opt, batch=B)
    for _ in range(3): step(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): step(x)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t) / 20
    # the optimiser is most of what is left: capturable foreach-Adam runs ~650 three-microsecond kernels per step over
    # the model's 450 small fp64 tensors; torch's fused Adam is one multi-tensor kernel
    model2 = realnvp(d, h, pairs)
    with torch.no_grad():
        model2.log_prob(x)
    opt2 = torch.optim.Adam(model2.parameters(), lr=1e-4, capturable=True, fused=True)
    step2 = nf.GraphedTrainStep(model2, opt2, batch=B)
    for _ in range(3): step2(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): step2(x)
    torch.cuda.synchronize(); tf = (time.perf_counter() - t) / 20
    print("RealNVP D=%d H=%d, %d x [MaskedAffineFlow, ActNorm], fp64, Adam step on forward_kld, B=%d: eager %.2f ms, one HIP graph %.2f ms "
          "(%.2f ms with torch's fused Adam)" % (d, h, pairs, B, te * 1e3, tg * 1e3, tf * 1e3), flush=True)
