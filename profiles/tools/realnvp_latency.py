"""log_prob / sample latency of the reference drivers' own model family (/root/reference/run.py:58-68: K x
[MaskedAffineFlow(b, t, s), ActNorm], s, t = MLP([D, 8 D, D]), D = 2, .double(), 1024 samples) per layer and as one launch
(csrc/masked_affine_stack.hip), eager and under GraphedFlow."""
import sys, time
import os
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))]
import torch
import vcnf_amd as nf


def model(d, h, pairs, dtype):
    b = torch.tensor([1.0 if i % 2 == 0 else 0.0 for i in range(d)])
    flows = []
    for i in range(pairs):
        s, t = nf.nets.MLP([d, h, d], init_zeros=True), nf.nets.MLP([d, h, d], init_zeros=True)
        flows += [nf.flows.MaskedAffineFlow(b if i % 2 == 0 else 1 - b, t, s), nf.flows.ActNorm(d)]
    m = nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if ".net.2." in n:
                p.normal_(0.0, 0.2)
    return m.to(dtype).cuda().eval()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


torch.manual_seed(0)
for d, h, pairs, dtype, B in ((2, 16, 32, torch.float64, 1024), (15, 30, 16, torch.float64, 2048), (2, 16, 32, torch.float32, 65536)):
    m = model(d, h, pairs, dtype)
    x = torch.randn(B, d, device="cuda", dtype=dtype)
    with torch.no_grad():
        m.log_prob(x)                                   # ActNorm initialisation
        res = {}
        for stacks in (False, True):
            m.fuse_masked_stacks = stacks
            te = timeit(lambda: m.log_prob(x))
            ts = timeit(lambda: m.sample(B))
            g = nf.GraphedFlow(m, B)
            tg = timeit(lambda: g.log_prob(x))
            res[stacks] = (te, ts, tg)
    print("D=%d H=%d, %d x [MaskedAffineFlow, ActNorm], %s, %d samples: log_prob eager %.3f -> %.3f ms, sample eager %.3f -> %.3f ms, "
          "log_prob under a HIP graph %.3f -> %.3f ms (per layer -> one launch)"
          % (d, h, pairs, str(dtype).split('.')[-1], B, res[False][0], res[True][0], res[False][1], res[True][1], res[False][2], res[True][2]), flush=True)
