"""All five BASELINE.json configs on one GPU (per-GPU share of the sharded ones), log_prob + sample."""
import sys, time
import os
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [_R]
import torch
import vcnf_amd as nf
torch.manual_seed(0)

def timeit(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

def affine(layers, d, widths):
    flows = []
    for _ in range(layers):
        flows += [nf.flows.AffineCouplingBlock(nf.nets.MLP(widths)), nf.flows.Permute(d, mode="swap")]
    return nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows).cuda()

def _glow_model(levels, blocks, hidden, input_shape):
    q0, merges, flows, L = [], [], [], levels
    for i in range(L):
        fl = [nf.flows.GlowBlock(input_shape[0] * 2 ** (L + 1 - i), hidden, split_mode="channel", scale=True)
              for _ in range(blocks)]
        fl += [nf.flows.Squeeze()]
        flows += [fl]
        if i > 0:
            merges += [nf.flows.Merge()]
            shape = (input_shape[0] * 2 ** (L - i), input_shape[1] // 2 ** (L - i), input_shape[2] // 2 ** (L - i))
        else:
            shape = (input_shape[0] * 2 ** (L + 1), input_shape[1] // 2 ** L, input_shape[2] // 2 ** L)
        q0 += [nf.distributions.DiagGaussian(shape)]
    return nf.MultiscaleFlow(q0, flows, merges, class_cond=False)


def report(name, B, dt, note=""):
    print("%-3s batch %8d: %9.2f ms/step  %10.2f M transforms/s  %s" % (name, B, dt * 1e3, 2 * B / dt / 1e6, note), flush=True)

with torch.no_grad():
    # C1: 2-D two moons, 4 affine couplings, batch 4096
    m = affine(4, 2, [1, 32, 32, 2]); B = 4096
    x, e = torch.randn(B, 2, device='cuda'), torch.randn(B, 2, device='cuda')
    report("C1", B, timeit(lambda: (m.log_prob(x), m.sample_from(e)), 50), "eager")
    g = nf.GraphedFlow(m, B)
    report("C1", B, timeit(lambda: (g.log_prob(x), g.sample_from(e)), 50), "HIP graph")
    # C2: tabular D=32, 8 affine couplings, batch 262144
    m = affine(8, 32, [16, 64, 64, 32]); B = 262144
    x, e = torch.randn(B, 32, device='cuda'), torch.randn(B, 32, device='cuda')
    report("C2", B, timeit(lambda: (m.log_prob(x), m.sample_from(e)), 20))
    # C3: the bench config
    flows = [nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, reverse_mask=bool(i % 2), num_context_channels=16) for i in range(12)]
    m = nf.NormalizingFlow(nf.distributions.DiagGaussian(64), flows).cuda(); B = 1 << 20
    x, c, e = torch.randn(B, 64, device='cuda'), torch.randn(B, 16, device='cuda'), torch.randn(B, 64, device='cuda')
    report("C3", B, timeit(lambda: (m.log_prob(x, c), m.sample_from(e, c)), 5))
    del m, x, c, e
    # C4: 32x32x3 multiscale Glow (3 levels x 4 blocks, 64 hidden channels), per-GPU share 16384 of 131072
    m = _glow_model(levels=3, blocks=4, hidden=64, input_shape=(3, 32, 32)).cuda(); B = 2048
    x = torch.randn(B, 3, 32, 32, device='cuda')
    m.log_prob(x)            # ActNorm data-dependent init
    eps = [torch.randn((B,) + tuple(q.shape), device='cuda') for q in m.q0]
    dt = timeit(lambda: (m.log_prob(x), m.sample_from(eps)), 3)
    report("C4", B, dt, "x8 chunks for the 16384-image shard: %.1f ms" % (8 * dt * 1e3))
    del m, x, eps
    torch.cuda.empty_cache()
    # C5: D=1024, 24 RQS layers, 16 bins, per-GPU share 524288 of 4M, walked in micro-batches
    flows = [nf.flows.CoupledRationalQuadraticSpline(1024, 2, 128, 16, reverse_mask=bool(i % 2)) for i in range(24)]
    m = nf.NormalizingFlow(nf.distributions.DiagGaussian(1024), flows).cuda(); B = 16384
    x, e = torch.randn(B, 1024, device='cuda'), torch.randn(B, 1024, device='cuda')
    dt = timeit(lambda: (m.log_prob(x), m.sample_from(e)), 2)
    report("C5", B, dt, "x32 micro-batches for the 524288-sample shard: %.2f s; peak mem %.1f GB" % (32 * dt, torch.cuda.max_memory_allocated() / 2**30))
