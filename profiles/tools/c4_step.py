"""Config C4 (Glow, 3 x 32 x 32, L = 3, 16 blocks per level, 256 hidden channels) log_prob + sample at a given batch:
the workload behind the per-kernel tables of profiles/r02_c4_*.  Random weights; the first call initialises the ActNorms.
Usage: python profiles/tools/c4_step.py [batch] [steps] [find]   (find = 1: MIOpen searches its solvers per shape)"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/profiles/", 1)[0])
import vcnf_amd as nf  # noqa: E402


def glow(levels=3, blocks=16, hidden=256, input_shape=(3, 32, 32)):
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


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    torch.backends.cudnn.benchmark = len(sys.argv) > 3 and sys.argv[3] == "1"
    torch.manual_seed(0)
    model = glow().cuda().eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("param_map.net.4.weight") or n.endswith("param_map.net.4.bias"):
                p.normal_(0.0, 0.02)               # the zero-initialised last conv would make every coupling the identity
        x = torch.rand(batch, 3, 32, 32, device="cuda")
        eps = [torch.randn(batch, *q.loc.shape[1:], device="cuda") for q in model.q0]
        for _ in range(2):
            lp = model.log_prob(x)
            z, lq = model.sample_from(eps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            lp = model.log_prob(x)
            z, lq = model.sample_from(eps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(lp).all() and torch.isfinite(lq).all()
    print("C4 batch %d (MIOpen find %s): %.2f ms per step (log_prob + sample), %.3f M transforms/s" % (
        batch, torch.backends.cudnn.benchmark, dt * 1e3, 2 * batch / dt / 1e6))


if __name__ == "__main__":
    main()
