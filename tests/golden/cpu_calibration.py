#!/usr/bin/env python3
"""BASELINE.md section 4, "cost-calibrated once against the real reference": time the CPU oracle (what bench.py's
cpu_baseline leg runs on the GPU box) next to the IMPORTED reference on the same C3 stack, same inputs, same thread
count.  Build container only (needs /root/reference; the reference never travels).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/cpu_calibration.py > profiles/r02_cpu_baseline_calibration.txt
(kept under tests/: it imports the oracle, which only tests/, smoke() and bench.py's cpu_baseline leg may do)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/golden -> repo root
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np      # noqa: E402
import torch            # noqa: E402
import make_golden as G  # noqa: E402  (imports the reference behind the SURVEY 8c placeholders)
from helpers import oracle_c3_stack  # noqa: E402

torch.set_grad_enabled(False)
threads = int(os.environ.get("THREADS", str(len(os.sched_getaffinity(0)))))
torch.set_num_threads(threads)
B = int(os.environ.get("BATCH", "16384"))
ref = G.C3Stack()                                   # the reference's modules, config C3
for n, p in ref.state_dict().items():
    if "unnormalized_" in n:
        p.normal_(0.0, 0.5)
sd = {k: v.clone() for k, v in ref.state_dict().items()}
ora = oracle_c3_stack(sd)
g = torch.Generator().manual_seed(3)
x, ctx, eps = (torch.randn(B, n, generator=g) for n in (64, 16, 64))


def best(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2]


# interleaved, so that a load change on the host hits both alike
f_ref = lambda: (ref.log_prob(x, ctx), ref.sample(eps, ctx))
f_ora = lambda: (ora.log_prob(x, ctx), ora.sample_from(eps, ctx))
f_ref(); f_ora()
tr, to = [], []
for _ in range(3):
    t0 = time.perf_counter(); f_ref(); tr.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); f_ora(); to.append(time.perf_counter() - t0)
t_ref, t_ora = sorted(tr)[1], sorted(to)[1]
lp_r, lp_o = ref.log_prob(x[:256], ctx[:256]), ora.log_prob(x[:256], ctx[:256])
lp_r = lp_r[1] if isinstance(lp_r, tuple) else lp_r     # the generator's stack returns (z, log_q, per-layer log_dets)
print("C3 stack (12 RQS couplings, D=64, ctx 16, hidden 128 x 2 blocks), batch %d, torch %s CPU fp32, %d threads" % (
    B, torch.__version__, threads))
print("log_prob + sample, 1 warm-up + 3 timed, median:")
print("  reference (imported from /root/reference): %.2f s  -> %.1f transforms/s" % (t_ref, 2 * B / t_ref))
print("  oracle    (oracle/, what bench.py times):  %.2f s  -> %.1f transforms/s" % (t_ora, 2 * B / t_ora))
print("  cost ratio oracle / reference: %.2f" % (t_ora / t_ref))
print("  max |log_prob difference| on 256 samples: %.3e (|log_prob| ~ %.0f)" % (
    float((lp_r - lp_o).abs().max()), float(lp_r.abs().mean())))
