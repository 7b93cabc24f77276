#!/usr/bin/env python3
"""Headline benchmark: flow transforms/sec (log_prob + sample) on BASELINE.json's
config C3 - conditional D=64 (context 16), 12 RQ-spline coupling layers (8 bins,
linear tails, ResidualNet hidden 128 x 2 blocks), batch 1M per GPU.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one log_prob pass + one sample pass over the rank's batch with all
inputs already resident in HBM: 2*B transforms (1 transform = one sample through
the 12-layer stack in one direction, base-distribution end cap included), plus
the single all-reduce of [sum log_prob, count].  Weak scaling: every rank holds
its own 1M-sample shard; no sample ever crosses ranks.

Scaling modes (--scaling): "weak" (default; every rank holds its own --batch samples) and "strong" (--batch is the
whole job, split contiguously over the ranks by vcnf_amd.shard_bounds).

The JSON line also carries
  roofline     - the dominant HIP kernel of the timed region.  Fused layers (default): one kernel per layer does
                 conditioner + splines; its work is the conditioner's algorithmic flop per launch (339 968 per
                 sample-layer) / mean launch duration measured with HIP events on the launch stream
                 during the timed region, against the dense f16 matrix peak / 3 (split-half operands: three
                 matrix instructions per product) or the fp32 matrix peak (--precision fp32).  --split: the
                 HBM-bound spline kernel, 3464 algorithmic bytes per sample-layer against 8 TB/s (and against
                 the measured 6.3 TB/s copy rate, frac_of_measured_copy_bw);
  other_matrix_path - the fused kernel's other matrix arithmetic on the same workload, timed exactly like the
                 headline: the same --warmup untimed steps, the same --steps timed steps, fenced by barrier +
                 synchronize on both sides, max over ranks;
  extra_configs - (N = 1, unless --no-extra) BASELINE.json's other GPU configurations C2 / C4 / C5 and the C3
                 training step, 1 warm-up + 2 timed steps each, each with its own ms_per_step, dtype and roofline;
  cpu_baseline - the CPU oracle (op-order-faithful PyTorch-CPU restatement of the reference path) timed on this
                 box's host cores on a bounded sample of the same workload: 1 warm-up + 3 timed runs, median
                 (rank 0, N=1 only).  Reported, not a target.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import vcnf_amd as nf            # noqa: E402
from vcnf_amd import _lib        # noqa: E402

D, CTX, LAYERS, HIDDEN, BLOCKS, BINS, TAIL = 64, 16, 12, 128, 2, 8, 3.0
HBM_PEAK = 8.0e12                                   # MI355X_MICROARCH.md: 8 TB/s spec
HBM_COPY = 6.29e12                                  # MI355X_MICROARCH.md: measured float4 copy rate
MFMA_F32_PEAK = 157.3e12                            # MI355X_MICROARCH.md: dense fp32 matrix peak
MFMA_F16_PEAK = 2500.0e12                           # MI355X_MICROARCH.md: dense f16/bf16 matrix peak
P = 3 * BINS - 1
BYTES_PER_SAMPLE_LAYER = 4 * D + 4 * (D // 2) * P + 4 * D + 8     # 3464, SURVEY 8(d)
# conditioner flop per sample-layer (339 968): 2 * (48*128 + 4*128*128 + 2*16*128 + 128*736)
FLOP_PER_SAMPLE_LAYER = 2 * ((D // 2 + CTX) * HIDDEN + 2 * BLOCKS * HIDDEN * HIDDEN +
                             BLOCKS * CTX * HIDDEN + HIDDEN * (D // 2) * P)


def build_model(device, seed=0):
    """Random-init weights of the C3 architecture exactly as the constructors
    leave them, except the unconditional spline logits, re-drawn N(0, 0.5^2):
    the identity init makes all bins equal."""
    torch.manual_seed(seed)
    flows = [nf.flows.CoupledRationalQuadraticSpline(D, BLOCKS, HIDDEN, BINS, tail_bound=TAIL,
                                                     reverse_mask=bool(i % 2), num_context_channels=CTX)
             for i in range(LAYERS)]
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(D), flows)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "unnormalized_" in n:
                p.normal_(0.0, 0.5)
    return model.to(device).eval()


def pmc_traffic(kernel, batch):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes of this
    same command (profiles/*_pmc_hbm_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs,
    gfx950 correction applied there).  bench.py cannot collect counters itself; null when
    no measurement for this kernel and batch size is on file, or when the file was measured
    on other kernel sources than the ones in the tree now (sha256 over vcnf_amd/csrc)."""
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    try:
        from make_pmc_traffic import kernel_source_hash
        current = kernel_source_hash()
    except Exception:
        return None
    best = None
    for name in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        if not name.endswith("_pmc_hbm_traffic.json"):
            continue
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if rec.get("batch_per_launch") != batch or rec.get("kernel_source_sha256") != current:
            continue
        vals = [v["hbm_bytes_per_launch"] for k, v in rec.get("kernels", {}).items()
                if kernel in k and "hbm_bytes_per_launch" in v]
        if vals:
            best = int(sum(vals) / len(vals))
    return best


def cpu_share():
    """Host cores this process can actually use: the scheduler affinity, cut down to the cgroup's CPU quota when
    there is one (the GPU box shows 256 cores in the affinity mask but gives one job a 16-core share; running 256
    threads on it took the baseline from 20 s to more than 5 minutes)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("VCNF_CPU_THREADS", "16"))))


def cpu_baseline(model, budget_s=20.0):
    """Oracle (kind 'port') on the host cores, bounded sample of the C3 workload."""
    from helpers import oracle_c3_stack
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    stack = oracle_c3_stack(sd, layers=LAYERS, num_bins=BINS, tail_bound=TAIL, hidden=HIDDEN)
    cores = cpu_share()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(99)

    def run(b):
        x, ctx, eps = (torch.randn(b, n, generator=g) for n in (D, CTX, D))
        t0 = time.perf_counter()
        with torch.no_grad():
            stack.log_prob(x, ctx)
            stack.sample_from(eps, ctx)
        return time.perf_counter() - t0

    run(512)                                   # thread pool, allocator
    probe_b = 2048
    t = run(probe_b)
    b = int(min(65536, max(probe_b, probe_b * (budget_s / 4.0 / max(t, 1e-3)))))     # 1 warm-up + 3 timed runs
    b -= b % 256
    run(b)                                     # warm-up at the timed size
    ts = sorted(run(b) for _ in range(3))
    t = ts[1]
    return {"value": round(2 * b / t, 1), "unit": "transforms/s", "cores": cores, "kind": "port",
            "sample": "oracle C3 stack, log_prob + sample at batch %d: 1 warm-up + 3 timed runs, median %.2f s "
                      "(min %.2f, max %.2f), torch CPU fp32, %d threads (min of scheduler affinity, cgroup CPU quota, 16-core job share)"
                      % (b, t, ts[0], ts[2], cores)}


def bench_other(args, device, rank, world, config=None, steps=None, warmup=None, cpu=True):
    """BASELINE.json's configurations C2, C4 and C5 (same schema as the headline line; not the headline metric).
    Returns the result dict on rank 0 (None elsewhere).  Weak scaling only: every rank holds the per-GPU batch."""
    from vcnf_amd.sharded import max_over_ranks
    config = config or args.config
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    if args.scaling != "weak":
        raise SystemExit("--config %s is reported per GPU (weak scaling) only" % config)
    torch.manual_seed(0)
    if config == "C2":
        d, layers, B = 32, 8, 262144 if args.batch == 1 << 20 else args.batch
        flows = []
        for _ in range(layers):
            flows += [nf.flows.AffineCouplingBlock(nf.nets.MLP([16, 64, 64, 32], init_zeros=False), scale_map="exp"),
                      nf.flows.Permute(d, mode="swap")]
        flop_sl, bytes_sl = 2 * (16 * 64 + 64 * 64 + 64 * 32), 264           # per sample-layer, SURVEY 8d
        tag, kname = "affine_stack_fused", "fused_affine_stack_kernel"
        # one launch = the whole stack in one direction; first conditioner layer on exact fp32 matrix instructions, second
        # and third on split-half f16 ones.  Ceiling of the mixture: time at peak = fp32 flop / fp32 peak + f16x3 flop / (f16 peak / 3)
        f32_flop, h3_flop = 2 * 16 * 64, 2 * (64 * 64 + 64 * 32)
        peak = flop_sl / (f32_flop / MFMA_F32_PEAK + h3_flop / (MFMA_F16_PEAK / 3.0))
        work, unit, bound = flop_sl * layers, "TFLOP/s", "mfma"
        hbm_bytes = 4 * d * 2 + 8                                             # x once, y once, log_q
        workload = "C2: tabular D=32, 8 affine couplings (MLP 16-64-64-32) + swap permutations, batch=%d per GPU" % B
        dtype = ("f32 (conditioner: first layer exact fp32 matrix instructions, second and third layer fp16x3 split "
                 "operands with fp32 accumulation and an on-device fp32 fallback for out-of-range activations)")
    elif config == "C4":
        d, layers, B = 3072, 48, 16384 if args.batch == 1 << 20 else args.batch
        levels = [(48, 4, 4), (24, 8, 8), (12, 16, 16)]
        q0, merges, flows = [], [], []
        for i, shape in enumerate(levels):
            fl = [nf.flows.GlowBlock(shape[0], 256, split_mode="channel", scale=True) for _ in range(16)]
            flows += [fl + [nf.flows.Squeeze()]]
            if i > 0:
                merges += [nf.flows.Merge()]
            q0 += [nf.distributions.DiagGaussian(shape if i == 0 else (shape[0] // 2,) + shape[1:])]
        # dominant kernel: the whole ConvNet2d conditioner of a GlowBlock (3x3 -> 1x1 -> nine tap matrices of the last
        # 3x3 convolution) in one launch, csrc/conv3x3_1x1.hip, on fp16 split-half operands.  Algorithmic flop per image
        # and launch at level (c, h, w): h w 2 (9 (c/2) 256 + 256 256 + 9 c 256); one launch per block and direction
        tag, kname = "convnet3_taps", "conv3x3_1x1_f16x3_kernel"
        flop_img = [hh * ww * 2 * (9 * (c // 2) * 256 + 256 * 256 + 9 * c * 256) for c, hh, ww in levels]
        work, peak, unit, bound = sum(flop_img) / 3.0, MFMA_F16_PEAK / 3.0, "TFLOP/s", "mfma"    # average launch
        bytes_sl = 0
        hbm_bytes = None
        workload = ("C4: 3 x 32 x 32 images, multiscale Glow (3 levels x 16 GlowBlocks, 256 hidden channels), batch=%d per "
                    "GPU; every kernel of the step is this repository's: the conv conditioner (3x3, 1x1, 3x3 convolutions) "
                    "of a block is one matrix-core launch + a shift-and-add launch, affine coupling, 1x1 convolution + "
                    "ActNorm mixers" % B)
        dtype = ("f32 (conv conditioner: fp16x3 split operands, 22-bit hi + lo fp16 halves, three matrix instructions per "
                 "product, fp32 accumulate, saturation counted; couplings and mixers fp32)")
    else:
        d, layers, B = 1024, 24, 524288 if args.batch == 1 << 20 else args.batch
        flows = [nf.flows.CoupledRationalQuadraticSpline(d, 2, 128, 16, reverse_mask=bool(i % 2)) for i in range(layers)]
        flop_sl, bytes_sl = 2 * (512 * 128 + 4 * 128 * 128 + 128 * 512 * 47), 104456
        tag, kname = "rqs_final_fused", "fused_final_kernel"
        work, peak, unit, bound = bytes_sl, HBM_PEAK, "GB/s", "hbm"          # SURVEY 8d yardstick for C5: operator-boundary bytes
        hbm_bytes = None
        workload = ("C5: D=1024, 24 RQS couplings (16 bins, ResidualNet 512 -> 24064), %d samples per GPU and step "
                    "(the per-GPU shard of the 4M batch is 524288; the conditioner logits never exist in memory, so the "
                    "shard is one pass)" % B)
        dtype = "f32 (last conditioner layer: fp16x3 split operands, fp32 accumulate; trunk fp32)"
    if config == "C4":
        model = nf.MultiscaleFlow(q0, flows, merges, class_cond=False).to(device).eval()
    else:
        model = nf.NormalizingFlow(nf.distributions.DiagGaussian(d), flows).to(device).eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "unnormalized_" in n:
                p.normal_(0.0, 0.5)
            if n.endswith("param_map.net.4.weight") or n.endswith("param_map.net.4.bias"):
                p.normal_(0.0, 0.02)          # Glow's zero-initialised last convolution would make every coupling the identity
    gen = torch.Generator(device=device).manual_seed(1000 + rank)
    if config == "C4":
        x = torch.rand(B, 3, 32, 32, device=device, generator=gen)
        eps = [torch.randn(B, *q.loc.shape[1:], device=device, generator=gen) for q in model.q0]
        with torch.no_grad():
            model.log_prob(x[:2048])          # the first batch initialises the ActNorms (normalization.py:22-27)
    else:
        x = torch.randn(B, d, device=device, generator=gen)
        eps = torch.randn(B, d, device=device, generator=gen)
    evaluator = nf.ShardedEvaluator(model.log_prob)

    def step():
        stats = evaluator.reduce_stats(model.log_prob(x))
        z, lq = model.sample_from(eps)
        return stats, lq

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
    with torch.no_grad():
        for _ in range(warmup):
            step()
        events = []
        _lib.EVENT_SINK = events
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            stats, lq = step()
        fence()
        dt = max_over_ranks(time.perf_counter() - t0, device)
        _lib.EVENT_SINK = None
    nf.check_discriminant(device)
    # C4 (conditioner) and C5 (trunk hand-over, last layer) run on clamping split-half operands: nothing may have clamped
    saturated = nf.check_saturation(device, model=model)
    assert saturated == 0, "a split-half matrix path clamped values in %d workgroup(s)" % saturated
    assert torch.isfinite(stats).all() and torch.isfinite(lq).all()
    durs = [a.elapsed_time(b) * 1e-3 for a, b, t in events if t == tag]
    kern_s = sum(durs) / max(len(durs), 1)
    per_launch = work * B
    scale = 1e9 if unit == "GB/s" else 1e12
    out = None
    if rank == 0:
        out = {"metric": "flow transforms/sec (log_prob + sample), config %s" % config,
               "value": round(2.0 * B * world * steps / dt, 1), "unit": "transforms/s", "n_gpus": world,
               "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * dt / steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "config": {"workload": workload, "batch_per_gpu": B, "layers": layers},
               "roofline": {"bound": bound, "kernel": kname, "achieved": round(per_launch / kern_s / scale, 1) if durs else 0.0,
                            "peak": peak / scale, "unit": unit,
                            "frac": round(per_launch / kern_s / peak, 4) if durs else 0.0, "traffic": None,
                            "launches": len(durs), "avg_launch_ms": round(kern_s * 1e3, 4),
                            "algorithmic_work_per_launch": per_launch,
                            "note": ("algorithmic flop %d per sample-layer x %d layers per launch; peak = the mixture's ceiling "
                                     "(first layer %d flop at the fp32 matrix peak, the rest at the f16 peak / 3); HBM side "
                                     "of a launch: %d B per sample (%.1f us at 8 TB/s)"
                                     % (flop_sl, layers, 2 * 16 * 64, hbm_bytes, 1e6 * hbm_bytes * B / HBM_PEAK)) if config == "C2" else
                                    ("the GlowBlock conditioner (conv3x3 -> 1x1 -> nine tap matrices of the last conv3x3) as "
                                     "one launch: %.1f MFLOP per image and launch averaged over the three levels "
                                     "(%s), against the dense f16 matrix peak / 3 (split-half operands); it is ~3/4 of "
                                     "the step.  SURVEY 8d's coupling-only yardstick for C4 (688 512 B per transform) "
                                     "would be %.2f ms per step at 8 TB/s" % (
                                         work / 1e6, " / ".join("%.1f" % (f / 1e6) for f in flop_img),
                                         1e3 * 2 * 688512.0 * B / HBM_PEAK)) if config == "C4" else
                                    ("operator-boundary bytes %d per sample-layer (x + the [d_t, 3K-1] logits + y + log_q, "
                                     "SURVEY 8d) per launch of the last-layer + spline kernel; the logits never reach "
                                     "HBM in this path, so this is the yardstick, not the traffic; as matrix work the same "
                                     "launch is %.2f MFLOP per sample on split-half operands: %.3f of the f16 peak / 3"
                                     % (bytes_sl, 2 * 128 * 512 * 47 / 1e6,
                                        (2.0 * 128 * 512 * 47 * B / kern_s) / (MFMA_F16_PEAK / 3.0) if durs else 0.0))}}
        if config == "C5":
            out["whole_step_frac_of_yardstick"] = round(2 * layers * bytes_sl * B / HBM_PEAK / (dt / steps), 4)
        if world == 1 and cpu and not args.no_cpu_baseline and config == "C2":
            from helpers import oracle_affine_stack
            sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            stack = oracle_affine_stack(sd, layers, d)
            cores = cpu_share()
            torch.set_num_threads(cores)
            xc, ec = x.cpu(), eps.cpu()

            def run():
                t1 = time.perf_counter()
                with torch.no_grad():
                    stack.log_prob(xc)
                    stack.sample_from(ec)
                return time.perf_counter() - t1
            run()
            ts = sorted(run() for _ in range(3))
            out["cpu_baseline"] = {"value": round(2 * B / ts[1], 1), "unit": "transforms/s", "cores": cores, "kind": "port",
                                   "sample": "oracle C2 stack at the full batch %d, 1 warm-up + 3 timed runs, median %.2f s, "
                                             "torch CPU fp32, %d threads" % (B, ts[1], cores)}
    del model, x, eps
    torch.cuda.empty_cache()
    return out


def bench_train(args, device, rank, world, steps=None, warmup=None):
    """Training path of the headline configuration: Adam steps on forward_kld (normflow/core.py:33-45) of the C3 model,
    131 072 samples per GPU.  Data-parallel over ranks would add a gradient all-reduce; only N = 1 is reported here."""
    if world != 1:
        raise SystemExit("--config C3-train reports the single-GPU training step only")
    steps = steps or args.steps
    warmup = max(args.warmup if warmup is None else warmup, 1)
    B = 131072 if args.batch == 1 << 20 else args.batch
    torch.manual_seed(0)
    flows = [nf.flows.CoupledRationalQuadraticSpline(D, BLOCKS, HIDDEN, BINS, tail_bound=TAIL, reverse_mask=bool(i % 2),
                                                    num_context_channels=CTX) for i in range(LAYERS)]
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(D), flows).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    gen = torch.Generator(device=device).manual_seed(1000)
    x = torch.randn(B, D, device=device, generator=gen)
    ctx = torch.randn(B, CTX, device=device, generator=gen)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.forward_kld(x, context=ctx)
        loss.backward()
        opt.step()
        return loss
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # kernel durations for the roofline object: the same steps once more with an event pair around every weight-gradient
    # launch.  Not inside the timed region above: a step has 84 such launches of 20 - 180 us, and the event pairs
    # stretch it by ~10 % (28.2 against 25.3 ms) - unlike the headline leg, whose 120 launches are 1.2 ms each.
    events = []
    _lib.EVENT_SINK = events
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    _lib.EVENT_SINK = None
    assert torch.isfinite(loss)
    assert nf.check_saturation(device) == 0, "a split-half training kernel clamped a value"
    durs = [a.elapsed_time(b) * 1e-3 for a, b, t in events if t == "linear_wgrad"]
    kern_s = sum(durs) / max(len(durs), 1)
    # algorithmic flop of the weight gradients of one layer's dense layers, averaged over the launches of a layer
    flop_layer = 2.0 * B * ((D // 2 + CTX) * HIDDEN + 2 * BLOCKS * HIDDEN * HIDDEN + BLOCKS * CTX * HIDDEN
                            + HIDDEN * (D // 2) * (3 * BINS - 1))
    launches_layer = 1 + 2 * BLOCKS + BLOCKS + 1
    per_launch = flop_layer / launches_layer
    out = {
        "metric": "training samples/sec (Adam step on forward_kld), config C3", "value": round(B * steps / dt, 1),
        "unit": "samples/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": round(1e3 * dt / steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 results; the conditioner's forward products and the square layers' input gradients on fp16x3 split "
                 "operands (csrc/linear_f16x3.hip), weight gradients likewise (csrc/linear_wgrad.hip, split-half form) - error "
                 "against fp64 at or below the library's fp32 GEMM on every shape; remaining input gradients on the library's fp32 "
                 "GEMMs", "data": "synthetic",
        "config": {"workload": "C3 model (D=64, 12 RQ-spline couplings, 8 bins, cond_dim=16), Adam step on forward_kld, "
                               "batch=%d" % B, "batch_per_gpu": B, "layers": LAYERS},
        "roofline": {"bound": "mfma", "kernel": "linear_wgrad_kernel", "achieved": round(per_launch / kern_s / 1e12, 1) if durs else 0.0,
                     "peak": round(MFMA_F16_PEAK / 3.0 / 1e12, 1), "unit": "TFLOP/s",
                     "frac": round(per_launch / kern_s / (MFMA_F16_PEAK / 3.0), 4) if durs else 0.0, "traffic": None,
                     "launches": len(durs), "avg_launch_ms": round(kern_s * 1e3, 4),
                     "note": "weight / bias gradients of the conditioner's dense layers (split-half form: three f16 matrix "
                             "instructions per product, peak = dense f16 matrix peak / 3; batch reduction split over the chip; the "
                             "launches are memory-bound: 57 us floor for the 736-row layer at 131 072 samples), averaged over a layer's "
                             "five shapes, measured with HIP events in a second pass of the same steps (outside the timed region: an event "
                             "pair per launch stretches the step by ~10 %); the step also contains "
                             "the split-half forward / input-gradient kernel, library GEMMs for the other input gradients, the "
                             "spline forward / VJP kernels and fused elementwise maps"}}
    del model, opt, x, ctx
    torch.cuda.empty_cache()
    return out


def bench_small_batch(device, batch=2048, reps=30):
    """log_prob latency of the C3 model at the batch size of the reference's own drivers (/root/reference/run.py:45-47:
    1024 - 2048 samples): eager (an unchanged caller of NormalizingFlow.log_prob) and replayed from a HIP graph
    (nf.GraphedFlow).  At this size the 12 layers are ONE launch of the 32-sample-tile kernel
    (vcnf_rqs_stack_fused_f32, csrc/fused_layer_v6s.hip) + the fp32 re-evaluation launch + the Gaussian end cap."""
    torch.manual_seed(0)
    flows = [nf.flows.CoupledRationalQuadraticSpline(D, BLOCKS, HIDDEN, BINS, tail_bound=TAIL, reverse_mask=bool(i % 2),
                                                    num_context_channels=CTX) for i in range(LAYERS)]
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(D), flows).to(device).eval()
    gen = torch.Generator(device=device).manual_seed(2000)
    x = torch.randn(batch, D, device=device, generator=gen)
    ctx = torch.randn(batch, CTX, device=device, generator=gen)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    with torch.no_grad():
        eager = timed(lambda: model.log_prob(x, ctx))
        per_layer = None
        model.fuse_rqs_stacks = False
        per_layer = timed(lambda: model.log_prob(x, ctx))
        model.fuse_rqs_stacks = True
    g = nf.GraphedFlow(model, batch, CTX)
    graph = timed(lambda: g.log_prob(x, ctx))
    redone = nf.range_redo_count(device)
    nf.check_discriminant(device)
    out = {"metric": "log_prob latency, config C3 at the reference drivers' batch size", "unit": "ms",
           "higher_is_better": False, "batch": batch, "eager_ms": round(eager * 1e3, 4), "graph_ms": round(graph * 1e3, 4),
           "eager_one_launch_per_layer_ms": round(per_layer * 1e3, 4),
           "transforms_per_s_graph": round(batch / graph, 1), "dtype": "f32 results on fp16x3 split operands (see dtype)",
           "range_redo_tiles": int(redone),
           "config": {"workload": "C3 model, log_prob of %d samples, one launch for the 12 coupling layers" % batch,
                      "layers": LAYERS}}
    del model, g, x, ctx
    torch.cuda.empty_cache()
    return out


def extra_configs(args, device):
    """Short legs of BASELINE.json's other GPU configurations and of the training step, for the default JSON line
    (N = 1): 1 warm-up + 2 timed steps each, so that every configuration's figure is driver-visible."""
    out = {}
    for cfg in ("C2", "C4", "C5"):
        r = bench_other(args, device, 0, 1, config=cfg, steps=2, warmup=1, cpu=False)
        out[cfg] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline")
                    if k in r}
        if "whole_step_frac_of_yardstick" in r:
            out[cfg]["whole_step_frac_of_yardstick"] = r["whole_step_frac_of_yardstick"]
    r = bench_train(args, device, 0, 1, steps=2, warmup=2)
    out["C3-train"] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline")}
    out["C3-latency-2048"] = bench_small_batch(device)
    return out


DTYPE = {
    "fp16x3": ("f32 results on fp16x3 split operands: every fp32 operand of a conditioner GEMM travels as hi + lo fp16 "
               "halves, three f16 matrix instructions per product (four in the 48- and 16-deep layers), fp32 "
               "accumulation; measured against fp64 (tests/test_gpu_gemm_error.py): GEMM error 0.6-0.75x the exact-fp32 "
               "matrix path's on the first, hidden and last layers, 0.79-1.09x (mean; p99.9 and max below) on the 16-deep "
               "gate layer, 0.63x at the conditioner's output; tiles holding values beyond the fp16 range are "
               "re-evaluated on the exact fp32 path on the device (never clamped)"),
    "fp32": "f32 (exact fp32 matrix instructions v_mfma_f32_16x16x4_f32)",
    "split": "f32",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1 << 20, help="samples per GPU (weak scaling) or in total (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank holds --batch samples; strong: --batch samples in total, split over the ranks")
    ap.add_argument("--config", choices=["C3", "C2", "C4", "C5", "C3-train"], default="C3",
                    help="C3 (default) is the headline metric's configuration; C2 (D=32, 8 affine couplings, batch 262144) "
                         "C4 (3x32x32 multiscale Glow, 16384 images per GPU) and C5 (D=1024, 24 RQS couplings, 16 bins, the per-GPU shard 524288) are BASELINE.json's other "
                         "GPU configurations, reported with their own roofline; C3-train: Adam steps on forward_kld of the C3 model "
                         "at 131072 samples per GPU (training path, samples/s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs legs (C2 / C4 / C5 / C3-train) of the default line")
    ap.add_argument("--split", action="store_true",
                    help="three-step layers (gather kernel, torch GEMMs, spline kernel) instead of the fused kernel")
    ap.add_argument("--precision", choices=["fp16x3", "fp32"], default="fp16x3",
                    help="matrix path of the fused layer kernel whose figure is `value`: fp16 split-half operands with "
                         "fp32 accumulation and device-side fp32 re-evaluation of out-of-range tiles (default), or exact "
                         "fp32 matrix instructions; the other one is timed the same way and reported as other_matrix_path")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    # rehearsal on a box with fewer GPUs than ranks (VCNF_BENCH_BACKEND=gloo): ranks share devices
    backend = os.environ.get("VCNF_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # A launcher (torch.distributed.run) sets RANK / WORLD_SIZE / MASTER_*: the process group is then initialised for
    # EVERY world size, one rank included, so that `torchrun --nproc-per-node 1 bench.py --gpus 1` runs the same RCCL
    # calls (barrier fencing, the fp64 [sum log_prob, count] all-reduce, the MAX of the timings) as an N-rank job.
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if launched:
        # this pool's host driver supports dmabuf IPC only: without the variable RCCL's (and torch's) cross-process
        # buffer sharing fails with hipIpcGetMemHandle: invalid argument (environment note of the GPU boxes)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    nf.lib()
    if args.config == "C3-train":
        print(json.dumps(bench_train(args, device, rank, world)))
        return
    if args.config != "C3":
        out = bench_other(args, device, rank, world)
        if rank == 0:
            print(json.dumps(out))
        if launched:
            dist.barrier()
            dist.destroy_process_group()
        return

    model = build_model(device, seed=0)                      # replicated weights

    def route(split, precision):
        for f in model.flows:
            f.prqct.fused = not split
            f.prqct.fused_precision = precision
    from vcnf_amd.sharded import bench_shard, max_over_ranks
    B, seed = bench_shard(args.batch, args.scaling, rank, world)     # this rank's samples and data seed
    gen = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(B, D, device=device, generator=gen)
    ctx = torch.randn(B, CTX, device=device, generator=gen)
    eps = torch.randn(B, D, device=device, generator=gen)
    evaluator = nf.ShardedEvaluator(model.log_prob)

    def step():
        lp = model.log_prob(x, ctx)
        stats = evaluator.reduce_stats(lp)                   # the one collective: [sum log_prob, count]
        z, lq = model.sample_from(eps, ctx)
        return stats, z, lq

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(split, precision):
        """--warmup untimed steps, then exactly --steps steps between two fences; whole-job time = max over ranks.
        Returns (seconds, per-launch HIP-event durations of the layer kernel inside the region)."""
        route(split, precision)
        with torch.no_grad():
            for _ in range(args.warmup):
                step()
            events = []
            _lib.EVENT_SINK = events
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                stats, z, lq = step()
            fence()
            dt = time.perf_counter() - t0
            _lib.EVENT_SINK = None
        nf.check_discriminant(device)
        assert torch.isfinite(stats).all() and torch.isfinite(lq).all()
        return max_over_ranks(dt, device), [a.elapsed_time(b) * 1e-3 for a, b, _ in events]

    head = "split" if args.split else args.precision
    dt, durs = timed_region(args.split, args.precision)
    redone = nf.range_redo_count(device)                     # tiles the split-half kernel handed to the exact fp32 path
    with torch.no_grad():
        # the two directions on their own (outside the timed region): rates + harmonic combination, SURVEY 8d
        def timed(fn, n=2):
            fn()
            fence()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            fence()
            return (time.perf_counter() - t1) / n
        dt_lp = max_over_ranks(timed(lambda: evaluator.reduce_stats(model.log_prob(x, ctx))), device)
        dt_sm = max_over_ranks(timed(lambda: model.sample_from(eps, ctx)), device)

    total = torch.tensor([float(B)], dtype=torch.float64, device=device)     # samples of the whole job
    if dist.is_initialized():
        dist.all_reduce(total)
    total = float(total.item())

    def roofline_of(path, durs):
        """Dominant kernel of a timed region: mean launch duration (HIP events on the launch stream) against the
        ceiling of its matrix arithmetic (fused layers) or HBM (three-step layers)."""
        kern_s = sum(durs) / max(len(durs), 1)
        if path == "split":
            work, peak, unit, bound, kname = BYTES_PER_SAMPLE_LAYER * B, HBM_PEAK, "GB/s", "hbm", "rqs_coupling_pf_kernel"
            note = "algorithmic bytes 3464 B/sample-layer (x + params + y + logdet)"
        elif path == "fp32":
            work, peak, unit, bound, kname = FLOP_PER_SAMPLE_LAYER * B, MFMA_F32_PEAK, "TFLOP/s", "mfma", "fused_rqs_layer_kernel"
            note = ("algorithmic flop %d per sample-layer (conditioner GEMMs) on v_mfma_f32_16x16x4_f32; HBM side of "
                    "the same launch: %d B/sample-layer" % (FLOP_PER_SAMPLE_LAYER, 4 * D + 4 * CTX + 4 * D + 8))
        else:
            # split-half path: every product costs three f16 matrix instructions, so the ceiling for
            # algorithmic flop is a third of the dense f16 peak
            work, peak, unit, bound, kname = (FLOP_PER_SAMPLE_LAYER * B, MFMA_F16_PEAK / 3.0, "TFLOP/s", "mfma",
                                              "fused_rqs_layer_v6_kernel")
            note = ("algorithmic flop %d per sample-layer (conditioner GEMMs); peak = dense f16 matrix peak / 3 "
                    "(hi*hi + hi*lo + lo*hi per product, fp32 accumulation); HBM side of the same "
                    "launch: %d B/sample-layer" % (FLOP_PER_SAMPLE_LAYER, 4 * D + 4 * CTX + 4 * D + 8))
        achieved = work / kern_s if durs else 0.0
        scale = 1e9 if unit == "GB/s" else 1e12
        return {"bound": bound, "kernel": kname, "achieved": round(achieved / scale, 1), "peak": peak / scale, "unit": unit,
                "frac": round(achieved / peak, 4), "traffic": pmc_traffic(kname, B), "launches": len(durs),
                "avg_launch_ms": round(kern_s * 1e3, 4), "algorithmic_work_per_launch": work, "note": note}

    # the other matrix path of the fused kernel: same workload, same warm-up, same steps, same fences
    other = None
    if not args.split:
        alt = "fp32" if args.precision == "fp16x3" else "fp16x3"
        dt_alt, durs_alt = timed_region(False, alt)
        redone += nf.range_redo_count(device)
        other = {"matrix_path": alt, "dtype": DTYPE[alt], "value": round(2.0 * total * args.steps / dt_alt, 1),
                 "unit": "transforms/s", "ms_per_step": round(1e3 * dt_alt / args.steps, 3), "steps": args.steps,
                 "warmup": args.warmup, "roofline": roofline_of(alt, durs_alt)}
        route(args.split, args.precision)

    # the HBM-bound spline kernel on its own (outside the timed region): one layer evaluated
    # through the three-step path with the conditioner output materialised, spline launch timed
    hbm_side = None
    if rank == 0 and not args.split:
        lay = model.flows[0].prqct
        with torch.no_grad():
            params = lay._params(x, ctx, False)
            shared = lay.unconditional_transform.logits()
            tf, idf, cfg = lay._index32('tf'), lay._index32('id'), lay._cfg(True)
            for _ in range(2):
                _lib.rqs_coupling(x, params, tf, idf, shared, cfg, False)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _lib.rqs_coupling(x, params, tf, idf, shared, cfg, False)
            e1.record()
            torch.cuda.synchronize()
            t_sp = e0.elapsed_time(e1) * 1e-4
            del params
        hbm_side = {"kernel": "rqs_coupling_pf_kernel (split path, not in the timed region)",
                    "achieved": round(BYTES_PER_SAMPLE_LAYER * B / t_sp / 1e9, 1), "peak": HBM_PEAK / 1e9,
                    "unit": "GB/s", "frac": round(BYTES_PER_SAMPLE_LAYER * B / t_sp / HBM_PEAK, 4),
                    "frac_of_measured_copy_bw": round(BYTES_PER_SAMPLE_LAYER * B / t_sp / HBM_COPY, 4),
                    "avg_launch_ms": round(t_sp * 1e3, 4)}

    out = None
    if rank == 0:
        transforms = 2.0 * total * args.steps
        out = {
            "metric": "flow transforms/sec (log_prob + sample), D=64 RQ-spline, batch=1M",
            "value": round(transforms / dt, 1),
            "unit": "transforms/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": DTYPE[head],
            "data": "synthetic",
            "log_prob_rate": round(total / dt_lp, 1), "sample_rate": round(total / dt_sm, 1),
            "harmonic_rate": round(2.0 * total / (dt_lp + dt_sm), 1),
            "frac_of_hbm_roofline_end_to_end": round(2 * LAYERS * BYTES_PER_SAMPLE_LAYER * B / HBM_PEAK / (dt / args.steps), 4),
            "config": {"workload": "C3: conditional D=64 (cond_dim=16), 12 RQ-spline coupling layers "
                                   "(8 bins), batch=%d per GPU, log_prob + sample per step" % B,
                       "batch_per_gpu": B, "batch_total": int(total), "layers": LAYERS, "bins": BINS, "hidden": HIDDEN,
                       "matrix_path": head, "range_redo_tiles": int(redone),
                       "process_group": ("%s, %d rank(s)" % (dist.get_backend(), world)) if dist.is_initialized() else "none",
                       "sharding": "%s scaling: %s over %d GPU(s) (contiguous shards, weights replicated), one "
                                   "all-reduce of [sum log_prob, count] per log_prob" % (
                                       args.scaling, ("%d samples per GPU" % B) if args.scaling == "weak" else
                                       ("%d samples in total" % int(total)), world)},
            "roofline": roofline_of(head, durs),
        }
        if hbm_side is not None:
            out["roofline_hbm_spline_kernel"] = hbm_side
        if other is not None:
            out["other_matrix_path"] = other
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model)
    del model, x, ctx, eps, evaluator
    torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_extra:
            out["extra_configs"] = extra_configs(args, device)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if launched:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
