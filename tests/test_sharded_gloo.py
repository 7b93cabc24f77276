"""N>1 path on CPU: two gloo ranks each evaluate their contiguous shard and the
single [sum, count] all-reduce yields the same mean log-prob as one process over
the whole batch.  The per-shard compute here is the oracle C3 stack (tests may
use the oracle as a stand-in compute function; the product kernels need a GPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import vcnf_amd as nf
from helpers import fixture, T, state_for, oracle_c3_stack


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    fx = fixture("g5_c3_stack")
    sd, _ = state_for(fx, "c3", 501, final_gain=1.0)
    stack = oracle_c3_stack(sd)
    x, ctx = T(fx["x"])[:130], T(fx["ctx"])[:130]      # 130: uneven split over 2 ranks is exercised below
    ev = nf.ShardedEvaluator(lambda a, c: stack.log_prob(a, c))
    lo, hi = ev.local_slice(129)
    mean = ev.mean_log_prob(x[lo:hi], ctx[lo:hi])
    stats = ev.reduce_stats(stack.log_prob(x[lo:hi], ctx[lo:hi]))
    # bench.py's sharding and timing logic (vcnf_amd.sharded.bench_shard / max_over_ranks)
    from vcnf_amd.sharded import bench_shard, max_over_ranks
    weak, strong = bench_shard(1001, "weak", rank, world), bench_shard(1001, "strong", rank, world)
    tmax = max_over_ranks(0.25 + rank, torch.device("cpu"))          # rank 1 is the slow one: 1.25 s
    counts = torch.tensor([float(strong[0])], dtype=torch.float64)
    dist.all_reduce(counts)
    if rank == 0:
        torch.save({"mean": mean, "stats": stats, "span": (lo, hi), "weak": weak, "strong": strong, "tmax": tmax,
                    "strong_total": float(counts)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_mean_log_prob(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    fx = fixture("g5_c3_stack")
    want = fx["c3/lp64"][:129]
    assert got["span"] == (0, 65)                       # 129 = 65 + 64
    assert float(got["stats"][1]) == 129.0
    assert abs(float(got["mean"]) - want.mean()) <= 1e-5 * abs(want.mean())
    assert abs(float(got["stats"][0]) - want.sum()) <= 1e-5 * abs(want.sum())
    # bench.py: weak scaling keeps the per-rank batch, strong scaling splits it (1001 = 501 + 500); seeds differ by rank;
    # the whole-job time is the slowest rank's
    assert got["weak"] == (1001, 1000) and got["strong"] == (501, 1000)
    assert got["strong_total"] == 1001.0 and got["tmax"] == 1.25


def test_bench_shard_single_process():
    from vcnf_amd.sharded import bench_shard, max_over_ranks
    assert bench_shard(1 << 20, "weak", 3, 8) == (1 << 20, 1003)
    assert sum(bench_shard(1 << 20, "strong", r, 8)[0] for r in range(8)) == 1 << 20
    assert bench_shard(10, "strong", 7, 8)[0] == 1 and bench_shard(10, "strong", 0, 8)[0] == 2
    assert max_over_ranks(0.5, torch.device("cpu")) == 0.5      # no process group: the local time
    import pytest
    with pytest.raises(ValueError):
        bench_shard(8, "diagonal", 0, 1)
