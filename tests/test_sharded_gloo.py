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
    if rank == 0:
        torch.save({"mean": mean, "stats": stats, "span": (lo, hi)}, out)
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
