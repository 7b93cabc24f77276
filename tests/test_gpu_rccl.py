"""The RCCL branch on the hardware that exists: a ONE-rank "nccl" (= RCCL on ROCm) process group
in this process - init_process_group(device_id=...), barrier, the fp64 [sum log_prob, count] all-reduce of
ShardedEvaluator on a HIP tensor, max_over_ranks - around the product kernels.  No N > 1 RCCL run exists: this pool
leases one GPU; the two-rank logic is covered on gloo (tests/test_sharded_gloo.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import vcnf_amd as nf
from helpers import fixture, T, state_for

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_all_reduce_through_sharded_evaluator(hip):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # this pool's driver supports dmabuf IPC only
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=hip)
    try:
        assert dist.get_backend() == "nccl"
        fx = fixture("g5_c3_stack")
        sd, _ = state_for(fx, "c3", 501, final_gain=1.0)
        flows = [nf.flows.CoupledRationalQuadraticSpline(64, 2, 128, 8, reverse_mask=bool(i % 2), num_context_channels=16)
                 for i in range(12)]
        model = nf.NormalizingFlow(nf.distributions.DiagGaussian(64), flows)
        model.load_state_dict(sd)
        model = model.to(hip).eval()
        x, ctx = T(fx["x"]).to(hip), T(fx["ctx"]).to(hip)
        ev = nf.ShardedEvaluator(model.log_prob)
        lo, hi = ev.local_slice(len(x))
        assert (lo, hi) == (0, len(x))
        dist.barrier()
        with torch.no_grad():
            lp = model.log_prob(x, ctx)
            stats = ev.reduce_stats(lp)                       # RCCL all-reduce (SUM) of a 2-element fp64 HIP tensor
            mean = ev.mean_log_prob(x, ctx)
        t = nf.sharded.max_over_ranks(0.125, hip)             # RCCL all-reduce (MAX)
        dist.barrier()
        torch.cuda.synchronize()
        want = fx["c3/lp64"]
        assert stats.dtype == torch.float64 and stats.is_cuda
        assert float(stats[1]) == float(len(x))
        assert abs(float(stats[0]) - want.sum()) <= 1e-5 * abs(want.sum())
        assert abs(float(mean) - want.mean()) <= 1e-5 * abs(want.mean())
        assert t == 0.125
    finally:
        dist.destroy_process_group()
