"""Data-parallel evaluation: one process per GPU, the batch split contiguously
by rank, weights replicated, and a single all-reduce (RCCL over xGMI on the GPU
box, gloo in CPU tests) of the 2-element fp64 vector [sum log_prob, count].
Samples are never gathered.  The reference has no multi-GPU path;
this is new design around its log_prob / sample loops (normflow/core.py:144-183).
"""
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous [lo, hi) slice of ``total`` samples owned by ``rank``; the
    first ``total % world_size`` ranks take one extra sample."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d / world_size %d" % (rank, world_size))
    base, extra = divmod(int(total), world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def bench_shard(batch, scaling, rank, world_size):
    """(samples this rank holds, seed of its synthetic data) for bench.py: 'weak' - every rank holds ``batch``
    samples; 'strong' - ``batch`` samples in total, split by ``shard_bounds``.  The seed depends on the rank only,
    so a rank draws different data from every other rank in both modes."""
    if scaling == "weak":
        n = int(batch)
    elif scaling == "strong":
        lo, hi = shard_bounds(batch, world_size, rank)
        n = hi - lo
    else:
        raise ValueError("scaling must be 'weak' or 'strong'")
    if not (0 <= rank < world_size):
        raise ValueError("bad rank %d / world_size %d" % (rank, world_size))
    return n, 1000 + rank


def max_over_ranks(seconds, device, group=None):
    """Whole-job duration of a timed region: the slowest rank's (one all-reduce MAX of an fp64 scalar)."""
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():      # also with ONE rank: the collective path is the same code
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class ShardedEvaluator:
    """Wraps per-shard callables; the only collective is ``mean_log_prob``'s
    all-reduce.  ``log_prob_fn(x[, context]) -> [b]`` is normally
    ``NormalizingFlow.log_prob`` on this rank's GPU."""

    def __init__(self, log_prob_fn, sample_fn=None, group=None, micro_batch=None):
        """``micro_batch``: evaluate the shard in chunks of at most this many samples.  Layers on
        the three-step path materialise the conditioner output [b, d_t * (3K-1)] (config C5:
        96 KB per sample and layer), so a 512K-sample shard is walked in bounded pieces; the
        per-sample results are written into one preallocated [b] buffer."""
        self.log_prob_fn = log_prob_fn
        self.sample_fn = sample_fn
        self.group = group
        self.micro_batch = micro_batch

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def local_slice(self, total):
        w, r = self._world()
        return shard_bounds(total, w, r)

    def _call(self, x, context):
        return self.log_prob_fn(x) if context is None else self.log_prob_fn(x, context)

    def log_prob_shard(self, x_shard, context_shard=None):
        n = len(x_shard)
        mb = self.micro_batch
        if not mb or n <= mb:
            return self._call(x_shard, context_shard)
        out = None
        for lo in range(0, n, mb):
            part = self._call(x_shard[lo:lo + mb], None if context_shard is None else context_shard[lo:lo + mb])
            if out is None:
                out = torch.empty(n, dtype=part.dtype, device=part.device)
            out[lo:lo + mb] = part
        return out

    def reduce_stats(self, log_q):
        """[sum, count] over all ranks as fp64, one fused all-reduce."""
        # (no torch.tensor(..., device=...) here: that is a synchronous host -> device copy per call)
        stats = torch.empty(2, dtype=torch.float64, device=log_q.device)
        stats[0] = log_q.sum(dtype=torch.float64)
        stats[1] = float(log_q.numel())
        if dist.is_available() and dist.is_initialized():
            # issued for every world size, 1 included: a single-rank launch (torchrun --nproc-per-node 1) runs the
            # same RCCL call as N ranks, so the collective path is exercised on a one-GPU box too
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
        return stats

    def mean_log_prob(self, x_shard, context_shard=None):
        """Global mean log-probability (the negative of the forward-KL / NLL the
        reference's drivers monitor) from this rank's shard."""
        stats = self.reduce_stats(self.log_prob_shard(x_shard, context_shard))
        return stats[0] / stats[1]

    def sample_shard(self, total, *args):
        lo, hi = self.local_slice(total)
        return self.sample_fn(hi - lo, *args)
