"""Multi-GPU sharding of the instance batch.

Instances are independent given the shared map + command sequence, so the batch shards embarrassingly: rank r of
G owns the contiguous global instances [r*per_rank, (r+1)*per_rank) and keys its noise streams with GLOBAL
instance ids (slam_set_instance_offset), so results are identical for any G.  There is no per-step exchange; the
only collective is the end-of-run gather of per-instance error statistics (plotting_node.py:195-218 metric),
over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import numpy as np


def shard_range(global_batch, rank, world):
    """Contiguous block partition; the first (global_batch % world) ranks get one extra instance."""
    base, extra = divmod(int(global_batch), int(world))
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def gather_error_stats(local_err, dist=None, device=None):
    """All-gather per-instance average errors (ragged shards allowed). Returns the global float64 vector on every
    rank.  `dist` is torch.distributed (initialised) or None for a single process."""
    local_err = np.ascontiguousarray(local_err, dtype=np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_err
    import torch
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    n = torch.tensor([local_err.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    pad = max(sizes)
    buf = torch.zeros(pad, dtype=torch.float64, device=dev)
    buf[:local_err.shape[0]] = torch.from_numpy(local_err).to(dev)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return np.concatenate([o[:s].cpu().numpy() for o, s in zip(out, sizes)])


def reduce_summary(local_err, dist=None, device=None):
    """{sum, sum of squares, count} all-reduced -> (mean, std, count) of the per-instance average error."""
    local_err = np.asarray(local_err, dtype=np.float64)
    acc = np.array([local_err.sum(), (local_err ** 2).sum(), float(local_err.size)])
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch
        t = torch.from_numpy(acc).to(device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        acc = t.cpu().numpy()
    mean = acc[0] / acc[2]
    var = max(acc[1] / acc[2] - mean * mean, 0.0)
    return mean, var ** 0.5, int(acc[2])


class ShardedRun:
    """The N>1 path of bench.py, factored out so that the world-size-2 gloo test drives exactly this code with a CPU
    engine (tests/test_distributed_cpu.py): shard plan, barrier-bracketed timing with the MAX over ranks, and the one
    collective of the run (error statistics, after the timed region).

    `engine` protocol (BatchedEKF satisfies it): run_sim(cmds) enqueues timesteps, sync() waits for them,
    error_stats() -> per-instance average position error of the local shard."""

    def __init__(self, dist=None, device=None):
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        self.device = device

    def plan(self, batch, scaling):
        """(first global instance, local instances, global instances).  strong: `batch` is the GLOBAL batch, split
        contiguously (BASELINE configs[3]: 65536 -> 8192 per GPU at 8 GPUs); weak: `batch` instances on every rank."""
        if scaling == "strong":
            start, n = shard_range(batch, self.rank, self.world)
            return start, n, int(batch)
        if scaling == "weak":
            return self.rank * int(batch), int(batch), int(batch) * self.world
        raise ValueError("scaling must be 'strong' or 'weak'")

    def barrier(self, engine):
        engine.sync()
        if self.dist:
            self.dist.barrier()
            engine.sync()

    def timed(self, engine, fn):
        """Seconds of fn() bracketed by engine sync + barrier on both sides, MAX over ranks."""
        import time
        self.barrier(engine)
        t0 = time.perf_counter()
        fn()
        self.barrier(engine)
        wall = time.perf_counter() - t0
        if self.dist:
            import torch
            tw = torch.tensor([wall], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
            self.dist.all_reduce(tw, op=self.dist.ReduceOp.MAX)
            wall = float(tw.item())
        return wall

    def max_over_ranks(self, v):
        """MAX over the ranks of a per-rank scalar (e.g. a HIP-event duration): one small all-reduce, outside any timed region."""
        if not self.dist:
            return float(v)
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def error_statistics(self, engine):
        """(global per-instance errors, mean, std, count): the end-of-run gather + reduce (RCCL on the GPU box)."""
        err = engine.error_stats()
        allerr = gather_error_stats(err, self.dist, self.device)
        mean, std, n = reduce_summary(err, self.dist, self.device)
        return allerr, mean, std, n
