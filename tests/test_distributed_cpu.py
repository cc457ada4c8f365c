"""world_size-2 gloo tests of the N>1 path: they drive parallel.ShardedRun, the code bench.py runs under torch.distributed
(shard plan for strong / weak scaling, global-instance-keyed noise streams, barrier-bracketed timing with the MAX over
ranks, the end-of-run error-statistics gather), with the CPU oracle as the per-rank engine (no GPU here)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from live_ekf_slam_amd.parallel import gather_error_stats, reduce_summary, shard_range
from live_ekf_slam_amd.scenario import make_scenario


def test_shard_range_partitions():
    for B, G in [(65536, 8), (10, 3), (7, 8), (1, 1)]:
        spans = [shard_range(B, r, G) for r in range(G)]
        assert spans[0][0] == 0 and sum(n for _, n in spans) == B
        for (s0, n0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + n0 == s1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class OracleEngine:
    """CPU stand-in for BatchedEKF behind the engine protocol of parallel.ShardedRun (run_sim / sync / error_stats): the
    per-rank compute is the oracle, everything around it (shard plan, barrier-bracketed timing with MAX over ranks, the
    end-of-run gather and reduce) is the code bench.py runs on the GPUs."""

    def __init__(self, lm, L, first, n, seed):
        self.lm, self.L, self.first, self.n, self.seed = lm, L, first, n, seed
        self.cmds = np.zeros((0, 2), np.float32)
        self.result = None

    def run_sim(self, cmds):
        self.cmds = np.concatenate([self.cmds, np.asarray(cmds, np.float32).reshape(-1, 2)])
        self.result = None

    def sync(self):
        if self.result is None and len(self.cmds):
            from oracle import oracle as O
            self.result = O.run_ekf_batch(self.lm, self.cmds, self.n, self.L, seed=self.seed, inst0=self.first, nthreads=1, want_P=False)

    def error_stats(self):
        self.sync()
        return self.result["avg_err"]


def _worker(rank, world, port, B, scaling, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from live_ekf_slam_amd.parallel import ShardedRun
    lm, cmds = make_scenario(1234, 20, 120)
    run = ShardedRun(dist)
    first, n, total = run.plan(B, scaling)
    eng = OracleEngine(lm, 20, first, n, seed=9)
    eng.run_sim(cmds[:100])                                   # "pre-roll + warm-up"
    wall = run.timed(eng, lambda: eng.run_sim(cmds[100:]))    # the timed window: barriers both sides, MAX over ranks
    allerr, mean, std, cnt = run.error_statistics(eng)        # the one collective of the run
    if rank == 0:
        q.put((allerr, (mean, std, cnt), total, wall))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(B, scaling):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, scaling, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=900)   # the first `import torch` of a cold container can take minutes
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_two_rank_strong_shard_equals_single_process():
    """bench.py's default for --gpus N: the GLOBAL batch split contiguously (ragged here: 11 + 10)."""
    B = 21
    allerr, summ, total, wall = _run_two_ranks(B, "strong")
    from oracle import oracle as O
    lm, cmds = make_scenario(1234, 20, 120)
    ref = O.run_ekf_batch(lm, cmds, B, 20, seed=9, inst0=0, nthreads=2, want_P=False)
    assert total == B and wall > 0
    assert np.array_equal(allerr, ref["avg_err"])            # identical regardless of the sharding
    assert summ[2] == B and abs(summ[0] - ref["avg_err"].mean()) < 1e-15
    assert abs(summ[1] - ref["avg_err"].std()) < 1e-12


def test_two_rank_weak_scaling_covers_disjoint_global_instances():
    """--scaling weak: B instances on every rank, global ids rank*B ..: the gathered vector equals one 2B-instance run."""
    B = 6
    allerr, summ, total, wall = _run_two_ranks(B, "weak")
    from oracle import oracle as O
    lm, cmds = make_scenario(1234, 20, 120)
    ref = O.run_ekf_batch(lm, cmds, 2 * B, 20, seed=9, inst0=0, nthreads=2, want_P=False)
    assert total == 2 * B and summ[2] == 2 * B
    assert np.array_equal(allerr, ref["avg_err"])


def test_strong_scaling_plan_at_eight_gpus_of_the_baseline_batch():
    """BASELINE configs[3]: 65 536 instances over 8 GPUs.  The plan ShardedRun makes for every rank gives eight contiguous 8 192-shards;
    an engine that runs a shard with that plan's first global instance produces, at both ends of every shard, exactly the errors of those
    GLOBAL instances in a single-process run (the noise streams are keyed by global id) - checked with the oracle on the two first and two
    last instances of each shard (the full 65 536 x 120 steps would be minutes of CPU)."""
    from live_ekf_slam_amd.parallel import ShardedRun
    from oracle import oracle as O

    class FakeDist:   # what ShardedRun reads of torch.distributed
        def __init__(self, rank, world): self.rank, self.world = rank, world
        def is_initialized(self): return True
        def get_world_size(self): return self.world
        def get_rank(self): return self.rank

    B, G = 65536, 8
    lm, cmds = make_scenario(1234, 20, 120)
    ends = []
    for r in range(G):
        first, n, total = ShardedRun(FakeDist(r, G)).plan(B, "strong")
        assert (first, n, total) == (r * 8192, 8192, B)
        for lo in (first, first + n - 2):          # the two first and the two last instances of the shard, run AS a shard at that offset
            eng = OracleEngine(lm, 20, lo, 2, seed=9)
            eng.run_sim(cmds)
            ends.append((lo, eng.error_stats()))
    # single process: the same global instances from ONE run whose first instance is 0 (ids 0, 1) or any other offset
    for lo, err in ends:
        ref = O.run_ekf_batch(lm, cmds, 4, 20, seed=9, inst0=lo - 1 if lo > 0 else 0, nthreads=1, want_P=False)["avg_err"]
        sel = ref[1:3] if lo > 0 else ref[0:2]
        assert np.array_equal(err, sel), lo
    assert len({lo for lo, _ in ends}) == 2 * G and np.unique(np.concatenate([e for _, e in ends])).size == 4 * G   # distinct streams
    # weak scaling at N = 8: 65 536 per rank, disjoint ranges
    for r in range(G):
        assert ShardedRun(FakeDist(r, G)).plan(B, "weak") == (r * B, B, G * B)
