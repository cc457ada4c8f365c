"""world_size-2 gloo test of the N>1 path: shard ranges, global-instance-keyed noise streams and the
end-of-run error-statistics gather.  The per-rank compute is done by the CPU oracle here (no GPU)."""
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


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    lm, cmds = make_scenario(1234, 20, 120)
    start, n = shard_range(B, rank, world)
    r = O.run_ekf_batch(lm, cmds, n, 20, seed=9, inst0=start, nthreads=1, want_P=False)
    allerr = gather_error_stats(r["avg_err"], dist)
    summ = reduce_summary(r["avg_err"], dist)
    if rank == 0:
        q.put((allerr, summ))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_equals_single_process():
    B = 21  # ragged split 11 + 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    allerr, summ = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import oracle as O
    lm, cmds = make_scenario(1234, 20, 120)
    ref = O.run_ekf_batch(lm, cmds, B, 20, seed=9, inst0=0, nthreads=2, want_P=False)
    assert np.array_equal(allerr, ref["avg_err"])            # identical regardless of the sharding
    assert summ[2] == B and abs(summ[0] - ref["avg_err"].mean()) < 1e-15
    assert abs(summ[1] - ref["avg_err"].std()) < 1e-12
