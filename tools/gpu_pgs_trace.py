#!/usr/bin/env python3
"""Trial-by-trial trace of one pose-graph batch solve (SLAM_PGS_TRACE): active instances and lambda lanes per trial, wall time.
usage: gpu_pgs_trace.py [B] [lanes]"""
import os, sys, time
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if len(sys.argv) > 2: os.environ["SLAM_PGS_LANES"] = sys.argv[2]
os.environ["SLAM_PGS_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
L, N = 200, 1000
lm, cmds = make_scenario(1234, L, N - 1)
pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=32).readParams()
pg.set_map(lm); pg.set_seed(2025); pg.init(0.0, 0.0, 0.0)
pg.run_sim(cmds); pg.sync()
for rep in range(2):
    t0 = time.perf_counter(); pg.solvePoseGraph(); pg.sync(); dt = time.perf_counter() - t0
    st = pg.stats()
    print(f"solve {rep}: {dt * 1e3:.1f} ms, {B / dt:.0f} solves/s, trials launched {pg.last_trials() if hasattr(pg, 'last_trials') else '?'}, "
          f"per-instance trials mean {st['trials'].mean():.1f} max {st['trials'].max()}, iterations mean {st['iterations'].mean():.1f}", flush=True)
