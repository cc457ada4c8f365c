#!/usr/bin/env python3
"""Per-trial, per-kernel HIP-event times of one profiled pose-graph batch solve (SLAM_PGS_TRACE + profiling: one group, kernels
back to back).  Columns: linearize chain syrk chol backsolve evaluate (+ decide)."""
import os, sys
os.environ["SLAM_PGS_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, N = 200, 1000
lm, cmds = make_scenario(1234, L, N - 1)
pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=32).readParams()
pg.set_map(lm); pg.set_seed(2025); pg.init(0.0, 0.0, 0.0)
pg.run_sim(cmds); pg.sync()
pg.solvePoseGraph(); pg.sync()
pg.set_profiling(True)
pg.solvePoseGraph(); pg.sync()
print(pg.last_solve_kernel_ms())
