import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
L, N, B, KP = int(sys.argv[1]), 1000, 256, int(sys.argv[2])
lm, cmds = make_scenario(1234, L, N - 1)
pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=KP).readParams(solve_graph_every_iteration=True)
pg.set_map(lm); pg.set_seed(2025); pg.init(0.0, 0.0, 0.0)
import time
t0 = time.time(); c = pg.run_sim_every_iteration(cmds); dt = time.time() - t0
tl = pg.last_solve_timeline()[0]
tr = np.sort(c[:, 1]) / (N - 1)
print(f"L={L}: {dt:.2f} s, rounds {len(tl)} ({len(tl)/(N-1):.2f} per tick), mean listed {tl.mean():.1f}; trials per tick per graph: mean {tr.mean():.2f} median {np.median(tr):.2f} p90 {tr[int(.9*B)]:.2f} p99 {tr[int(.99*B)]:.2f} max {tr[-1]:.2f}")
print("listed per round, deciles of the run:", [int(tl[int(len(tl) * q / 10):int(len(tl) * (q + 1) / 10)].mean()) for q in range(10)])
