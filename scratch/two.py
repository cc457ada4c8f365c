import os, sys, threading, time, numpy as np
sys.path.insert(0, os.getcwd())
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
H = int(sys.argv[1]); T0 = int(sys.argv[2]); T1 = int(sys.argv[3]); B = int(sys.argv[4]) if len(sys.argv) > 4 else 256 // H
L, N = 200, 1000
lm, cmds = make_scenario(1234, L, N - 1)
hs = []
for k in range(H):
    pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=32).readParams(solve_graph_every_iteration=True)
    pg.set_map(lm); pg.set_seed(2025); pg.set_instance_offset(k * B); pg.init(0.0, 0.0, 0.0)
    if T0 > 0: pg.run_sim(cmds[:T0])      # build the graph to T0 poses without solving
    hs.append(pg)
for p in hs: p.sync()
t0 = time.time()
th = [threading.Thread(target=lambda p=p: p.run_sim_every_iteration(cmds[T0:T1])) for p in hs]
[t.start() for t in th]; [t.join() for t in th]
for p in hs: p.sync()
dt = time.time() - t0
print(f"{H} handles x {B} graphs, ticks {T0}..{T1}: {dt:.2f} s -> {H * B * (T1 - T0) / dt:.0f} graph-ticks/s", flush=True)
