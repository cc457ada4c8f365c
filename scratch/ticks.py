import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
L, N, B = 200, 1000, 256
lm, cmds = make_scenario(1234, L, N - 1)
pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=32).readParams(solve_graph_every_iteration=True)
pg.set_map(lm); pg.set_seed(2025); pg.init(0.0, 0.0, 0.0)
edges = [0, 16, 32, 64, 128, 256, 384, 512, 640, 768, 896, 999]
print("ticks        seconds   ms/tick   trials launched/tick   trials consumed/tick (mean)   landmarks mapped (mean)")
for a, b in zip(edges, edges[1:]):
    t0 = time.time(); c = pg.run_sim_every_iteration(cmds[a:b]); pg.sync(); dt = time.time() - t0
    ph = pg.last_iter_phases()
    M = np.mean([pg.get_graph(i, 1)["M"] for i in range(0, B, 32)])
    print(f"{a:4d}-{b:4d}   {dt:8.3f}  {dt / (b - a) * 1e3:8.2f}   {ph['trials_launched'] / (b - a):8.1f}              {c[:, 1].mean() / (b - a):8.2f}                      {M:6.1f}", flush=True)
