"""Experiment: several pose-graph handles (own stream each) solved concurrently from host threads vs solve groups inside one handle."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
L, N = 200, 1000
lm, cmds = make_scenario(1234, L, N - 1)
def make(B, off, G):
    pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=32).readParams()
    pg.set_groups(G)
    pg.set_map(lm); pg.set_seed(2025); pg.set_instance_offset(off); pg.init(0, 0, 0); pg.run_sim(cmds)
    return pg
for nh, B, G in ((4, 256, 1), (1, 1024, 4), (1, 1024, 1), (4, 64, 1), (1, 256, 4), (2, 512, 1), (1, 1024, 2)):
    hs = [make(B, i * B, G) for i in range(nh)]
    for h in hs: h.solvePoseGraph()
    reps = 3
    t0 = time.time()
    def work(h):
        for _ in range(reps): h.solvePoseGraph()
    th = [threading.Thread(target=work, args=(h,)) for h in hs]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.time() - t0
    print(f"{nh} handles x B={B} x groups={G}: {nh * B * reps / dt:.0f} solves/s ({dt / reps * 1e3:.0f} ms per round)", flush=True)
    for h in hs: h.close()
