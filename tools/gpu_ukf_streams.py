"""Experiment: UKF batch split over several handles/streams driven by host threads."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
L, T0, T1 = 20, 30, 100
lm, cmds = make_scenario(1234, L, 1 + T0 + T1)
def make(B, off):
    f = S.BatchedUKF(B, L).readParams(); f.set_map(lm); f.set_seed(2025); f.set_instance_offset(off); f.init(0, 0, 0)
    f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
    f.run_sim(cmds[1:1 + T0]); f.sync()
    return f
for nh, B in ((1, 4096), (2, 2048), (4, 1024), (8, 512)):
    hs = [make(B, i * B) for i in range(nh)]
    def work(h):
        h.run_sim(cmds[1 + T0:]); h.sync()
    t0 = time.time()
    th = [threading.Thread(target=work, args=(h,)) for h in hs]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.time() - t0
    print(f"{nh} handles x B={B}: {nh * B * T1 / dt / 1e6:.3f} M steps/s ({dt / T1 * 1e3:.3f} ms/step)", flush=True)
    for h in hs: h.close()
