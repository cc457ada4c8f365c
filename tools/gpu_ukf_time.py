#!/usr/bin/env python3
"""UKF-SLAM timing (BASELINE configs[2]: batch 4096, L=20) plus L=50, with the CPU oracle beside it."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
from oracle import oracle as O
cfgs = ((20, 4096, 40), (50, 4096, 20), (20, 65536, 10))
if len(sys.argv) > 1:
    cfgs = tuple(c for c in cfgs if str(c[0]) in sys.argv[1].split(","))
for L, B, steps in cfgs:
    lm, cmds = make_scenario(1234, L, 120)
    f = S.BatchedUKF(B, L).readParams(); f.set_map(lm); f.init(0, 0, 0)
    f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
    f.run_sim(cmds[1:20]); f.sync()
    t0 = time.time(); f.run_sim(cmds[20:20 + steps]); f.sync(); dt = time.time() - t0
    n = 4 + 2 * L
    print(f"UKF tpb={os.environ.get('SLAM_UKF_TPB','-')} L={L} B={B}: {dt / steps * 1e3:.3f} ms/step  {B * steps / dt / 1e3:.1f} k steps/s  M mean {f.landmark_counts().mean():.1f} flags {np.unique(f.status())} err {f.error_stats().mean():.4f}", flush=True)
    f.close()
if len(sys.argv) > 2: sys.exit(0)
vis = np.tile([3.0, -1.57, 1.57], (60, 1)); vis[0] = [1e9, -4.0, 4.0]
for L in (20, 50):
    lm, cmds = make_scenario(1234, L, 60)
    nc = os.cpu_count()
    r = O.run_ukf_batch(lm, cmds, 2 * nc, L, nthreads=nc, want_P=False, vision=vis)
    r1 = O.run_ukf_batch(lm, cmds, 2, L, nthreads=1, want_P=False, vision=vis)
    print(f"oracle UKF L={L}: {2 * nc * 60 / r['seconds']:.0f} steps/s on {nc} cores; {2 * 60 / r1['seconds']:.0f} steps/s on 1 core", flush=True)
