#!/usr/bin/env python3
"""Timing-only ablations of the step kernel (results are WRONG with dbg flags set; never used by tests/bench)."""
import os, subprocess, sys
code = r'''
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
L, B, steps = int(sys.argv[1]), 65536, 200
lm, cmds = make_scenario(1234, L, 1000)
f = S.BatchedEKF(B, L).readParams(); f.set_map(lm); f.init(0, 0, 0)
f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
f.run_sim(cmds[1:644]); f.sync()
dt = 1e9
for rep in range(1):   # (one pass: repeating the same commands drives the vehicle away from the map)
    t0 = time.time(); f.run_sim(cmds[644:644 + steps]); f.sync(); dt = min(dt, time.time() - t0)
print("flagged", int((f.status() != 0).sum()), "M", f.landmark_counts()[:4], end="  ")
print(f"L={L} dbg={os.environ.get('SLAM_DEBUG_FLAGS','0')} wpf={os.environ.get('SLAM_WAVES_PER_FILTER','-')}: {dt / steps * 1e3:.3f} ms/step", flush=True)
'''
for L in (50,):
    for dbg in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("0", "1", "2", "3")):
        env = dict(os.environ, SLAM_DEBUG_FLAGS=dbg)
        subprocess.run([sys.executable, "-c", code, str(L)], env=env)
