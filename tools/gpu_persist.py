#!/usr/bin/env python3
"""Timing of slam_run_sim at the headline size for several timesteps-per-launch settings (SLAM_RUN_CHUNK)."""
import os, subprocess, sys
code = r'''
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
L, B, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lm, cmds = make_scenario(1234, L, 400)
f = S.BatchedEKF(B, L).readParams(); f.set_map(lm); f.init(0, 0, 0)
f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
f.run_sim(cmds[1:40]); f.sync()
best = 1e9
for rep in range(3):
    t0 = time.time(); f.run_sim(cmds[40:40 + steps]); f.sync(); best = min(best, time.time() - t0)
ab = f.algorithmic_bytes()
print(f"L={L} B={B} chunk={os.environ.get('SLAM_RUN_CHUNK','all')} wpf={os.environ.get('SLAM_WAVES_PER_FILTER','-')}: {best / steps * 1e3:.3f} ms/step  {B * steps / best / 1e6:.2f} M steps/s  {ab / (best / steps) / 1e12:.3f} TB/s  flags={int((f.status()!=0).sum())}", flush=True)
'''
L = sys.argv[1] if len(sys.argv) > 1 else "50"
B = sys.argv[2] if len(sys.argv) > 2 else "65536"
steps = sys.argv[3] if len(sys.argv) > 3 else "100"
chunks = sys.argv[4].split(",") if len(sys.argv) > 4 else ["1", "0", "10", "1", "0"]
for c in chunks:
    subprocess.run([sys.executable, "-c", code, L, B, steps], env=dict(os.environ, SLAM_RUN_CHUNK=c))
