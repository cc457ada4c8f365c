#!/usr/bin/env python3
"""Timing sweep over kernel tuning variants (code = W*100 + KG*10 + UNR)."""
import os, subprocess, sys
code = r'''
import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
L, B, steps = int(sys.argv[1]), 65536, 40
lm, cmds = make_scenario(1234, L, 200)
f = S.BatchedEKF(B, L).readParams(); f.set_map(lm); f.init(0, 0, 0)
f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
f.run_sim(cmds[1:40]); f.sync()
best = 1e9
for rep in range(3):
    t0 = time.time(); f.run_sim(cmds[40:40 + steps]); f.sync(); best = min(best, time.time() - t0)
ab = f.algorithmic_bytes()
print(f"L={L} variant={os.environ.get('SLAM_WAVES_PER_FILTER','-')}: {best / steps * 1e3:.3f} ms/step  {B * steps / best / 1e6:.2f} M steps/s  {ab / (best / steps) / 1e12:.3f} TB/s", flush=True)
'''
sweeps = {50: sys.argv[1].split(",") if len(sys.argv) > 1 else ["444", "434", "238", "248", "842", "448", "424"],
          20: sys.argv[2].split(",") if len(sys.argv) > 2 else ["244", "444", "148", "248"]}
for L, vs in sweeps.items():
    for v in vs:
        if v:
            subprocess.run([sys.executable, "-c", code, str(L)], env=dict(os.environ, SLAM_WAVES_PER_FILTER=v))
