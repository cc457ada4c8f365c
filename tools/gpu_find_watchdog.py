#!/usr/bin/env python3
"""Find the launch at which an EKF instance raises SLAM_INST_WATCHDOG in a scenario (debug): runs the scenario in launches of `chunk`
timesteps and prints the first launch after which any status flag is set, with the oracle's detection counts around it.
usage: gpu_find_watchdog.py L T B seed scenario inst0 f32 idknown wide chunk"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
a = sys.argv[1:]
L, T, B, seed, sc, inst0 = (int(v) for v in a[:6]); f32 = a[6] == "True"; idknown = int(a[7]); wide = a[8] == "True"; chunk = int(a[9])
os.environ["SLAM_RUN_CHUNK"] = str(chunk)
lm, cmds = make_scenario(sc, L, T)
cfg = S.default_config(); cfg.landmark_id_is_known = idknown
if wide: cfg.range_max = 1e9; cfg.fov_min = -4.0; cfg.fov_max = 4.0
f = S.BatchedEKF(B, L, dtype=S.F32 if f32 else S.F64).readParams(cfg)
f.set_map(lm); f.set_seed(seed); f.set_instance_offset(inst0); f.init(0, 0, 0)
t = 0
while t < T:
    f.run_sim(cmds[t:t + chunk]); t += chunk
    st = f.status()
    if st.any():
        print(f"flags {st.tolist()} after the launch of timesteps [{t - chunk}, {t})", "timesteps reached:", [f.get_state(b)['timestep'] for b in range(B)], "M:", f.landmark_counts().tolist())
        break
else:
    print("no flag")
f.close()
