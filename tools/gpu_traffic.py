#!/usr/bin/env python3
"""Run N steps of the L=50 B=65536 steady-state workload (for rocprofv3 --pmc passes)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
L, B, steps = 50, 65536, 20
lm, cmds = make_scenario(1234, L, 200)
f = S.BatchedEKF(B, L).readParams(); f.set_map(lm); f.init(0, 0, 0)
f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
f.run_sim(cmds[1:40 + steps]); f.sync()
print("done", f.landmark_counts().mean())
