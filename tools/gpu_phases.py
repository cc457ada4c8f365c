#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the step kernel (lane-0 timers, SLAM_DEBUG_FLAGS=4)."""
import ctypes as C, os, sys, time
os.environ["SLAM_DEBUG_FLAGS"] = "4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
names = ["init loads", "sim/meas", "association", "xpred+group", "thin gather", "predict", "detections", "bulk stream", "epilogue"]
for L in (50, 20):
    B, steps = 65536, 30
    lm, cmds = make_scenario(1234, L, 200)
    f = S.BatchedEKF(B, L).readParams(); f.set_map(lm); f.init(0, 0, 0)
    f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
    f.run_sim(cmds[1:40]); f.sync()
    out = (C.c_ulonglong * 16)()
    _lib.lib().slam_debug_read_prof(f.h, out)
    t0 = time.time(); f.run_sim(cmds[40:40 + steps]); f.sync(); dt = time.time() - t0
    _lib.lib().slam_debug_read_prof(f.h, out)
    tot = sum(out[:9]); steps_n = 1  # timers hold the last launch only
    print(f"L={L}: {dt / steps * 1e3:.3f} ms/step; mean cycles per block-step {tot / B:.0f}")
    for i, nm in enumerate(names):
        print(f"   {nm:14s} {out[i] / B:9.0f} cycles  {100.0 * out[i] / tot:5.1f} %")
    f.close()
