#!/usr/bin/env python3
"""Per-phase breakdown of the fp32-storage step kernel vs the fp64 one (lane-0 timers of the last single-step launch)."""
import ctypes as C, os, sys, time
os.environ["SLAM_DEBUG_FLAGS"] = "4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
names = ["init loads", "sim/meas", "association", "xpred+group", "thin gather", "predict", "detections", "bulk stream", "epilogue"]
L, B = 50, 65536
lm, cmds = make_scenario(1234, L, 200)
for dt_ in (S.F64, S.F32):
    f = S.BatchedEKF(B, L, dtype=dt_).readParams(); f.set_map(lm); f.init(0, 0, 0)
    f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
    for t in range(1, 40): f.update_sim(cmds[t])
    f.sync()
    t0 = time.time()
    for t in range(40, 60): f.update_sim(cmds[t])
    f.sync(); dt = time.time() - t0
    out = (C.c_ulonglong * 16)()
    _lib.lib().slam_debug_read_prof(f.h, out)
    tot = sum(out[:9])
    print(f"dtype={'f64' if dt_ == S.F64 else 'f32'}: {dt / 20 * 1e3:.3f} ms/step (one launch per step); mean cycles per block-step {tot / B:.0f}")
    for i, nm in enumerate(names):
        print(f"   {nm:14s} {out[i] / B:9.0f} cycles  {100.0 * out[i] / tot:5.1f} %")
    f.close()
