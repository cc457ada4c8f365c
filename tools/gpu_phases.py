#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of ONE timestep of the multi-step EKF kernel by detection count (thread-0 timers,
SLAM_DEBUG_FLAGS=4: every step overwrites the slots, so a launch that ENDS on timestep t leaves the breakdown of t).
The bench trajectory (seed 1234, L=50) has k = 3 for every instance at t = 44..50, k = 2 at t = 54..61, k = 1 at
t = 70..87 and k = 0 at t = 96..104."""
import ctypes as C, os, sys, time
os.environ["SLAM_DEBUG_FLAGS"] = "4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
names = ["init loads", "pre-step(0)", "association", "xpred+group", "thin gather", "predict", "detections", "bulk stream", "epilogue"]
dt = sys.argv[1] if len(sys.argv) > 1 else "f64"
if len(sys.argv) > 2:
    os.environ["SLAM_WAVES_PER_FILTER"] = sys.argv[2]
L, B = 50, 65536
lm, cmds = make_scenario(1234, L, 200)
f = S.BatchedEKF(B, L, dtype=S.F32 if dt == "f32" else S.F64).readParams(); f.set_map(lm); f.set_seed(2025); f.init(0, 0, 0)
f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
f.run_sim(cmds[1:40]); f.sync()
out = (C.c_ulonglong * 16)()
t = 40
for k, t_end in ((3, 49), (2, 59), (1, 80), (0, 100)):
    f.run_sim(cmds[t:t_end - 4]); f.sync()
    f.k_histogram(reset=True)
    t0 = time.time(); f.run_sim(cmds[t_end - 4:t_end + 1]); f.sync(); el = time.time() - t0
    t = t_end + 1
    h = f.k_histogram()
    _lib.check(_lib.lib().slam_debug_read_prof(f.h, out))
    tot = sum(out[2:8])
    print(f"{dt} step t={t_end} (k={k}; 5-step launch histogram {h[:5].tolist()}, {el / 5 * 1e3:.3f} ms/step): "
          f"cycles per workgroup-step {tot / B:.0f} (2.4 GHz: {tot / B / 2400:.1f} us)")
    for i in range(2, 8):
        print(f"   {names[i]:14s} {out[i] / B:9.0f} cycles  {100.0 * out[i] / tot:5.1f} %")
f.close()
