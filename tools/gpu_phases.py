#!/usr/bin/env python3
"""Steady-state per-phase shader-clock breakdown of the multi-step EKF kernel (thread-0 timers summed over the timesteps of one
launch, SLAM_DEBUG_FLAGS=4), bench trajectory, window t = 644 .. 743 (mean k 1.68).
usage: gpu_phases.py [f64|f32] [variant]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
names = {0: "init loads", 1: "pre-step(0)", 2: "wait for pre-step", 3: "group formation", 9: "pre-flush pass", 4: "thin gather", 5: "predict",
         6: "detections", 7: "pass / end barrier", 10: "end of step", 8: "epilogue",
         16: "C: loop overhead", 17: "C: group formation", 18: "C: flush wait + gather", 19: "C: prediction", 20: "C: scalar chain / upd",
         21: "C: ring slot wait", 22: "C: H P, K, x", 23: "C: thin downdates", 24: "C: end of step", 25: "C: pre-step(t+1)", 26: "C: wait for generator"}
dt = sys.argv[1] if len(sys.argv) > 1 else "f64"
if len(sys.argv) > 2:
    os.environ["SLAM_WAVES_PER_FILTER"] = sys.argv[2]
L, B, t0, steps = int(os.environ.get("PHASE_L", "50")), int(os.environ.get("PHASE_B", "65536")), int(os.environ.get("PHASE_T0", "644")), int(os.environ.get("PHASE_STEPS", "100"))
lm, cmds = make_scenario(1234, L, t0 + steps + 1)
f = S.BatchedEKF(B, L, dtype=S.F32 if dt == "f32" else S.F64).readParams(); f.set_map(lm); f.set_seed(2025); f.init(0, 0, 0)
f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
f.run_sim(cmds[1:t0]); f.sync()
f.set_debug_flags(4); f.k_histogram(reset=True)
t1 = time.time(); f.run_sim(cmds[t0:t0 + steps]); f.sync(); el = time.time() - t1
h = f.k_histogram().astype(float)
out = (C.c_ulonglong * 64)()
_lib.check(_lib.lib().slam_debug_read_prof(f.h, out))
tot = sum(out[i] for i in names)
print(f"{dt} {steps} steps from t={t0}: {el / steps * 1e3:.3f} ms/step with timers, mean k {(h * np.arange(8)).sum() / h.sum():.2f}; "
      f"cycles per workgroup-step {tot / B / steps:.0f}")
for i, nm in names.items():
    print(f"   {nm:20s} {out[i] / B / steps:9.0f} cycles per step  {100.0 * out[i] / tot:5.1f} %")
f.close()
