#!/usr/bin/env python3
"""PCIe-inclusive rate of the per-step boundary: slam_step with HOST measurement buffers ([B][k][3] f32 + counts copied
to the device every step) vs slam_step_dev (same buffers already resident) vs slam_step_sim (generated on the device),
L=50, batch=65536, steady state.  The headline `value` of bench.py never includes this copy."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario

L, B, K, KS = 50, 65536, 8, 8
lm, cmds = make_scenario(1234, L, 64)
f = S.BatchedEKF(B, L).readParams(); f.set_map(lm); f.init(0, 0, 0)
f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
f.run_sim(cmds[1:20]); f.sync()
# record K steps of generated measurements (the generator's own output, so ids / ranges are consistent with the state)
rec = []
for t in range(20, 20 + K):
    f.update_sim(cmds[t]); rec.append(f.last_meas(KS))
f.sync()
hip = C.CDLL("libamdhip64.so")
def run(label, fn):
    f.sync(); t0 = time.perf_counter()
    for i in range(K): fn(i)
    f.sync(); dt = time.perf_counter() - t0
    print(f"{label:34s} {dt / K * 1e3:7.3f} ms/step  {B * K / dt / 1e6:6.2f} M steps/s")
run("slam_step_sim (device generator)", lambda i: f.update_sim(cmds[28 + i]))
run("slam_step (host buffers, H2D/step)", lambda i: f.update(cmds[36 + i], rec[i][0], rec[i][1]))
dm, dc = C.c_void_p(), C.c_void_p()
hip.hipMalloc(C.byref(dm), B * KS * 12); hip.hipMalloc(C.byref(dc), B * 4)
hip.hipMemcpy(dm, rec[0][0].ctypes.data_as(C.c_void_p), B * KS * 12, 1); hip.hipMemcpy(dc, rec[0][1].ctypes.data_as(C.c_void_p), B * 4, 1)
run("slam_step_dev (resident buffers)", lambda i: f.update_dev(cmds[44 + i], dm.value, dc.value, KS))
print(f"copied per step: {B * KS * 12 / 1e6:.1f} MB measurements + {B * 4 / 1e6:.2f} MB counts (pageable host memory)")
f.close()
