#!/usr/bin/env python3
"""Wall-clock time of each of the first 16 timesteps inside one multi-step launch (SLAM_DEBUG_FLAGS=32)."""
import ctypes as C, os, sys
os.environ["SLAM_DEBUG_FLAGS"] = "32"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
L, B, steps = 50, 65536, 16
lm, cmds = make_scenario(1234, L, 200)
f = S.BatchedEKF(B, L).readParams(); f.set_map(lm); f.init(0, 0, 0)
f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
f.run_sim(cmds[1:40]); f.sync()
f.run_sim(cmds[40:40 + steps]); f.sync()
lib = _lib.lib()
lib.slam_debug_read_prof_raw.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros((B, 16), dtype=np.uint64)
lib.slam_debug_read_prof_raw(f.h, buf.ctypes.data_as(C.c_void_p))
st = buf.astype(np.int64)
d = np.diff(st, axis=1) / 100.0   # wall_clock64 ticks at 100 MHz -> microseconds
for name, sel in (("first round (b<1024)", slice(0, 1024)), ("middle blocks", slice(20000, 40000)), ("last blocks", slice(64000, 65536))):
    print(name, "step durations us (steps 1..15):", np.round(np.median(d[sel], axis=0), 1))
life = (st[:, 15] - st[:, 0]) / 100.0
print("median 15-step span us:", np.median(life), " total span ms:", (st[:, 15].max() - st[:, 0].min()) / 1e5)
