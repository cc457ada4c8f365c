#!/usr/bin/env python3
"""Where inside a multi-step launch the time goes, by STEP INDEX (SLAM_DEBUG_FLAGS=32: every workgroup stamps the 100 MHz wall clock at the
end of each timestep): mean us per workgroup for step 1, 2, ... of the launch, for the workgroups of the first round (ids < 1024) and the
rest, and the span of the launch.  usage: gpu_step_profile.py [steps] [t0]"""
import ctypes as C, os, sys
os.environ["SLAM_DEBUG_FLAGS"] = "32"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t0 = int(sys.argv[2]) if len(sys.argv) > 2 else 644
L, B = 50, 65536
lm, cmds = make_scenario(1234, L, t0 + steps + 1)
f = S.BatchedEKF(B, L).readParams(); f.set_map(lm); f.set_seed(2025); f.init(0, 0, 0)
f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
f.run_sim(cmds[1:t0]); f.sync()
f.run_sim(cmds[t0:t0 + steps]); f.sync()
lib = _lib.lib(); lib.slam_debug_read_prof_raw.argtypes = [C.c_void_p, C.c_void_p]
buf = np.zeros((B, 128), dtype=np.uint64)
_lib.check(lib.slam_debug_read_prof_raw(f.h, buf.ctypes.data_as(C.c_void_p)))
st = (buf[:, :steps] >> np.uint64(4)).astype(np.int64)
d = np.diff(st, axis=1) / 100.0
span = (st[:, steps - 1].max() - st[:, 0].min()) / 100.0
print(f"steps={steps} t0={t0}: launch span (first stamp to last stamp) {span / 1e3:.2f} ms; a workgroup from its first to its last stamp: {((st[:, -1] - st[:, 0]) / 100.0).mean():.1f} us")
first = np.arange(B) < 1024
print("step index: mean us per workgroup-step (first round | later rounds)")
for s in range(d.shape[1]):
    print(f"  {s + 1:3d}: {d[first, s].mean():7.2f} | {d[~first, s].mean():7.2f}")
# the gap between a workgroup's last stamp and the first stamp of the workgroup that takes its slot cannot be seen directly; estimate the
# overhead per workgroup as span * 1024 / B - (mean first-to-last)
print(f"per-workgroup slot time span*1024/B = {span * 1024 / B:.1f} us")
f.close()
