#!/usr/bin/env python3
"""Phase timers of the UKF sqrt kernel (nearestSPD + warm-started parallel Jacobi), SLAM_DEBUG_FLAGS=4: mean 100 MHz ticks
per instance of the LAST step, slots 10..15 of the debug buffer (0..9 belong to the step kernel)."""
import ctypes as C, os, sys
os.environ["SLAM_DEBUG_FLAGS"] = "4"
os.environ["SLAM_UKF_SPLIT_MIN"] = "100000000"     # one stream: the two kernels of a step back to back
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
L, B = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 4096
lm, cmds = make_scenario(1234, L, 80)
f = S.BatchedUKF(B, L).readParams(); f.set_map(lm); f.init(0, 0, 0)
f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
f.run_sim(cmds[1:60]); f.sync()
out = (C.c_ulonglong * 16)()
_lib.lib().slam_debug_read_prof(f.h, out)
names = ["load + warm transform", "convergence checks", "rotation parameters", "rotation phase", "epilogue (sqtP)"]
tot = sum(out[10:15]); rounds = out[15] / B
print(f"L={L} B={B}: sqrt kernel {tot / B / 100:.1f} us per instance (wavefront 0), {rounds:.1f} rounds per step")
for i, nm in enumerate(names):
    print(f"   {nm:24s} {out[10 + i] / B / 100:8.1f} us  {100.0 * out[10 + i] / tot:5.1f} %")
print(f"   per round: parameters {out[12] / B / 100 / rounds * 1e3:.0f} ns, rotation {out[13] / B / 100 / rounds * 1e3:.0f} ns")
f.close()
