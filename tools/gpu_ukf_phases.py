"""Phase timers of the UKF step kernel (debug; SLAM_DEBUG_FLAGS=4): mean 100 MHz ticks per block of the last launch."""
import ctypes as C, os, sys
os.environ["SLAM_DEBUG_FLAGS"] = "4"
os.environ["SLAM_UKF_SPLIT_MIN"] = "100000000"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
names = ["prologue loads", "measurements", "assoc+sigma rows0-3", "weighted mean", "upd: sensing", "upd: leader S", "upd: C,K,x", "P pass+insert", "epilogue"]
tot = sum(out[:9])
print(f"L={L} B={B}: mean us per block-step {tot / B / 100:.1f}")
for i, nm in enumerate(names):
    print(f"   {nm:22s} {out[i] / B / 100:8.1f} us  {100.0 * out[i] / tot:5.1f} %")
