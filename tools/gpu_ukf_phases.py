"""Phase timers of the UKF step kernel (debug; SLAM_DEBUG_FLAGS=4 -> the PROF instantiation): mean 100 MHz ticks per
instance of the last launch.  After the 60-step run it single-steps on and prints every step, because the update
phases only run in steps that have detections (all instances share the trajectory, so a step has them or not)."""
import ctypes as C, os, sys
os.environ["SLAM_DEBUG_FLAGS"] = "4"
os.environ["SLAM_UKF_SPLIT_MIN"] = "100000000"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
L, B = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 4096
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lm, cmds = make_scenario(1234, L, 80 + extra)
f = S.BatchedUKF(B, L).readParams(); f.set_map(lm); f.init(0, 0, 0)
f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
f.run_sim(cmds[1:60]); f.sync()
names = ["prologue loads", "measurements", "assoc+sigma rows0-3", "weighted mean", "upd: sensing", "upd: leader S", "upd: C,K,x", "P pass+insert", "epilogue"]
def show(tag):
    out = (C.c_ulonglong * 16)()
    _lib.lib().slam_debug_read_prof(f.h, out)
    tot = sum(out[:10])
    print(f"{tag} L={L} B={B}: mean us per block-step {tot / B / 100:.1f}")
    for i, nm in enumerate(names):
        print(f"   {nm:22s} {out[i] / B / 100:8.1f} us  {100.0 * out[i] / tot:5.1f} %")
    print(f"   {'(of the P pass: MFMA contraction)':22s} {out[9] / B / 100:8.1f} us")
    return out
show("step 59")
for t in range(60, 60 + extra):
    f.run_sim(cmds[t:t + 1]); f.sync()
    out = (C.c_ulonglong * 16)(); _lib.lib().slam_debug_read_prof(f.h, out)
    meas, cnt = f.last_meas(8) if hasattr(f, "last_meas") else (None, np.zeros(1))
    upd = (out[4] + out[5] + out[6]) / B / 100
    print(f"step {t}: total {sum(out[:10]) / B / 100:6.1f} us, update phases {upd:6.1f} us (sensing {out[4] / B / 100:.1f}, leader S {out[5] / B / 100:.1f}, C,K,x {out[6] / B / 100:.1f}), P pass {(out[7] + out[9]) / B / 100:.1f} (contraction {out[9] / B / 100:.1f}), mean detections {float(np.mean(cnt)):.2f}")
f.close()
