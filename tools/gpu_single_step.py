import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
L, B, K = 50, 65536, 40
lm, cmds = make_scenario(1234, L, 400)
Lc = _lib.lib()
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
cm = [np.ascontiguousarray(c, dtype=np.float32) for c in cmds]
for code in sys.argv[1:]:
    os.environ["SLAM_WAVES_PER_FILTER"] = code
    os.environ["SLAM_LAZY_STEPS"] = "0"
    g = S.BatchedEKF(B, L).readParams(); g.set_map(lm); g.init(0, 0, 0)
    g.set_vision(1e9, -4.0, 4.0); g.update_sim(cmds[0]); g.set_vision(3.0, -1.57, 1.57)
    g.run_sim(cmds[1:20]); g.sync()
    t0 = time.perf_counter()
    for i in range(K): _lib.check(Lc.slam_step_sim(g.h, fp(cm[20 + i])))
    g.sync(); dt = time.perf_counter() - t0
    print(f"single-step sim, variant {code}: {dt / K * 1e3:.3f} ms/step {B * K / dt / 1e6:.2f} M steps/s", flush=True)
    g.close()
