"""Phase timers of the pose-graph dense Cholesky kernel (debug).  Run on the GPU box: SLAM_PGS_PROF=1 python tools/gpu_pgs_phases.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLAM_PGS_PROF"] = "1"
os.environ["SLAM_PGS_MAX_TRIALS"] = os.environ.get("SLAM_PGS_MAX_TRIALS", "3")
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
L, B, N = 200, 256, 1000
lm, cmds = make_scenario(1234, L, N - 1)
pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=32).readParams()
pg.set_map(lm); pg.set_seed(2025); pg.init(0, 0, 0); pg.run_sim(cmds)
pg.set_profiling(True)
pg.solvePoseGraph()
out = np.zeros((B, 8), dtype=np.uint64)
fn = _lib.lib().pgs_debug_prof; fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
assert fn(pg.h, out.ctypes.data_as(C.c_void_p)) == 0
names = ["diag load", "diag factor+writeback", "panel solve", "trailing", "backward", "-"]
us = out[:, :6].astype(np.float64) / 100.0   # 100 MHz
print("chol phases, mean over instances (us):", {n: round(float(v), 1) for n, v in zip(names, us.mean(0))}, "total", round(float(us.sum(1).mean()), 1))
print("kernel ms (3 trials):", pg.last_solve_kernel_ms())
