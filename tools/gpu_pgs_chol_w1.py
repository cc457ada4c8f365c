"""Timers of wavefront 1 in the pose-graph Cholesky (an EXPERIMENTAL library: docs/dev/sessions/gpu_r5ap.sh): per factorisation the time its lane 0 spends
staging the next panel's block rows (until its loads have landed), forming its tiles over k < j0, and the number of 64-k batches of A operands."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLAM_PGS_PROF"] = "1"
os.environ["SLAM_PGS_MAX_TRIALS"] = os.environ.get("SLAM_PGS_MAX_TRIALS", "3")
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
L, B, N = 200, int(os.environ.get("W1_BATCH", "256")), 1000
lm, cmds = make_scenario(1234, L, N - 1)
pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=32).readParams()
pg.set_map(lm); pg.set_seed(2025); pg.init(0, 0, 0); pg.run_sim(cmds)
pg.set_profiling(True)
pg.solvePoseGraph()
out = np.zeros((B, 8), dtype=np.uint64)
fn = _lib.lib().pgs_debug_prof; fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
assert fn(pg.h, out.ctypes.data_as(C.c_void_p)) == 0
o = out.astype(np.float64)
print(f"batch {B}: phases (us) diag half-phases {o[:,1].mean()/100:.1f}, panel solve {o[:,2].mean()/100:.1f}, completion {o[:,3].mean()/100:.1f}, backward {o[:,4].mean()/100:.1f}")
print(f"wavefront 1: staging until the loads have landed {o[:,5].mean()/100:.1f} us, tile formation {o[:,6].mean()/100:.1f} us over {o[:,7].mean():.1f} batches of 64 k "
      f"= {o[:,6].mean()/100/max(o[:,7].mean(),1):.2f} us per batch; panels {int(np.ceil(343/16))}")
