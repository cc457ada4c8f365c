#!/usr/bin/env python3
"""Per-workgroup begin / end / placement of the last fused chain + SYRK launch of a pose-graph solve (debug).
usage: gpu_pgs_fused_wg.py [trials]   (the solve stops after `trials` trials; the stamps are those of the last one)
SLAM_PGS_NOTRIM=256*(w-1) times the phases of consumer wavefront w (1..7) instead of wavefront 1; =16 prints the SIMD of every wavefront."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLAM_PGS_PROF"] = "1"
os.environ["SLAM_PGS_MAX_TRIALS"] = sys.argv[1] if len(sys.argv) > 1 else "1"
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
L, B, N = 200, 256, 1000
lm, cmds = make_scenario(1234, L, N - 1)
pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=32).readParams()
pg.set_map(lm); pg.set_seed(2025); pg.init(0, 0, 0); pg.run_sim(cmds)
pg.set_profiling(True)
pg.solvePoseGraph()
out = np.zeros((B, 16), dtype=np.uint64)
fn = _lib.lib().pgs_debug_prof2; fn.argtypes = [C.c_void_p, C.c_void_p]; fn.restype = C.c_int
assert fn(pg.h, out.ctypes.data_as(C.c_void_p)) == 0
w = out.reshape(B * 2, 8)
if os.environ.get("SLAM_PGS_NOTRIM") == "16":
    simd = (w.astype(np.int64) >> 4) & 3
    u, cnt = np.unique(simd, axis=0, return_counts=True)
    print("SIMD of wavefronts 0..7 (pattern: count):", {tuple(int(v) for v in a): int(n) for a, n in zip(u, cnt)})
    sys.exit(0)
w = w[w[:, 1] > 0]
w = w[w[:, 0].astype(np.int64) >= w[:, 0].astype(np.int64).max() - 200000]   # the last launch only (stamps of slots that did not run are older)
t0 = w[:, 0].astype(np.int64); t1 = w[:, 1].astype(np.int64)
base = t0.min()
dur = (t1 - t0) / 100.0
print(f"{len(w)} workgroups stamped; duration us: min {dur.min():.0f} median {np.median(dur):.0f} max {dur.max():.0f}; "
      f"launch span {(t1.max() - base) / 100.0:.0f} us; started later than 100 us after the first: {int(((t0 - base) > 10000).sum())}")
late = (t0 - base) > 10000
if late.any():
    print("late starters begin at us:", np.sort((t0[late] - base) / 100.0)[:10], "...")
print("mean us per workgroup: producer before %.0f, after %.0f, recursion %.0f | timed consumer wavefront: columns %.0f, tiles + z %.0f, barrier %.0f" % tuple(w[:, 2:8].astype(np.float64).mean(0) / 100.0))
print("kernel ms:", pg.last_solve_kernel_ms())
