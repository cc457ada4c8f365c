#!/usr/bin/env python3
"""Left-looking vs right-looking dense Cholesky of the pose-graph solve (SLAM_PGS_CHOL_LL=1 / 0): bitwise comparison of the solve results
on the bench workload (1000 poses x 200 landmarks) and solves/s of both.  usage: gpu_pgs_chol_ab.py [batch]"""
import os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import time, numpy as np
    import live_ekf_slam_amd as S
    from live_ekf_slam_amd.scenario import make_scenario
    B = int(sys.argv[1]); L, N = 200, 1000
    lm, cmds = make_scenario(1234, L, N - 1)
    pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=32).readParams()
    pg.set_map(lm); pg.set_seed(2025); pg.init(0.0, 0.0, 0.0); pg.run_sim(cmds)
    pg.solvePoseGraph()
    t0 = time.perf_counter()
    for _ in range(3): pg.solvePoseGraph()
    dt = (time.perf_counter() - t0) / 3
    st = pg.stats()
    poses = np.stack([pg.get_graph(b, 1)["poses"] for b in range(0, B, max(1, B // 16))])
    np.savez(sys.argv[3], poses=poses, it=st["iterations"], tr=st["trials"], fl=st["flags"])
    pg.set_profiling(True); pg.solvePoseGraph(); kms = pg.last_solve_kernel_ms(); pg.set_profiling(False)
    print(json.dumps({"solves_per_s": B / dt, "ms": dt * 1e3, "kernel_ms": {k: round(v, 2) for k, v in kms.items()}}))
    sys.exit(0)
import numpy as np
B = sys.argv[1] if len(sys.argv) > 1 else "256"
res = {}
for ll in ("0", "1", "0", "1"):
    env = dict(os.environ, SLAM_PGS_CHOL_LL=ll)
    out = subprocess.run([sys.executable, __file__, B, "child", f"/tmp/pgs_ll{ll}.npz"], capture_output=True, text=True, env=env)
    print(f"LL={ll}:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-800:], flush=True)
a, b = np.load("/tmp/pgs_ll0.npz"), np.load("/tmp/pgs_ll1.npz")
print("bitwise equal poses:", np.array_equal(a["poses"], b["poses"]), " max |diff|:", float(np.abs(a["poses"] - b["poses"]).max()),
      " equal LM iterations / trials / flags:", np.array_equal(a["it"], b["it"]), np.array_equal(a["tr"], b["tr"]), np.array_equal(a["fl"], b["fl"]))
