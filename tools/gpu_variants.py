#!/usr/bin/env python3
"""Time tuning variants of the EKF step kernel on the headline workload (L=50, batch 65536) over three windows of the bench
trajectory: the detection-heavy one the round-1 driver run hit (t = 46..65, mean k 2.42), the bench window (t = 644..663,
mean k 1.71, 17 % of the steps with k = 3) and 100 steps from t = 644 (mean k 1.68).  Needs a SLAM_SWEEP=1 build.
usage: gpu_variants.py [f64|f32] code [code ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario

dtype = S.F32 if sys.argv[1] == "f32" else S.F64
codes = [int(c) for c in sys.argv[2:]]
L, B = 50, 65536
lm, cmds = make_scenario(1234, L, 800)
esz = 4 if dtype == S.F32 else 8
for code in codes:
    os.environ["SLAM_WAVES_PER_FILTER"] = str(code)
    try:
        f = S.BatchedEKF(B, L, dtype=dtype).readParams(); f.set_map(lm); f.set_seed(2025); f.init(0, 0, 0)
        f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
        f.run_sim(cmds[1:41]); f.run_sim(cmds[41:46]); f.sync()
        out = []
        def timed(a, b):
            f.k_histogram(reset=True)
            t0 = time.perf_counter(); f.run_sim(cmds[a:b]); f.sync(); dt = time.perf_counter() - t0
            h = f.k_histogram().astype(float); kbar = (h * np.arange(8)).sum() / h.sum()
            return dt / (b - a) * 1e3, kbar
        out.append(timed(46, 66))
        f.run_sim(cmds[66:639]); f.run_sim(cmds[639:644]); f.sync()
        out.append(timed(644, 664))
        out.append(timed(664, 744))
        ab = 2 * (103 * 103 + 103) * esz * B
        s = "  ".join(f"{ms:.3f} ms/step (k {kb:.2f}) frac {ab / (ms * 1e-3) / 8e12:.3f}" for ms, kb in out)
        print(f"variant {code:5d} {sys.argv[1]}: t46+20: {out[0][0]:.3f} ms (k {out[0][1]:.2f}, frac {ab / (out[0][0] * 1e-3) / 8e12:.3f}) | "
              f"t644+20: {out[1][0]:.3f} ms (k {out[1][1]:.2f}, frac {ab / (out[1][0] * 1e-3) / 8e12:.3f}) | "
              f"t664+80: {out[2][0]:.3f} ms (k {out[2][1]:.2f}, frac {ab / (out[2][0] * 1e-3) / 8e12:.3f})  flags {np.unique(f.status())}", flush=True)
        f.close()
    except Exception as e:
        print(f"variant {code}: {e}", flush=True)
