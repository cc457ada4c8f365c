#!/usr/bin/env python3
"""Per-detection-count step times inside ONE multi-step launch (SLAM_DEBUG_FLAGS=32: every workgroup stamps the 100 MHz
wall clock and its instance's detection count at the end of each timestep).  Prints, per k, the mean time a workgroup
spends on a step with k detections and the batch-level ms/step that corresponds to (mean x batch / resident workgroups).
usage: gpu_steptimes.py [f64|f32] [t0] [steps<=128] [variant]"""
import ctypes as C, os, sys
os.environ["SLAM_DEBUG_FLAGS"] = "32"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario


def per_k_table(f, B, steps, resident):
    lib = _lib.lib()
    lib.slam_debug_read_prof_raw.argtypes = [C.c_void_p, C.c_void_p]
    buf = np.zeros((B, 128), dtype=np.uint64)
    _lib.check(lib.slam_debug_read_prof_raw(f.h, buf.ctypes.data_as(C.c_void_p)))
    st = (buf[:, :steps] >> np.uint64(4)).astype(np.int64)
    kk = (buf[:, :steps] & np.uint64(15)).astype(np.int64)
    d = np.diff(st, axis=1) / 100.0          # microseconds per workgroup-step (steps 1..)
    k1 = kk[:, 1:]
    rows = []
    for k in range(int(k1.max()) + 1):
        sel = k1 == k
        if sel.sum() == 0:
            continue
        us = float(d[sel].mean())
        rows.append(dict(k=k, share=float(sel.mean()), us_per_workgroup_step=round(us, 2), ms_per_batch_step=round(us * B / resident * 1e-3, 4)))
    span_ms = (st[:, steps - 1].max() - st[:, 0].min()) / 1e5
    return rows, span_ms


if __name__ == "__main__":
    dt = sys.argv[1] if len(sys.argv) > 1 else "f64"
    t0 = int(sys.argv[2]) if len(sys.argv) > 2 else 644
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    if len(sys.argv) > 4:
        os.environ["SLAM_WAVES_PER_FILTER"] = sys.argv[4]
    L, B = 50, 65536
    lm, cmds = make_scenario(1234, L, t0 + steps + 1)
    f = S.BatchedEKF(B, L, dtype=S.F32 if dt == "f32" else S.F64).readParams(); f.set_map(lm); f.set_seed(2025); f.init(0, 0, 0)
    f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)
    f.run_sim(cmds[1:t0]); f.sync()
    f.run_sim(cmds[t0:t0 + steps]); f.sync()
    rows, span = per_k_table(f, B, steps, 1024)
    print(f"{dt} t0={t0} steps={steps}: launch span {span:.2f} ms = {span / (steps - 1):.3f} ms/step")
    for r in rows:
        print("  k=%d  share %.3f  %.2f us per workgroup-step  -> %.3f ms per batch step (1024 resident workgroups)" % (r["k"], r["share"], r["us_per_workgroup_step"], r["ms_per_batch_step"]))
