#!/usr/bin/env python3
"""Debug: where does a two-launch EKF run leave the oracle?  Runs [0, split) in launches of `chunk` steps and compares with the
oracle run over the same prefix, then the rest.  usage: gpu_debug_split.py L T B seed scenario inst0 f32 idknown wide chunk variant split"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
a = sys.argv[1:]
L, T, B, seed, sc, inst0 = (int(v) for v in a[:6]); f32 = a[6] == "True"; idknown = int(a[7]); wide = a[8] == "True"; chunk, var, split = int(a[9]), int(a[10]), int(a[11])
os.environ["SLAM_RUN_CHUNK"] = str(chunk)
if var: os.environ["SLAM_WAVES_PER_FILTER"] = str(var)
lm, cmds = make_scenario(sc, L, T)
cfg = S.default_config(); cfg.landmark_id_is_known = idknown
if wide: cfg.range_max = 1e9; cfg.fov_min = -4.0; cfg.fov_max = 4.0
mode = O.MODE_FAST | (O.STORAGE_F32 if f32 else 0)
def cmp(f, upto, tag):
    r = O.run_ekf_batch(lm, cmds[:upto], B, L, seed=seed, inst0=inst0, nthreads=8, cfg=cfg, mode=mode)
    bad = []
    for b in range(B):
        n = 3 + 2 * r["M"][b]; sg = f.get_state(b)
        if sg["M"] != r["M"][b] or not np.array_equal(sg["x"], r["x"][b, :n]) or not np.array_equal(sg["P"], r["P"][b, :n * n].reshape(n, n)):
            dP = np.abs(sg["P"] - r["P"][b, :n * n].reshape(n, n)) if sg["M"] == r["M"][b] else None
            bad.append((b, int(sg["M"]), int(r["M"][b]), None if dP is None else (float(dP.max()), np.argwhere(dP > 0)[:6].tolist())))
    print(tag, "after", upto, "steps:", "OK" if not bad else bad[:3], "err stats equal:", np.array_equal(f.error_stats(), r["avg_err"]))
f = S.BatchedEKF(B, L, dtype=S.F32 if f32 else S.F64).readParams(cfg)
f.set_map(lm); f.set_seed(seed); f.set_instance_offset(inst0); f.init(0, 0, 0)
f.run_sim(cmds[:split]); cmp(f, split, "first call")
f.run_sim(cmds[split:]); cmp(f, T, "second call")
f.close()
