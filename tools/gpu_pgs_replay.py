#!/usr/bin/env python3
"""Replay one configuration of tools/gpu_soak_pgs.py under every launch shape: max |pose / landmark difference| to the oracle per shape.
usage: gpu_pgs_replay.py L T KP B seed scenario lanes groups"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
import live_ekf_slam_amd as S
from live_ekf_slam_amd.config import default_config
from live_ekf_slam_amd.scenario import make_scenario
L, T, KP, B, seed, sc, lanes, groups = (int(v) for v in sys.argv[1:9])
lm, cmds = make_scenario(sc, L, T)
cfg = default_config()
r = O.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=seed, cfg=cfg, nthreads=8)
rd = O.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=seed, cfg=cfg, nthreads=8, lin_mode=O.LIN_DENSE)
ed = max(float(np.abs(rd["pose_res"] - r["pose_res"]).max()), float(np.abs(rd["lm_res"] - r["lm_res"]).max()))
print(f"oracle Schur vs oracle dense elimination: max diff {ed:.3e}, same LM counts: {np.array_equal(rd['trials'], r['trials'])}")
os.environ["SLAM_PGS_LANES"] = str(lanes)
for fused in ("0", "2", "3", "4"):
    os.environ["SLAM_PGS_FUSED"] = fused
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
    if groups: pg.set_groups(groups)
    pg.set_map(lm); pg.set_seed(seed); pg.init(0.0, 0.0, 0.0); pg.run_sim(cmds); pg.solvePoseGraph()
    st = pg.stats()
    errs = []
    for b in range(B):
        g1 = pg.get_graph(b, 1); M = r["M"][b]
        errs.append(max(float(np.abs(g1["poses"] - r["pose_res"][b]).max()), float(np.abs(g1["landmarks"] - r["lm_res"][b, :M]).max()) if M else 0.0))
    b = int(np.argmax(errs))
    print(f"fused={fused}: max err {max(errs):.3e} (instance {b}: {int(st['iterations'][b])} iterations, {int(st['trials'][b])} trials, final lambda {st['lam'][b]:.1e}, error {st['err_final'][b]:.6e} vs oracle {r['err_final'][b]:.6e});"
          f" counts equal: {np.array_equal(st['trials'], r['trials'])}")
    pg.close()
