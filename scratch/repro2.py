import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
kv = dict(a.split("=") for a in sys.argv[1:])
L, T, KP, B = 200, 196, int(kv.get("KP", 64)), 1
os.environ["SLAM_PGS_FUSED"] = "-1"; os.environ["SLAM_PGS_LIST"] = kv.get("list", "0"); os.environ["SLAM_PGS_LANES"] = kv.get("lanes", "4")
os.environ["SLAM_PGS_SEG"] = kv.get("seg", "32"); os.environ["SLAM_PGS_SEG_BACK_GLOBAL"] = "0"
import live_ekf_slam_amd as S
from oracle import oracle as O
from live_ekf_slam_amd.scenario import make_scenario
from live_ekf_slam_amd.config import default_config
lm, cmds = make_scenario(1050712621, L, T)
cfg = default_config(); cfg.range_max = 6.413; cfg.fov_min = -3.068; cfg.fov_max = 3.068
r = O.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=602889364, cfg=cfg, nthreads=2)
pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
pg.set_map(lm); pg.set_seed(602889364); pg.init(0.0, 0.0, 0.0)
pg.run_sim(cmds); pg.solvePoseGraph()
g1 = pg.get_graph(0, 1); g0 = pg.get_graph(0, 0); M = r["M"][0]
dp = np.abs(g1["poses"] - r["pose_res"][0]).max(axis=1); dl = np.abs(g1["landmarks"] - r["lm_res"][0, :M]).max(axis=1)
print(kv, "path", pg.last_solve_paths()["segmented"], pg.last_solve_paths()["segment_length"], "M", M, g1["M"], "flags", pg.stats()["flags"], r["flags"])
print("init poses equal", np.array_equal(g0["poses"], r["pose_init"][0]), "pose err max", dp.max(), "at", dp.argmax(), "lm err max", dl.max(), "at", dl.argmax(), "n bad lm", (dl > 1e-7).sum(), "bad idx", np.nonzero(dl > 1e-7)[0][:20])
print("err_final", pg.stats()["err_final"], r["err_final"], "conn count", len(pg.connections(0)))
