import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
kv = dict(a.split("=") for a in sys.argv[1:])
L, T, KP, B = int(kv.get("L", 200)), int(kv.get("T", 219)), int(kv.get("KP", 64)), int(kv.get("B", 7))
os.environ["SLAM_PGS_FUSED"] = kv.get("fused", "3"); os.environ["SLAM_PGS_LIST"] = kv.get("list", "0"); os.environ["SLAM_PGS_LANES"] = kv.get("lanes", "4")
os.environ["SLAM_PGS_SEG"] = kv.get("seg", "32"); os.environ["SLAM_PGS_SEG_BACK_GLOBAL"] = "0"
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
from live_ekf_slam_amd.config import default_config
lm, cmds = make_scenario(847024989, L, T)
cfg = default_config(); cfg.range_max = float(kv.get("range", 6.196)); cfg.fov_min = -2.224; cfg.fov_max = 2.224
pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
pg.set_slots(int(kv.get("slots", 3)))
pg.set_map(lm); pg.set_seed(966224186); pg.init(0.0, 0.0, 0.0)
pg.run_sim(cmds); pg.solvePoseGraph()
print("OK", kv, pg.last_solve_paths()["segmented"], pg.last_solve_paths()["segment_length"], pg.stats()["trials"].tolist(), flush=True)
