import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
L, B, N = 200, 256, 1000
lm, cmds = make_scenario(1234, L, N - 1)
g = S.BatchedPoseGraph(B, N, L, 8).readParams(); g.set_map(lm); g.set_seed(2025); g.init(0, 0, 0)
g.run_sim(cmds); g.solvePoseGraph()
st = g.stats()
tr = np.asarray(st["trials"]); it = np.asarray(st["iterations"])
print("trials: mean", tr.mean(), "max", tr.max(), "percentiles 50/75/90/95/99:", np.percentile(tr, [50, 75, 90, 95, 99]))
print("active after trial t:", [(t, int((tr > t).sum())) for t in (5, 10, 15, 20, 25, 30, 35, 40, 45, 50)])
print("iterations: mean", it.mean(), "max", it.max())
