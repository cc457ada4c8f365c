import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
from oracle import oracle as O
L, T, B = 20, 40, 4
lm, cmds = make_scenario(1234, L, T)
f = S.BatchedEKF(B, L, dtype=S.F32).readParams(); f.set_map(lm); f.set_seed(21); f.init(0, 0, 0)
for t in range(T):
    f.update_sim(cmds[t])
    r = O.run_ekf_batch(lm, cmds[:t + 1], B, L, seed=21, mode=O.MODE_FAST | O.STORAGE_F32)
    sg = f.get_state(0); n = 3 + 2 * r["M"][0]
    dx = np.abs(sg["x"] - r["x"][0, :n]).max() if sg["M"] == r["M"][0] else -1
    dP = np.abs(sg["P"].ravel() - r["P"][0, :n * n]).max() if sg["M"] == r["M"][0] else -1
    if dx != 0 or dP != 0:
        print("first mismatch at step", t, "M", sg["M"], r["M"][0], "dx", dx, "dP", dP)
        print(" x gpu", sg["x"][:5], "\n x orc", r["x"][0, :5])
        d = np.abs(sg["P"].ravel() - r["P"][0, :n * n]).reshape(n, n); print(" P diff nonzero at", np.argwhere(d > 0)[:10].tolist())
        break
else:
    print("no mismatch in", T, "steps")
