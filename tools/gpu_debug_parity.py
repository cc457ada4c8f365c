#!/usr/bin/env python3
"""Debug aid: first timestep / elements where a multi-step run differs from the oracle (usage: f32|f64 [L] [T] [blind])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
from oracle import oracle as O
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = int(sys.argv[3]) if len(sys.argv) > 3 else 110
blind = len(sys.argv) > 4
B = 8
lm, cmds = make_scenario(777, L, T)
vis = np.tile([3.0, -1.57, 1.57], (T, 1)); vis[0] = [1e9, -4.0, 4.0]
if blind: vis[35:75] = [1e-6, -1.57, 1.57]
mode = O.MODE_FAST | (O.STORAGE_F32 if dt == "f32" else 0)
for Tend in list(range(2, T + 1, 3)):
    f = S.BatchedEKF(B, L, dtype=S.F32 if dt == "f32" else S.F64).readParams()
    f.set_map(lm); f.set_seed(5); f.set_instance_offset(9); f.init(0, 0, 0)
    f.set_vision(*vis[0]); f.run_sim(cmds[0:1])
    # one launch for the rest, but the vision changes: split at the vision changes
    t = 1
    for t1 in (35, 75, T):
        t1 = min(t1, Tend)
        if t1 > t:
            f.set_vision(*vis[t]); f.run_sim(cmds[t:t1]); t = t1
    r = O.run_ekf_batch(lm, cmds[:Tend], B, L, seed=5, inst0=9, nthreads=4, mode=mode, vision=vis[:Tend])
    bad = False
    for b in range(B):
        n = 3 + 2 * r["M"][b]
        s = f.get_state(b)
        Pd = s["P"] - r["P"][b, :n * n].reshape(n, n)
        xd = s["x"] - r["x"][b, :n]
        if np.any(Pd != 0) or np.any(xd != 0):
            rr, cc = np.nonzero(Pd)
            print(f"Tend={Tend} inst {b}: {len(rr)} P entries differ (max {np.abs(Pd).max():.3e}), rows {sorted(set(rr.tolist()))[:12]} cols {sorted(set(cc.tolist()))[:12]}; x diff at {np.nonzero(xd)[0].tolist()[:8]}")
            bad = True
            break
    f.close()
    if bad:
        break
else:
    print("no mismatch")
