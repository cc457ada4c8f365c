#!/usr/bin/env python3
"""Debug: the external-measurement path (update() with the buffers a twin recorded from the device generator) against the twin."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario
L, T, B, seed, sc, inst0 = 35, 15, 9, 439150776, 763968, 84460
f32 = len(sys.argv) > 1 and sys.argv[1] == "f32"
lazy = int(sys.argv[2]) if len(sys.argv) > 2 else 0
KS = 16
lm, cmds = make_scenario(sc, L, T)
def fresh():
    g = S.BatchedEKF(B, L, dtype=S.F32 if f32 else S.F64).readParams(); g.set_map(lm); g.set_seed(seed); g.set_instance_offset(inst0); g.init(0, 0, 0); return g
twin = fresh(); f = fresh()
twin.last_meas(KS)
if lazy: f.set_lazy_steps(lazy)
for t in range(T):
    twin.update_sim(cmds[t]); m, c = twin.last_meas(KS)
    f.update(cmds[t], m, c)
    if not lazy or t == T - 1:
        bad = [b for b in range(B) if not (np.array_equal(f.get_state(b)["x"], twin.get_state(b)["x"]) and np.array_equal(f.get_state(b)["P"], twin.get_state(b)["P"]))]
        print("step", t, "k", c.tolist(), "M", twin.landmark_counts().tolist(), "differs:", bad)
        if bad:
            b = bad[0]; sa, sb = f.get_state(b), twin.get_state(b)
            print("  instance", b, "M", sa["M"], sb["M"], "ids", sa["ids"].tolist(), sb["ids"].tolist(), "max |dx|", np.abs(sa["x"] - sb["x"]).max() if sa["M"] == sb["M"] else None)
            print("  meas", m[b, :c[b]].tolist())
            break
