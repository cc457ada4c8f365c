#!/usr/bin/env python3
"""Soak test of the pose-graph solver against the CPU oracle: random map sizes, pose counts, factor slots, batch sizes, seeds
and launch shapes (slot list on / off, chain + SYRK fused with 2 / 3 / 4 workgroups per instance, two launches, solve
groups) for a given number of seconds.  Same criteria as tests/test_parity_pgs_gpu.py (identical LM iteration / trial
counts and flags, 1e-7 m on poses and landmarks - or, for the rare ill-conditioned instance, 10 x the distance between the oracle's own
Schur and dense eliminations).  usage: gpu_soak_pgs.py [seconds] [seed] [big]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
import live_ekf_slam_amd as S
from live_ekf_slam_amd.config import default_config
from live_ekf_slam_amd.scenario import make_scenario

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
big = len(sys.argv) > 3 and sys.argv[3] == "big"
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"   # round 6: wide sensors on dense maps - segments of 32 poses see more than 63 landmarks
t_end = time.time() + budget
runs = fails = soft = 0
paths = {}
while time.time() < t_end:
    L = int(rng.choice([3, 8, 20, 40, 60, 100, 150, 200]))
    T = int(rng.integers(5, 400)) if L > 100 else int(rng.integers(5, 1000))
    KP = int(rng.choice([4, 8, 16, 32]))
    B = int(rng.integers(1, 24))
    if big:   # BASELINE-size graphs in batches that reach every per-trial launch shape (<= 64 / 85 / 128 / more running slots)
        L = int(rng.choice([120, 170, 200])); T = int(rng.integers(300, 1000)); KP = int(rng.choice([16, 32])); B = int(rng.choice([6, 40, 75, 110, 150]))
    seed, sc = int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30))
    fused = str(rng.choice(["-1", "0", "2", "3", "4"])); lst = str(rng.choice(["1", "1", "0"])); groups = int(rng.choice([0, 0, 2, 3]))
    lanes = str(rng.choice(["4", "4", "1", "2"]))
    seg = str(rng.choice(["32", "32", "16", "8", "5", "0"]))   # round 5: poses per segment of the segmented elimination (0: the sequential chain)
    os.environ["SLAM_PGS_FUSED"] = fused; os.environ["SLAM_PGS_LIST"] = lst; os.environ["SLAM_PGS_LANES"] = lanes; os.environ["SLAM_PGS_SEG"] = seg
    os.environ["SLAM_PGS_SEG_BACK_GLOBAL"] = str(rng.choice(["0", "0", "1"]))
    slots = int(rng.choice([0, 0, 0, 3, 16]))   # round 6: streaming (graphs wait for running slots)
    lm, cmds = make_scenario(sc, L, T)
    cfg = default_config()
    if wide:
        L = int(rng.choice([120, 200])); T = int(rng.integers(40, 400)); KP = 64; B = int(rng.integers(1, 8)); seg = "32"
        os.environ["SLAM_PGS_SEG"] = seg
        lm, cmds = make_scenario(sc, L, T)
        cfg.range_max = float(rng.uniform(4.5, 6.6)); fv = float(rng.uniform(1.6, 3.1)); cfg.fov_min = -fv; cfg.fov_max = fv
    if os.environ.get("SOAK_TRACE"):
        print(f"RUN L={L} T={T} KP={KP} B={B} seed={seed} scenario={sc} fused={fused} list={lst} groups={groups} lanes={lanes} seg={seg} slots={slots} "
              f"back_global={os.environ['SLAM_PGS_SEG_BACK_GLOBAL']} range_max={cfg.range_max:.3f} fov={cfg.fov_max:.3f}", flush=True)
    r = O.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=seed, cfg=cfg, nthreads=8)
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
    if groups: pg.set_groups(groups)
    pg.set_slots(slots)
    pg.set_map(lm); pg.set_seed(seed); pg.init(0.0, 0.0, 0.0)
    pg.run_sim(cmds); pg.solvePoseGraph()
    lp = pg.last_solve_paths()
    key = f"segments of {lp['segment_length']}" if lp["segmented"] else "sequential chain"
    paths[key] = paths.get(key, 0) + 1
    st = pg.stats()
    ok = (np.array_equal(st["flags"], r["flags"]) and np.array_equal(st["iterations"], r["iterations"]) and np.array_equal(st["trials"], r["trials"]))
    err = 0.0
    for b in range(B):
        g1 = pg.get_graph(b, 1); M = r["M"][b]
        ok = ok and g1["M"] == M
        if g1["M"] == M:
            err = max(err, float(np.abs(g1["poses"] - r["pose_res"][b]).max()), float(np.abs(g1["landmarks"] - r["lm_res"][b, :M]).max()) if M else 0.0)
    note = ""
    if ok and not err < 1e-7:
        # A long LM path (15+ iterations, lambda walking) ends in a poorly conditioned system: the oracle's own two eliminations
        # (Schur complement, poses first / dense Cholesky of the whole system) then differ by more than 1e-7 m themselves.  The
        # yardstick for such an instance is that difference, not the fixed tolerance.
        # (round 5: the oracle's segmented orders join the yardstick - for the instance round 4's soak reported, Schur vs dense is 2.5e-9 m
        # and Schur vs 8-pose segments 7.2e-8 m: tests/test_parity_pgs_gpu.py::test_the_ill_conditioned_instance_of_the_round_4_soak)
        ed = 0.0
        for lmode in ([O.LIN_DENSE] if T * 3 + 2 * L <= 1400 else []) + [O.LIN_SEG | (32 << 8), O.LIN_SEG | (16 << 8), O.LIN_SEG | (8 << 8)]:
            rd = O.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=seed, cfg=cfg, nthreads=8, lin_mode=lmode)
            ed = max(ed, float(np.abs(rd["pose_res"] - r["pose_res"]).max()), float(np.abs(rd["lm_res"] - r["lm_res"]).max()))
        if err < 10.0 * ed:
            soft += 1; note = f" (ill-conditioned: the oracle's two eliminations differ by {ed:.1e} m, the GPU by {err:.1e} m)"
            if os.environ.get("SOAK_VERBOSE"): print("NOTE", note, flush=True)
        else:
            ok = False
    pg.close()
    runs += 1
    if not ok:
        fails += 1
        print(f"MISMATCH L={L} T={T} KP={KP} B={B} seed={seed} scenario={sc} fused={fused} list={lst} groups={groups} lanes={lanes} seg={seg} slots={slots} range_max={cfg.range_max:.3f} fov={cfg.fov_max:.3f}: max err {err:.3e}, "
              f"iterations {st['iterations'].tolist()} vs {r['iterations'].tolist()}, trials {st['trials'].tolist()} vs {r['trials'].tolist()}, flags {st['flags'].tolist()} vs {r['flags'].tolist()}", flush=True)
print("elimination orders the solves ran:", dict(sorted(paths.items())))
print(f"{runs} random pose-graph configurations in {budget:.0f} s, {fails} mismatches" + (f"; {soft} ill-conditioned instances beyond 1e-7 m but within 10 x the distance of the oracle's own two eliminations" if soft else ""))
sys.exit(1 if fails else 0)
