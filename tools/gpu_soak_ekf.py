#!/usr/bin/env python3
"""Soak test of the EKF / UKF step kernels against the CPU oracle (BIT-EXACT): random map sizes, step counts, batch sizes,
seeds, storage type, association mode, sensor range (few / many detections per message), launch chunking and, for the EKF, the
step-kernel variants the library holds.  usage: gpu_soak_ekf.py [seconds] [seed] [ekf|ukf|both]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
which = sys.argv[3] if len(sys.argv) > 3 else "both"
t_end = time.time() + budget
runs = fails = 0
VARIANTS = [0, 0, 1254, 1244, 1464, 1454, 1444, 1442]
REPLAY = os.environ.get("SOAK_REPLAY")   # "ekf L T B seed scenario inst0 f32 idknown wide chunk variant split"
while time.time() < t_end:
    ukf = which == "ukf" or (which == "both" and rng.random() < 0.3)
    L = int(rng.choice([2, 5, 12, 20, 35, 50] if ukf else [2, 5, 12, 20, 35, 50, 80, 100, 150, 200]))
    T = int(rng.integers(3, 120 if L > 50 else 400))
    B = int(rng.integers(1, 40 if L <= 50 else 8))
    seed, sc, inst0 = int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1 << 20))
    f32 = (not ukf) and L <= 50 and rng.random() < 0.3
    idknown = int(rng.random() < 0.7)
    wide = rng.random() < 0.25
    chunk = int(rng.choice([1, 3, 8, 32, 1000]))
    var = 0 if ukf else int(rng.choice(VARIANTS))
    if var and not _lib.lib().slam_variant_available(L, 1 if f32 else 0, var):
        var = 0
    split = -1
    if REPLAY:
        a = REPLAY.split()
        ukf = a[0] == "ukf"; L, T, B, seed, sc, inst0 = (int(v) for v in a[1:7]); f32 = a[7] == "True"; idknown = int(a[8]); wide = a[9] == "True"
        chunk, var, split = int(a[10]), int(a[11]), int(a[12]); t_end = 0
    os.environ["SLAM_RUN_CHUNK"] = str(chunk)
    # UKF: the thread-count variants of its two kernels (SLAM_UKF_TPB = sqrt threads * 10000 + step threads), drawn from a second RNG
    tr = np.random.default_rng(seed ^ 0x2545f491)
    if ukf and os.environ.get("SOAK_PLAIN") is None and tr.random() < 0.4:
        sq, stp = ((256, 128, 64), (128, 64, 256, 192)) if L <= 20 else ((1024, 512, 256), (1024, 256, 512))
        os.environ["SLAM_UKF_TPB"] = str(int(tr.choice(sq)) * 10000 + int(tr.choice(stp)))
    else:
        os.environ.pop("SLAM_UKF_TPB", None)
    if var: os.environ["SLAM_WAVES_PER_FILTER"] = str(var)
    else: os.environ.pop("SLAM_WAVES_PER_FILTER", None)
    desc = (f"{'ukf' if ukf else 'ekf'} L={L} T={T} B={B} seed={seed} scenario={sc} inst0={inst0} f32={f32} idknown={idknown} wide={wide} chunk={chunk} "
            f"variant={var} ukf_tpb={os.environ.get('SLAM_UKF_TPB', '-')}")
    if os.environ.get("SOAK_VERBOSE"): print("RUN", desc, flush=True)
    lm, cmds = make_scenario(sc, L, T)
    cfg = S.default_config(); cfg.landmark_id_is_known = idknown
    if wide:
        cfg.range_max = 1e9; cfg.fov_min = -4.0; cfg.fov_max = 4.0
    if split < 0: split = int(rng.integers(0, T + 1))
    # extras (a second RNG so that SOAK_REPLAY lines of the plain mode stay valid): config switches, a sensor schedule with blind /
    # wide / normal stretches, now and then a batch of more workgroups than the device holds at once, UKF localisation
    xr = np.random.default_rng(seed ^ 0x5bd1e995)
    extras = os.environ.get("SOAK_PLAIN") is None and xr.random() < 0.5
    vis = None; loc = False
    if extras:
        cfg.replicate_vw_quirk = int(xr.random() < 0.7); cfg.ukf_float_trig = int(xr.random() < 0.7)
        if xr.random() < 0.4: cfg.w_r, cfg.w_b, cfg.v_d, cfg.v_th = (float(v) for v in xr.normal(0, [0.01, 0.003, 0.002, 0.002]))
        if xr.random() < 0.3: cfg.min_landmark_separation = float(xr.choice([0.05, 0.2, 0.5]))
        if xr.random() < 0.6:
            vis = np.tile([cfg.range_max, cfg.fov_min, cfg.fov_max], (T, 1)); t0 = 0
            while t0 < T:
                n = int(xr.integers(1, 40)); kind = xr.choice(["normal", "blind", "wide", "narrow"])
                vis[t0:t0 + n] = {"normal": [3.0, -1.57, 1.57], "blind": [1e-6, -1.57, 1.57], "wide": [1e9, -4.0, 4.0], "narrow": [1.5, -0.5, 0.5]}[kind]
                t0 += n
        if L <= 20 and xr.random() < 0.15: B = int(xr.integers(300, 3000))
        loc = ukf and xr.random() < 0.25
    if ukf:
        f = (S.BatchedUKFLoc(B) if loc else S.BatchedUKF(B, L)).readParams(cfg)
    else:
        f = S.BatchedEKF(B, L, dtype=S.F32 if f32 else S.F64).readParams(cfg)
    f.set_map(lm); f.set_seed(seed); f.set_instance_offset(inst0); f.init(0, 0, 0)
    if vis is None:
        f.run_sim(cmds[:split]); f.run_sim(cmds[split:])
    else:   # the sensor limits are per launch: one run_sim per stretch of equal limits (and the split)
        t0 = 0
        while t0 < T:
            t1 = t0 + 1
            while t1 < T and t1 != split and np.array_equal(vis[t1], vis[t0]): t1 += 1
            f.set_vision(*vis[t0]); f.run_sim(cmds[t0:t1]); t0 = t1
    if ukf:
        r = O.run_ukf_batch(lm, cmds, B, 1 if loc else L, seed=seed, inst0=inst0, nthreads=8, cfg=cfg, vision=vis, loc=loc)
    else:
        r = O.run_ekf_batch(lm, cmds, B, L, seed=seed, inst0=inst0, nthreads=8, cfg=cfg, mode=O.MODE_FAST | (O.STORAGE_F32 if f32 else 0), vision=vis)
    why = []
    if not np.array_equal(f.status(), r["flags"]): why.append(f"flags {f.status().tolist()} vs {r['flags'].tolist()}")
    clean = r["flags"] == 0        # a flagged instance (capacity, singular S, ...) is only required to carry the same flag
    if not np.array_equal(f.landmark_counts()[clean], r["M"][clean]): why.append("landmark counts")
    if not np.array_equal(f.truth(), r["truth"]): why.append("true poses")
    if clean.all() and not np.array_equal(f.error_stats(), r["avg_err"]): why.append("error statistics")
    ok = not why
    worst = 0.0
    for b in range(B):
        if not clean[b]:
            continue
        n = (4 if ukf else 3) + 2 * (0 if loc else r["M"][b])
        sg = f.get_state(b)
        if sg["M"] != r["M"][b]:
            ok = False; continue
        dx = np.abs(sg["x"] - r["x"][b, :n]).max(); dP = np.abs(sg["P"] - r["P"][b, :n * n].reshape(n, n)).max()
        if not (np.array_equal(sg["x"], r["x"][b, :n]) and np.array_equal(sg["P"], r["P"][b, :n * n].reshape(n, n)) and np.array_equal(sg["ids"], r["ids"][b, :r["M"][b]])):
            ok = False; worst = max(worst, float(dx), float(dP))
    f.close()
    runs += 1
    if not ok:
        fails += 1
        print(f"MISMATCH {desc} split={split} extras={extras} loc={loc} B={B} vis={'yes' if vis is not None else 'no'}: {'; '.join(why)} max |diff| {worst:.3e}, oracle flags {r['flags'].tolist()}", flush=True)
print(f"{runs} random filter configurations in {budget:.0f} s, {fails} mismatches")
sys.exit(1 if fails else 0)
