#!/usr/bin/env python3
"""Adversarial-message soak of the EKF boundary against the oracle (BIT-EXACT for unflagged instances, equal flags for all): every
instance gets its OWN random external messages - repeated ids inside a message, ids beyond the landmark capacity, more
detections than one wavefront associates at once (> 64), empty messages, tiny and huge ranges - in known-id and unknown-id mode,
fp64 / fp32 storage, random queue depths and getters in between (Filter::update through slam_step; ekf.cpp:65-146).
usage: gpu_soak_adversarial.py [seconds] [seed] [ekf|ukf|both]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
import live_ekf_slam_amd as S

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
which = sys.argv[3] if len(sys.argv) > 3 else "ekf"
t_end = time.time() + budget
runs = fails = 0
cat = {}
while time.time() < t_end:
    ukf = which == "ukf" or (which == "both" and rng.random() < 0.35)
    L = int(rng.choice([3, 8, 20, 50] if ukf else [3, 8, 20, 50, 100, 230]))   # 230: the HBM-streamed EKF class (no per-message limit)
    f32 = (not ukf) and L <= 50 and rng.random() < 0.3
    T = int(rng.integers(3, 50 if L <= 200 else 12))
    B = int(rng.integers(1, 10))
    idknown = int(rng.random() < 0.75)
    kcap = int(rng.choice([2, 6, 20, 70] + ([300] if L > 200 else [])))
    # Round 5: no per-message limit any more - the instances whose message is longer than the size class holds go through the HBM-streamed
    # kernels (EkfStepParams / UkfStepParams::long_mode; fp32 storage included), so the oracle runs exactly as the reference does.
    idmax = int(rng.choice([max(2, L // 2), L, 2 * L, 400]))
    idmin = int(rng.choice([0, 0, -3, -40]))   # any int is an id for the reference (ekf.cpp:99-108), negative ones included (-1 / -2 were sentinels once)
    seed = int(rng.integers(1, 1 << 30))
    if os.environ.get("SOAK_REPLAY"):   # "ekf|ukf L T B f32 idknown kcap idmax seed [idmin]"
        a = os.environ["SOAK_REPLAY"].split()
        ukf = a[0] == "ukf"; a = a[1:]
        L, T, B = int(a[0]), int(a[1]), int(a[2]); f32 = a[3] == "True"; idknown, kcap, idmax, seed = int(a[4]), int(a[5]), int(a[6]), int(a[7]); t_end = 0
        idmin = int(a[8]) if len(a) > 8 else 0
    desc = f"{'ukf' if ukf else 'ekf'} L={L} T={T} B={B} f32={f32} idknown={idknown} kcap={kcap} idmax={idmax} seed={seed} idmin={idmin}"
    if os.environ.get("SOAK_VERBOSE"): print("RUN", desc, flush=True)
    mr = np.random.default_rng(seed)
    cfg = S.default_config(); cfg.landmark_id_is_known = idknown
    f = (S.BatchedUKF(B, L) if ukf else S.BatchedEKF(B, L, dtype=S.F32 if f32 else S.F64)).readParams(cfg); f.init(0.0, 0.0, 0.0)
    if not ukf and mr.random() < 0.5: f.set_lazy_steps(int(mr.choice([1, 3, 32])))
    es = []
    for b in range(B):
        e = O.OracleUKF(cfg, L_max=L) if ukf else O.OracleEKF(cfg, L_max=L, mode=O.MODE_FAST | (O.STORAGE_F32 if f32 else 0))
        e.init(0, 0, 0); es.append(e)
    oflags = np.zeros(B, dtype=np.int64)
    for t in range(T):
        cmd = np.array([mr.uniform(0, 0.1), mr.uniform(-0.05, 0.05)], dtype=np.float32)
        ks = mr.integers(0, kcap + 1, B)
        if mr.random() < 0.1: ks[:] = 0
        K = max(1, int(ks.max()))
        meas = np.zeros((B, K, 3), dtype=np.float32)
        for b in range(B):
            k = int(ks[b])
            ids = mr.integers(idmin, idmax, k)
            if k > 1 and mr.random() < 0.3: ids[mr.integers(0, k)] = ids[mr.integers(0, k)]      # a repeated id
            meas[b, :k, 0] = ids
            meas[b, :k, 1] = mr.choice([mr.uniform(0.05, 5.0, k), mr.uniform(1e-4, 1e-2, k), mr.uniform(50, 500, k)][:1 + int(mr.random() < 0.2) * 2])
            meas[b, :k, 2] = mr.uniform(-3.1, 3.1, k)
        f.update(cmd, meas, ks.astype(np.int32))
        for b in range(B):
            if oflags[b] == 0 or True:
                oflags[b] |= es[b].update(cmd[0], cmd[1], meas[b, :ks[b]])
        if mr.random() < 0.15: f.get_state(int(mr.integers(0, B)))
        if os.environ.get("SOAK_REPLAY"):
            gf = f.status().astype(np.int64)
            if not np.array_equal(gf, oflags):
                b = int(np.flatnonzero(gf != oflags)[0])
                print(f"step {t}: flags {gf.tolist()} vs oracle {oflags.tolist()}; instance {b}: M gpu {f.get_state(b)['M']} oracle {es[b].state()['M']}, message ids {meas[b, :ks[b], 0].astype(int).tolist()}"
                      f" r {np.round(meas[b, :ks[b], 1], 3).tolist()} b {np.round(meas[b, :ks[b], 2], 3).tolist()}")
                break
    gflags = f.status().astype(np.int64)
    why = []
    if not np.array_equal(gflags, oflags):
        d = gflags ^ oflags
        both_frozen = ((gflags & 4) != 0) & ((oflags & 4) != 0)
        kind = "capacity bit of an instance frozen in both" if np.all((d == 0) | ((d == 8) & both_frozen)) else "OTHER"
        cat[kind + (" idknown" if idknown else " unknown-id")] = cat.get(kind + (" idknown" if idknown else " unknown-id"), 0) + 1
        why.append(f"[{kind}] flags {gflags.tolist()} vs oracle {oflags.tolist()}")
    for b in range(B):
        if oflags[b] != 0 or gflags[b] != 0:
            continue
        so, sg = es[b].state(), f.get_state(b)
        if sg["M"] != so["M"] or not (np.array_equal(sg["ids"], so["ids"]) and np.array_equal(sg["x"], so["x"]) and np.array_equal(sg["P"], so["P"])):
            why.append(f"state of instance {b} (M {sg['M']} vs {so['M']})"); break
    f.close()
    runs += 1
    if why:
        fails += 1
        print(f"MISMATCH {desc}: {'; '.join(why)[:500]}", flush=True)
print(f"{runs} adversarial configurations in {budget:.0f} s, {fails} mismatches", cat)
sys.exit(1 if fails else 0)
