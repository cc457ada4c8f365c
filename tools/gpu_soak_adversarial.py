#!/usr/bin/env python3
"""Adversarial-message soak of the EKF boundary against the oracle (BIT-EXACT for unflagged instances, equal flags for all): every
instance gets its OWN random external messages - repeated ids inside a message, ids beyond the landmark capacity, more
detections than one wavefront associates at once (> 64), empty messages, tiny and huge ranges - in known-id and unknown-id mode,
fp64 / fp32 storage, random queue depths and getters in between (Filter::update through slam_step; ekf.cpp:65-146).
Round 5: also UKF_LOC against a random map (ids inside and outside of it, messages longer than the step kernel's size class holds) and the
device-buffer entry (slam_step_dev with the caller's stride as the bound: the launch pair of LDS kernel + streamed kernel) for every kind.
usage: gpu_soak_adversarial.py [seconds] [seed] [ekf|ukf|both]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
import live_ekf_slam_amd as S
import ctypes as C
_hip = C.CDLL("libamdhip64.so")
_hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
_hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
_hip.hipFree.argtypes = [C.c_void_p]

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
which = sys.argv[3] if len(sys.argv) > 3 else "ekf"
t_end = time.time() + budget
runs = fails = 0
cat = {}
while time.time() < t_end:
    ukf = which == "ukf" or (which == "both" and rng.random() < 0.35)
    L = int(rng.choice([3, 8, 20, 50] if ukf else [3, 8, 20, 50, 100, 230]))   # 230: the HBM-streamed EKF class (no per-message limit)
    f32 = (not ukf) and rng.random() < 0.3     # (fp32 storage beyond 50 landmarks: the streamed kernel, round 5)
    T = int(rng.integers(3, 50 if L <= 200 else 12))
    B = int(rng.integers(1, 10))
    idknown = int(rng.random() < 0.75)
    kcap = int(rng.choice([2, 6, 20, 70] + ([300] if L > 200 else [])))
    # Round 5: no per-message limit any more - the instances whose message is longer than the size class holds go through the HBM-streamed
    # kernels (EkfStepParams / UkfStepParams::long_mode; fp32 storage included), so the oracle runs exactly as the reference does.
    idmax = int(rng.choice([max(2, L // 2), L, 2 * L, 400]))
    idmin = int(rng.choice([0, 0, -3, -40]))   # any int is an id for the reference (ekf.cpp:99-108), negative ones included (-1 / -2 were sentinels once)
    seed = int(rng.integers(1, 1 << 30))
    loc = ukf and rng.random() < 0.3          # UKF_LOC: the state is the vehicle, the map (Lmap landmarks) is known
    Lmap = int(rng.choice([5, 20, 35, 80])) if loc else 0
    dev = rng.random() < 0.3                  # the device-buffer entry (slam_step_dev)
    if os.environ.get("SOAK_REPLAY"):   # "ekf|ukf L T B f32 idknown kcap idmax seed [idmin]"
        a = os.environ["SOAK_REPLAY"].split()
        ukf = a[0] == "ukf"; a = a[1:]
        L, T, B = int(a[0]), int(a[1]), int(a[2]); f32 = a[3] == "True"; idknown, kcap, idmax, seed = int(a[4]), int(a[5]), int(a[6]), int(a[7]); t_end = 0
        idmin = int(a[8]) if len(a) > 8 else 0
        Lmap = int(a[9]) if len(a) > 9 else 0; loc = Lmap > 0; dev = len(a) > 10 and a[10] == "1"
    desc = f"{'ukf' if ukf else 'ekf'} L={L} T={T} B={B} f32={f32} idknown={idknown} kcap={kcap} idmax={idmax} seed={seed} idmin={idmin} Lmap={Lmap} dev={int(dev)}"
    if os.environ.get("SOAK_VERBOSE"): print("RUN", desc, flush=True)
    mr = np.random.default_rng(seed)
    cfg = S.default_config(); cfg.landmark_id_is_known = idknown
    if loc:
        mapxy = mr.uniform(-4.0, 4.0, (Lmap, 2))
        f = S.BatchedUKFLoc(B).readParams(cfg); f.set_map(mapxy); f.init(0.0, 0.0, 0.0)
        idmax = int(mr.choice([Lmap, Lmap + 3]))          # ids of the map, now and then one beyond it (SLAM_INST_INDEX_OOR)
        idmin = int(mr.choice([0, 0, 0, -2]))
    else:
        f = (S.BatchedUKF(B, L) if ukf else S.BatchedEKF(B, L, dtype=S.F32 if f32 else S.F64)).readParams(cfg); f.init(0.0, 0.0, 0.0)
    if not ukf and mr.random() < 0.5: f.set_lazy_steps(int(mr.choice([1, 3, 32])))
    es = []
    for b in range(B):
        if loc:
            e = O.OracleUKF(cfg, L_max=1); e.set_loc_map(mapxy)
        else:
            e = O.OracleUKF(cfg, L_max=L) if ukf else O.OracleEKF(cfg, L_max=L, mode=O.MODE_FAST | (O.STORAGE_F32 if f32 else 0))
        e.init(0, 0, 0); es.append(e)
    d_meas, d_cnt = C.c_void_p(), C.c_void_p()
    if dev:
        KSD = kcap + int(mr.integers(0, 8))               # the caller's stride: at least the longest message
        assert _hip.hipMalloc(C.byref(d_meas), B * KSD * 3 * 4) == 0 and _hip.hipMalloc(C.byref(d_cnt), B * 4) == 0
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
        if dev:
            md = np.zeros((B, KSD, 3), dtype=np.float32); md[:, :K] = meas
            cd = ks.astype(np.int32)
            f.sync()                                       # the previous step has read the buffers
            assert _hip.hipMemcpy(d_meas, md.ctypes.data_as(C.c_void_p), md.nbytes, 1) == 0 and _hip.hipMemcpy(d_cnt, cd.ctypes.data_as(C.c_void_p), cd.nbytes, 1) == 0
            f.update_dev(cmd, d_meas.value, d_cnt.value, KSD)
        else:
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
    f.sync(); f.close()
    if dev: _hip.hipFree(d_meas); _hip.hipFree(d_cnt)
    runs += 1
    if why:
        fails += 1
        print(f"MISMATCH {desc}: {'; '.join(why)[:500]}", flush=True)
print(f"{runs} adversarial configurations in {budget:.0f} s, {fails} mismatches", cat)
sys.exit(1 if fails else 0)
