#!/usr/bin/env python3
"""API soak of the batched EKF against the oracle (BIT-EXACT): the same simulated run driven through random interleavings of the
boundary's entry points - update_sim (one step), run_sim (many steps per launch, random chunking), update() with host
measurement buffers recorded from a twin filter (the lazy queue of slam_step), getters in between (get_state, poses,
publishState, status: they flush the queue), a tracked instance (slam_track_instance), checkpoint / resume into a NEW handle
(slam_save_state / slam_load_state), set_lazy_steps.  At the end x, P, ids, M, flags of every instance equal the oracle's.
usage: gpu_soak_api.py [seconds] [seed]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
runs = fails = 0
KS = 16
while time.time() < t_end:
    L = int(rng.choice([3, 8, 20, 35, 50]))
    T = int(rng.integers(5, 160))
    B = int(rng.integers(1, 24))
    seed, sc, inst0 = int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1 << 20))
    f32 = rng.random() < 0.3
    desc = f"L={L} T={T} B={B} seed={seed} scenario={sc} inst0={inst0} f32={f32}"
    if os.environ.get("SOAK_VERBOSE"): print("RUN", desc, flush=True)
    os.environ["SLAM_RUN_CHUNK"] = str(int(rng.choice([1, 2, 5, 16, 1000])))
    lm, cmds = make_scenario(sc, L, T)
    cfg = S.default_config()
    dt = S.F32 if f32 else S.F64
    def fresh():
        g = S.BatchedEKF(B, L, dtype=dt).readParams(cfg); g.set_map(lm); g.set_seed(seed); g.set_instance_offset(inst0); g.init(0, 0, 0)
        return g
    # a twin records the generated measurements of every step (host buffers for update())
    # mode "ext": every step through update() with the host buffers a twin recorded (the external-measurement path never advances
    # the simulator, so it cannot be mixed with generated steps); mode "sim": every step generated on the device
    ext = rng.random() < 0.5
    twin = fresh(); rec = []
    twin.last_meas(KS)   # the first call only switches the dump on
    for t in range(T):
        twin.update_sim(cmds[t]); rec.append(twin.last_meas(KS))
    kmax = max(int(r[1].max()) for r in rec)
    if kmax > KS: ext = False
    f = fresh()
    if rng.random() < 0.5: f.set_lazy_steps(int(rng.choice([1, 2, 8, 32])))
    tracked = int(rng.integers(0, B)) if rng.random() < 0.4 else -1
    if tracked >= 0: f.track_instance(tracked)
    log = []
    t = 0
    while t < T:
        op = rng.choice(["host", "host", "host", "get", "ckpt"] if ext else ["sim1", "simN", "simN", "get", "ckpt"])
        if op == "ckpt" and os.environ.get("SOAK_NO_CKPT"): op = "get"
        if op == "host":
            n = int(rng.integers(1, 12))
            for _ in range(min(n, T - t)):
                f.update(cmds[t], rec[t][0], rec[t][1]); t += 1
            log.append(f"host x{n}")
        elif op == "sim1":
            # device generator: same measurements as the recorded ones (same seed / instance ids / timestep)
            f.update_sim(cmds[t]); t += 1; log.append("sim1")
        elif op == "simN":
            n = int(rng.integers(1, 40)); f.run_sim(cmds[t:t + n]); t = min(T, t + n); log.append(f"run_sim {n}")
        elif op == "get":
            b = int(rng.integers(0, B)); k = int(rng.integers(0, 4))
            if k == 0: f.get_state(b)
            elif k == 1: f.poses()
            elif k == 2: f.publishState(tracked if tracked >= 0 and rng.random() < 0.7 else b)
            else: f.status()
            log.append("get")
        elif op == "ckpt":
            with tempfile.NamedTemporaryFile(suffix=".ckpt") as tf:
                f.save_state(tf.name)
                g = S.BatchedEKF(B, L, dtype=dt).readParams(cfg); g.set_map(lm); g.load_state(tf.name)
            f.close(); f = g
            if tracked >= 0: f.track_instance(tracked)
            log.append("checkpoint -> new handle")
    r = O.run_ekf_batch(lm, cmds, B, L, seed=seed, inst0=inst0, nthreads=8, cfg=cfg, mode=O.MODE_FAST | (O.STORAGE_F32 if f32 else 0))
    twin.close()
    why = []
    if not np.array_equal(f.status(), r["flags"]): why.append(f"flags {f.status().tolist()} vs {r['flags'].tolist()}")
    if not np.array_equal(f.landmark_counts(), r["M"]): why.append("landmark counts")
    for b in range(B):
        n = 3 + 2 * r["M"][b]; sg = f.get_state(b)
        if sg["M"] != r["M"][b] or not (np.array_equal(sg["x"], r["x"][b, :n]) and np.array_equal(sg["P"], r["P"][b, :n * n].reshape(n, n)) and np.array_equal(sg["ids"], r["ids"][b, :r["M"][b]])):
            why.append(f"state of instance {b}"); break
        if sg["timestep"] != T: why.append(f"timestep {sg['timestep']}"); break
    f.close()
    runs += 1
    if why:
        fails += 1
        print(f"MISMATCH {desc} mode={'ext' if ext else 'sim'} chunk={os.environ['SLAM_RUN_CHUNK']} tracked={tracked}: {'; '.join(why)}; ops: {' | '.join(log)[:600]}", flush=True)
print(f"{runs} random API interleavings in {budget:.0f} s, {fails} mismatches")
sys.exit(1 if fails else 0)
