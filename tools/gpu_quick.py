#!/usr/bin/env python3
"""Quick on-GPU bring-up check (not a test): math probe, EXT/SIM parity vs the oracle, rough timing."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import live_ekf_slam_amd as S
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.scenario import make_scenario
from oracle import oracle as O

def dp(a): return a.ctypes.data_as(C.POINTER(C.c_double))

def math_probe():
    rng = np.random.default_rng(1)
    n = 200000
    a = np.concatenate([rng.uniform(-8, 8, n // 2), rng.uniform(-300, 300, n // 2)])
    b = rng.uniform(-8, 8, n)
    out = np.zeros(8 * n)
    _lib.check(_lib.lib().slam_math_probe(dp(a), dp(b), dp(out), n, 0))
    out = out.reshape(n, 8)
    L = O.lib()
    s = np.zeros(n); c = np.zeros(n); at = np.zeros(n); rem = np.zeros(n)
    L.orc_det_sincos(dp(a), dp(s), dp(c), n); L.orc_det_atan2(dp(a), dp(b), dp(at), n); L.orc_libm_remainder2pi(dp(a), dp(rem), n)
    ref = [s, c, at, rem, np.sqrt(np.abs(a)), a / b, a.astype(np.float32).astype(np.float64)]
    names = ["sin", "cos", "atan2", "remainder", "sqrt", "div", "f32cvt"]
    for i, (nm, r) in enumerate(zip(names, ref)):
        print(f"  math {nm}: mismatches {int(np.sum(out[:, i] != r))} / {n}")
    nz = np.zeros(2); ok = 0
    for i in range(0, n, 997):
        L.orc_noise_pair(12345, i, 7, 3, dp(nz)); ok += (out[i, 7] == nz[0] + nz[1])
    print("  philox u53 matches:", ok, "of", len(range(0, n, 997)))

def ext_parity(fix, L_max, B=8, wpf=None):
    if wpf: os.environ["SLAM_WAVES_PER_FILTER"] = str(wpf)
    g = np.load(fix); T = int(g["T"])
    f = S.BatchedEKF(B, L_max).readParams(); f.init(0, 0, 0)
    e = O.OracleEKF(L_max=L_max, math=O.MATH_DET, mode=O.MODE_FAST); e.init(0, 0, 0)
    bad = None
    for t in range(T):
        k = int(g["meas_count"][t]); m = g["meas"][t, :k]
        f.update(g["cmds"][t], m); e.update(g["cmds"][t, 0], g["cmds"][t, 1], m)
        if t % 25 == 24 or t == T - 1:
            so = e.state()
            for b in (0, B - 1):
                sg = f.get_state(b)
                if sg["M"] != so["M"] or not np.array_equal(sg["x"], so["x"]) or not np.array_equal(sg["P"], so["P"]):
                    dx = np.abs(sg["x"] - so["x"]).max() if sg["M"] == so["M"] else -1
                    dP = np.abs(sg["P"] - so["P"]).max() if sg["M"] == so["M"] else -1
                    bad = (t, b, sg["M"], so["M"], dx, dP); break
            if bad: break
    print(f"  EXT parity {os.path.basename(fix)} L_max={L_max} wpf={wpf}: ", "BIT-EXACT all checkpoints" if bad is None else f"MISMATCH at {bad}", " flags", f.status()[:2])
    f.close()

def sim_parity(L, T, B, wpf=None, seed=77):
    if wpf: os.environ["SLAM_WAVES_PER_FILTER"] = str(wpf)
    lm, cmds = make_scenario(1234, L, T)
    f = S.BatchedEKF(B, L).readParams(); f.set_map(lm); f.set_seed(seed); f.set_instance_offset(1000); f.init(0, 0, 0)
    t0 = time.time(); f.run_sim(cmds); f.sync(); tg = time.time() - t0
    r = O.run_ekf_batch(lm, cmds, B, L, seed=seed, inst0=1000, nthreads=8)
    nb = 0; worst = 0
    for b in range(B):
        sg = f.get_state(b); n = 3 + 2 * r["M"][b]
        okb = sg["M"] == r["M"][b] and np.array_equal(sg["x"], r["x"][b, :n]) and np.array_equal(sg["P"].ravel(), r["P"][b, :n * n]) and np.array_equal(sg["ids"], r["ids"][b, :r["M"][b]])
        if not okb:
            nb += 1
            if sg["M"] == r["M"][b]: worst = max(worst, np.abs(sg["x"] - r["x"][b, :n]).max())
    print(f"  SIM parity L={L} T={T} B={B} wpf={wpf}: state mismatches {nb}/{B} worst dx {worst:.3e}; truth equal {np.array_equal(f.truth(), r['truth'])}; avg_err equal {np.array_equal(f.error_stats(), r['avg_err'])}; mean err {f.error_stats().mean():.4f}; flags {np.unique(f.status())}; gpu {tg:.2f}s oracle {r['seconds']:.2f}s meanM {r['M'].mean():.1f}")
    f.close()

def timing(L, B, steps, wpf=None):
    if wpf: os.environ["SLAM_WAVES_PER_FILTER"] = str(wpf)
    lm, cmds = make_scenario(1234, L, 400)
    f = S.BatchedEKF(B, L).readParams(); f.set_map(lm); f.init(0, 0, 0)
    f.set_vision(1e9, -4.0, 4.0); f.update_sim(cmds[0]); f.set_vision(3.0, -1.57, 1.57)   # discover all landmarks
    f.run_sim(cmds[1:50]); f.sync()
    M = f.landmark_counts()
    t0 = time.time(); f.run_sim(cmds[50:50 + steps]); f.sync(); dt = time.time() - t0
    ab = f.algorithmic_bytes()
    print(f"  timing L={L} B={B} wpf={wpf}: M min/mean {M.min()}/{M.mean():.1f}  {dt / steps * 1e3:.3f} ms/step  {B * steps / dt / 1e6:.3f} M steps/s  {ab / (dt / steps) / 1e12:.3f} TB/s algorithmic  flags {np.unique(f.status())}")
    f.close()

if __name__ == "__main__":
    which = sys.argv[1:] or ["math", "ext", "sim", "time"]
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if "math" in which: math_probe()
    if "ext" in which:
        ext_parity(os.path.join(here, "tests/golden/sim_seed0_L20_T1000.npz"), 20)
        ext_parity(os.path.join(here, "tests/golden/sim_seed2_L50_T1000.npz"), 50, wpf=4)
        ext_parity(os.path.join(here, "tests/golden/sim_seed2_L50_T1000.npz"), 50, wpf=8)
    if "sim" in which:
        sim_parity(20, 300, 64)
        sim_parity(50, 300, 64, wpf=4)
        sim_parity(50, 300, 64, wpf=8)
    if "time" in which:
        timing(20, 65536, 50, wpf=2)
        timing(20, 65536, 50, wpf=4)
        timing(50, 65536, 30, wpf=4)
        timing(50, 65536, 30, wpf=8)
