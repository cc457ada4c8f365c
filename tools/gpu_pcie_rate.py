#!/usr/bin/env python3
"""PCIe-inclusive rate of the per-step boundary: slam_step with HOST measurement buffers ([B][k][3] f32 + counts copied
to the device every step) vs slam_step_dev (same buffers already resident) vs slam_step_sim (generated on the device),
L=50, batch=65536, steady state.  The headline `value` of bench.py never includes this copy."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import live_ekf_slam_amd as S
from live_ekf_slam_amd.scenario import make_scenario

L, B, K, KS = 50, 65536, int(os.environ.get('PCIE_K', '40')), 8
lm, cmds = make_scenario(1234, L, max(400, 200 + K))
def fresh():
    g = S.BatchedEKF(B, L).readParams(); g.set_map(lm); g.init(0, 0, 0)
    g.set_vision(1e9, -4.0, 4.0); g.update_sim(cmds[0]); g.set_vision(3.0, -1.57, 1.57)
    g.run_sim(cmds[1:20]); g.sync()
    return g
# record K steps of generated measurements on a throw-away filter (the measurement dump switches the multi-step paths off)
frec = fresh()
rec = []
frec.last_meas(KS)   # the first call only switches the dump on
for t in range(20, 20 + K):
    frec.update_sim(cmds[t]); rec.append(frec.last_meas(KS))
frec.close()
f = fresh()
hip = C.CDLL("libamdhip64.so")
def run(label, fn):
    f.sync(); t0 = time.perf_counter()
    for i in range(K): fn(i)
    f.sync(); dt = time.perf_counter() - t0
    print(f"{label:34s} {dt / K * 1e3:7.3f} ms/step  {B * K / dt / 1e6:6.2f} M steps/s")
from live_ekf_slam_amd import _lib
Lc = _lib.lib()
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)); ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
cm = [np.ascontiguousarray(c, dtype=np.float32) for c in cmds]
run("slam_step_sim (device generator)", lambda i: _lib.check(Lc.slam_step_sim(f.h, fp(cm[60 + i]))))
dm, dc = C.c_void_p(), C.c_void_p()
hip.hipMalloc(C.byref(dm), B * KS * 12); hip.hipMalloc(C.byref(dc), B * 4)
hip.hipMemcpy(dm, rec[0][0].ctypes.data_as(C.c_void_p), B * KS * 12, 1); hip.hipMemcpy(dc, rec[0][1].ctypes.data_as(C.c_void_p), B * 4, 1)
dms, dcs = [], []
for i in range(K):   # every step's message resident on the device: the GPU time of the host-measurement step without any copy
    a_, b_ = C.c_void_p(), C.c_void_p()
    hip.hipMalloc(C.byref(a_), B * KS * 12); hip.hipMalloc(C.byref(b_), B * 4)
    hip.hipMemcpy(a_, rec[i][0].ctypes.data_as(C.c_void_p), B * KS * 12, 1); hip.hipMemcpy(b_, rec[i][1].ctypes.data_as(C.c_void_p), B * 4, 1)
    dms.append(a_); dcs.append(b_)
g2 = fresh()
g2.set_lazy_steps(32)   # the queue of slam_step_dev is opt-in (round 3): by default every call enqueues its step at once
import time as _t
def run2(label, fn, flt):
    flt.sync(); t0 = _t.perf_counter(); hostt = 0.0
    for i in range(K):
        h0 = _t.perf_counter(); fn(i); hostt += _t.perf_counter() - h0
    flt.sync(); dt = _t.perf_counter() - t0
    print(f"{label:34s} {dt / K * 1e3:7.3f} ms/step  {B * K / dt / 1e6:6.2f} M steps/s   (host time inside the calls {hostt / K * 1e3:.3f} ms/step)")
run2("slam_step_dev (each step's own msg)", lambda i: _lib.check(Lc.slam_step_dev(g2.h, fp(cm[20 + i]), dms[i], dcs[i], KS)), g2)
if os.environ.get("SLAM_EAGER_FLUSH") == "0":   # packing cost alone: 15 calls that only fill the queue
    gp = fresh(); t0 = _t.perf_counter()
    for i in range(15): _lib.check(Lc.slam_step(gp.h, fp(cm[20 + i]), fp(rec[i][0]), ip(rec[i][1]), KS))
    print(f"slam_step host packing alone: {(_t.perf_counter() - t0) / 15 * 1e3:.3f} ms per call"); gp.close()
g3 = fresh()
# two untimed calls + sync first: each of the two message queues pins ~50 MB of host memory on first use (several ms, once per handle)
for i in range(2): _lib.check(Lc.slam_step(g3.h, fp(cm[20 + i]), fp(rec[i][0]), ip(rec[i][1]), KS)); g3.sync()
K0 = K; K = K - 2
run2("slam_step (host buffers, queued)", lambda i: _lib.check(Lc.slam_step(g3.h, fp(cm[22 + i]), fp(rec[2 + i][0]), ip(rec[2 + i][1]), KS)), g3)
K = K0
g4 = fresh()
g4.sync(); t0 = time.perf_counter(); g4.run_sim(cmds[20:20 + K]); g4.sync(); dt = time.perf_counter() - t0
print(f"{'run_sim over the same timesteps':34s} {dt / K * 1e3:7.3f} ms/step  {B * K / dt / 1e6:6.2f} M steps/s   (one launch, device generator)")
g4.close()
g5 = fresh()
run2("slam_step_sim over the same steps", lambda i: _lib.check(Lc.slam_step_sim(g5.h, fp(cm[20 + i]))), g5)
g5.close()
f.set_lazy_steps(32)
run("slam_step_dev (one msg repeated)", lambda i: _lib.check(Lc.slam_step_dev(f.h, fp(cm[140 + i]), dm, dc, KS)))
# ---- the reference's loop: update -> publishState every tick (localization_node.cpp:131-139) ----
def tick_loop(label, flt, track):
    if track: flt.track_instance(0)
    for i in range(2): _lib.check(Lc.slam_step(flt.h, fp(cm[20 + i]), fp(rec[i][0]), ip(rec[i][1]), KS)); flt.sync()
    n = K - 2
    flt.sync(); t0 = _t.perf_counter()
    for i in range(n):
        _lib.check(Lc.slam_step(flt.h, fp(cm[22 + i]), fp(rec[2 + i][0]), ip(rec[2 + i][1]), KS))
        st = flt.publishState(0)
    flt.sync(); dt = _t.perf_counter() - t0
    print(f"{label:34s} {dt / n * 1e3:7.3f} ms/step  {B * n / dt / 1e6:6.2f} M steps/s   (last published timestep {st['timestep']})")
g6 = fresh(); tick_loop("slam_step + publishState(0), tracked", g6, True); g6.close()
g7 = fresh(); tick_loop("slam_step + publishState(0), flushing", g7, False); g7.close()
kmax = max(int(r[1].max()) for r in rec)
print(f"caller buffers per step: {B * KS * 12 / 1e6:.1f} MB measurements (stride {KS}) + {B * 4 / 1e6:.2f} MB counts in pageable host memory; "
      f"slam_step packs them to stride max(count) = {kmax} into pinned staging and copies on its own stream")
f.close()
