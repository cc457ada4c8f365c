"""The C++ host over include/slam_filter.hpp (the reference's Filter interface + the node harness of localization_node.cpp)
on the GPU: parity against the oracle on the reference simulator's own measurement streams, through BOTH FIFO queues."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "live_ekf_slam_amd", "filter_driver")


def _write_stream(path, g, T, with_map=False):
    with open(path, "w") as f:
        if with_map:
            m = g["map"]
            f.write("map %d " % len(m) + " ".join("%d %.9g %.9g" % (i, np.float32(x), np.float32(y)) for i, (x, y) in enumerate(m)) + "\n")
        for t in range(T):
            k = int(g["meas_count"][t])
            f.write("%.9g %.9g %d " % (g["cmds"][t, 0], g["cmds"][t, 1], k) + " ".join("%.9g" % v for v in g["meas"][t, :k].ravel()) + "\n")


def _read_dump(path):
    raw = open(path, "rb").read()
    out, off = [], 0
    for _ in range(2):
        M, n = np.frombuffer(raw, np.int64, 2, off); off += 16
        x = np.frombuffer(raw, np.float64, n, off); off += 8 * n
        P = np.frombuffer(raw, np.float64, n * n, off).reshape(n, n); off += 8 * n * n
        out.append((int(M), x, P))
    return out


@pytest.mark.parametrize("fixture,L_max,T", [("sim_seed1_L20_T400.npz", 20, 400), ("sim_seed1234_L50_T400.npz", 50, 250)])
def test_cpp_node_harness_ekf_matches_oracle_on_golden_stream(oracle, tmp_path, fixture, L_max, T):
    """iterate() with the command queue AND the measurement queue (localization_node.cpp:108-140; the
    driver delivers the two topics separately, ticks the timer in between and lets the queues run ahead), EKF behind the
    Filter pointer; final x and P of the first and last instance equal the oracle's bit for bit."""
    assert os.path.exists(EXE), "build the extension first (__graft_entry__.build())"
    g = load_golden(fixture)
    stream, dump = str(tmp_path / "stream.txt"), str(tmp_path / "dump.bin")
    _write_stream(stream, g, T)
    out = subprocess.run([EXE, "stream", "ekf", "7", str(L_max), stream, dump], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    m = re.search(r"iterations=(\d+) early_returns=(\d+) queues_left=(\d+)/(\d+)", out.stdout)
    assert m and int(m.group(1)) == T and int(m.group(2)) > 0 and m.group(3) == "0" and m.group(4) == "0", out.stdout
    e = oracle.OracleEKF(L_max=L_max); e.init(0, 0, 0)
    for t in range(T):
        k = int(g["meas_count"][t])
        e.update(g["cmds"][t, 0], g["cmds"][t, 1], g["meas"][t, :k])
    so = e.state()
    for M, x, P in _read_dump(dump):
        assert M == so["M"] and np.array_equal(x, so["x"]) and np.array_equal(P, so["P"])


def test_cpp_node_harness_ukf_loc_waits_for_the_map(oracle, tmp_path):
    """FilterChoice::UKF_LOC: iterate() must not consume anything before trueMapCallback delivered the map
    (localization_node.cpp:113-116, 152-156); afterwards the run equals the oracle's localisation filter."""
    g = load_golden("sim_seed1_L20_T400.npz")
    T = 120
    stream, dump = str(tmp_path / "stream.txt"), str(tmp_path / "dump.bin")
    _write_stream(stream, g, T, with_map=True)
    out = subprocess.run([EXE, "stream", "ukf_loc", "3", "20", stream, dump], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert re.search(r"iterations=%d " % T, out.stdout), out.stdout
    e = oracle.OracleUKF(L_max=1); e.set_loc_map(g["map"].astype(np.float32).astype(np.float64)); e.init(0, 0, 0)
    for t in range(T):
        k = int(g["meas_count"][t])
        e.update(g["cmds"][t, 0], g["cmds"][t, 1], g["meas"][t, :k])
    so = e.state()
    for M, x, P in _read_dump(dump):
        assert M == 0 and np.array_equal(x, so["x"]) and np.array_equal(P, so["P"])


def test_cpp_driver_runs_a_baseline_scenario_without_python(oracle):
    """`filter_driver run ekf`: map + TSP commands from the C++ generators, measurements generated on the device; the mean of
    the per-instance error statistic equals the oracle's for the same scenario, seeds and instances."""
    from live_ekf_slam_amd.scenario import make_scenario
    B, L, T = 96, 20, 150
    out = subprocess.run([EXE, "run", "ekf", str(B), str(L), str(T), "1234"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    m = re.search(r"M0=(\d+) timestep=(\d+) P_len=(\d+) mean_avg_err=([0-9.]+)", out.stdout)
    assert m, out.stdout
    lm, cmds = make_scenario(1234, L, T)
    r = oracle.run_ekf_batch(lm, cmds, B, L, seed=2025, inst0=0, nthreads=4, want_P=False)
    assert int(m.group(1)) == r["M"][0] and int(m.group(2)) == T and int(m.group(3)) == (3 + 2 * r["M"][0]) ** 2
    assert abs(float(m.group(4)) - r["avg_err"].mean()) < 5e-10     # printed with 9 decimals


def test_cpp_pose_graph_driver_runs():
    """iterate() with `filter: pose_graph` + NaiveFilter secondary (localization_node.cpp:124-131) over the C++ mirror."""
    out = subprocess.run([EXE, "pose_graph", "8", "10", "90"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"poses=(\d+) solved=(\d) M0=(\d+) result_topic=(\d) x_len=(\d+) conns=(\d+)", out.stdout)
    assert m, out.stdout
    poses, solved, M0, res, xlen, conns = map(int, m.groups())
    assert poses == 90 and solved == 1 and res == 1 and M0 == 3 and xlen == 89 and conns == 30


def test_cpp_ukf_driver_runs():
    out = subprocess.run([EXE, "run", "ukf", "64", "20", "80"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"M0=(\d+) timestep=(\d+) P_len=(\d+) X_len=(\d+)", out.stdout)
    assert m, out.stdout
    M0, ts, plen, xlen = map(int, m.groups())
    n = 4 + 2 * M0
    assert ts == 80 and plen == n * n and xlen > 0


def test_single_process_multi_gpu_entry_points(oracle):
    """include/slam_multi.h on the one GPU of the box (N = 1, the degenerate plan) and shard invariance: the global batch
    through slam_multi == the same instances through one plain handle == the oracle keyed by global instance id; the gather of
    the error statistics by host concatenation (mode 0) and by RCCL all-gather (mode 1) give the same array."""
    import ctypes as C
    from live_ekf_slam_amd import _lib
    from live_ekf_slam_amd.config import default_config
    from live_ekf_slam_amd.scenario import make_scenario
    import live_ekf_slam_amd as S
    Lc = _lib.lib()
    L, B, T = 20, 96, 90
    lm, cmds = make_scenario(21, L, T)
    lm = np.ascontiguousarray(lm, dtype=np.float64); cm = np.ascontiguousarray(cmds, dtype=np.float32)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)); fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    cfg = default_config(); m = C.c_void_p(); dev = (C.c_int32 * 1)(0)
    _lib.check(Lc.slam_multi_create(C.byref(cfg), S.EKF_SLAM, B, L, S.F64, dev, 1, C.byref(m)))
    assert Lc.slam_multi_devices(m) == 1 and Lc.slam_multi_batch(m) == B
    _lib.check(Lc.slam_multi_set_seed(m, 77)); _lib.check(Lc.slam_multi_set_map(m, dp(lm), L)); _lib.check(Lc.slam_multi_init(m, 0.0, 0.0, 0.0))
    _lib.check(Lc.slam_multi_run_sim(m, fp(cm), T - 1)); _lib.check(Lc.slam_multi_step_sim(m, fp(cm[T - 1]))); _lib.check(Lc.slam_multi_sync(m))
    e0, e1 = np.zeros(B), np.zeros(B); fl = np.zeros(B, dtype=np.int32)
    _lib.check(Lc.slam_multi_error_stats(m, dp(e0), 0))
    _lib.check(Lc.slam_multi_error_stats(m, dp(e1), 1))      # RCCL: ncclCommInitAll over the one device + ncclAllGather
    _lib.check(Lc.slam_multi_status(m, ip(fl)))
    r = oracle.run_ekf_batch(lm, cmds, B, L, seed=77, inst0=0, nthreads=8)
    assert np.array_equal(e0, r["avg_err"]) and np.array_equal(e1, e0) and np.array_equal(fl, r["flags"])
    nmax = 3 + 2 * L
    for g in (0, 41, B - 1):
        x = np.zeros(nmax); P = np.zeros(nmax * nmax); M = C.c_int32(); ts = C.c_int32(); ids = np.zeros(L, dtype=np.int32)
        _lib.check(Lc.slam_multi_get_state(m, g, dp(x), dp(P), C.byref(M), ip(ids), C.byref(ts)))
        n = 3 + 2 * int(r["M"][g])
        assert M.value == r["M"][g] and ts.value == T and np.array_equal(x[:n], r["x"][g, :n]) and np.array_equal(P[:n * n], r["P"][g, :n * n])
    # a shard of a larger plan: shard 1 of 3 over the same global batch, run through a plain handle, equals the slice
    f1, c1 = C.c_int64(), C.c_int64()
    _lib.check(Lc.slam_shard_range(B, 1, 3, C.byref(f1), C.byref(c1)))
    f = S.BatchedEKF(int(c1.value), L).readParams(); f.set_map(lm); f.set_seed(77); f.set_instance_offset(int(f1.value)); f.init(0, 0, 0)
    f.run_sim(cmds)
    assert np.array_equal(f.error_stats(), e0[f1.value:f1.value + c1.value])
    f.close()
    _lib.check(Lc.slam_multi_destroy(m))


def test_cpp_driver_run_multi():
    out = subprocess.run([EXE, "run_multi", "ekf", "4096", "20", "60", "1", "1234", "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-1500:]
    a = re.search(r"mean_avg_err=([0-9.eE+-]+)", out.stdout)
    out2 = subprocess.run([EXE, "run", "ekf", "4096", "20", "60", "1234"], capture_output=True, text=True, timeout=600)
    b = re.search(r"mean_avg_err=([0-9.eE+-]+)", out2.stdout)
    assert a and b and a.group(1) == b.group(1), (out.stdout, out2.stdout)
    assert "flagged=0" in out.stdout and "gpus=1" in out.stdout


def test_publish_state_every_tick_from_the_tracked_instance(oracle):
    """The reference's loop is update -> publishState every tick (localization_node.cpp:131-139).  With slam_track_instance the
    published instance runs in a one-instance shadow and slam_get_state answers from it while the batch's steps stay queued:
    every tick's state must be bit-identical to the oracle's (and so to a run that flushes the whole batch every tick), for
    the host-message entry point and for the device generator, with tracking switched on in the middle of a run."""
    import ctypes as C
    import live_ekf_slam_amd as S
    from live_ekf_slam_amd import _lib
    g = load_golden("sim_seed1234_L50_T400.npz")
    L, B, T = 50, 256, 150
    f = S.BatchedEKF(B, L).readParams(); f.init(0, 0, 0)
    f.set_lazy_steps(32)
    o = oracle.OracleEKF(L_max=L); o.init(0, 0, 0)
    inst = 5
    flushes = 0
    for t in range(T):
        if t == 20:
            f.track_instance(inst)          # mid-run: the instance's current state is copied into the shadow
        k = int(g["meas_count"][t])
        f.update(g["cmds"][t], g["meas"][t, :k])
        o.update(g["cmds"][t, 0], g["cmds"][t, 1], g["meas"][t, :k])
        if t >= 20:
            sg, so = f.get_state(inst), o.state()          # publishState(inst) every tick
            assert sg["M"] == so["M"] and sg["timestep"] == t + 1
            assert np.array_equal(sg["x"], so["x"]) and np.array_equal(sg["P"], so["P"]), t
    # the batch itself: every instance got the same messages, so all of them equal the oracle at the end
    for b in (0, inst, B - 1):
        sg, so = f.get_state(b), o.state()
        assert np.array_equal(sg["x"], so["x"]) and np.array_equal(sg["P"], so["P"])
    f.track_instance(-1)
    f.close()
    # device generator + per-instance noise: the shadow is keyed by the GLOBAL instance id
    from live_ekf_slam_amd.scenario import make_scenario
    lm, cmds = make_scenario(3, 20, 80)
    f = S.BatchedEKF(128, 20).readParams(); f.set_map(lm); f.set_seed(9); f.set_instance_offset(1000); f.init(0, 0, 0)
    f.track_instance(77)
    for t in range(80):
        f.update_sim(cmds[t])
        if t % 7 == 0 or t == 79:
            r = oracle.run_ekf_batch(lm, cmds[:t + 1], 1, 20, seed=9, inst0=1077)
            sg = f.get_state(77); n = 3 + 2 * int(r["M"][0])
            assert sg["M"] == r["M"][0] and np.array_equal(sg["x"], r["x"][0, :n]) and np.array_equal(sg["P"].ravel(), r["P"][0, :n * n])
    r = oracle.run_ekf_batch(lm, cmds, 128, 20, seed=9, inst0=1000, nthreads=8)
    assert np.array_equal(f.error_stats(), r["avg_err"])
    f.close()


def test_tracked_instance_reads_the_callers_device_buffers_before_they_are_reused(oracle):
    """slam_step_dev with a tracked instance runs the shadow's step on the shadow's OWN stream, reading the caller's
    d_meas / d_count.  The header allows the caller to overwrite those buffers by work enqueued on the handle's stream right after
    the call - so the handle's stream must wait for the shadow's read.  Here the buffers are scrubbed on the stream immediately
    after every call; the tracked instance must still equal the oracle (and the batch) at every tick.  (Device buffers through the
    HIP runtime libslam_hip.so itself links, not through torch: two HIP runtimes in one process do not share stream handles.)"""
    import ctypes as C
    import live_ekf_slam_amd as S
    g = load_golden("sim_seed1234_L50_T400.npz")
    L, B, T, inst, ks = 50, 4096, 60, 4000, 8
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    stream, d_meas, d_cnt = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreate(C.byref(stream)) == 0
    assert hip.hipMalloc(C.byref(d_meas), B * ks * 3 * 4) == 0 and hip.hipMalloc(C.byref(d_cnt), B * 4) == 0
    f = S.BatchedEKF(B, L).readParams(); f.set_stream(stream.value); f.init(0, 0, 0)
    f.track_instance(inst)
    o = oracle.OracleEKF(L_max=L); o.init(0, 0, 0)
    for t in range(T):
        k = int(g["meas_count"][t])
        m = np.zeros((ks, 3), dtype=np.float32); m[:k] = g["meas"][t, :k]
        hm = np.ascontiguousarray(np.broadcast_to(m, (B, ks, 3))); hc = np.full(B, k, dtype=np.int32)
        assert hip.hipMemcpyAsync(d_meas, hm.ctypes.data_as(C.c_void_p), hm.nbytes, 1, stream) == 0
        assert hip.hipMemcpyAsync(d_cnt, hc.ctypes.data_as(C.c_void_p), hc.nbytes, 1, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0            # (pageable host arrays: the copies are done before they go out of scope)
        f.update_dev(g["cmds"][t], d_meas.value, d_cnt.value, ks)
        # the caller reuses its buffers at once, on the handle's stream: 0x41 bytes = ids / ranges / bearings of 12.08, counts of 1 094 795 585
        assert hip.hipMemsetAsync(d_meas, 0x41, B * ks * 3 * 4, stream) == 0 and hip.hipMemsetAsync(d_cnt, 0x41, B * 4, stream) == 0
        o.update(g["cmds"][t, 0], g["cmds"][t, 1], g["meas"][t, :k])
        sg, so = f.get_state(inst), o.state()                   # answered by the shadow
        assert sg["M"] == so["M"] and np.array_equal(sg["x"], so["x"]) and np.array_equal(sg["P"], so["P"]), t
    f.track_instance(-1)
    sg, so = f.get_state(inst), o.state()                       # the batch's own copy of the instance
    assert np.array_equal(sg["x"], so["x"]) and np.array_equal(sg["P"], so["P"])
    f.close()
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipFree(d_meas); hip.hipFree(d_cnt)


def test_a_corrupted_checkpoint_is_refused_before_it_reaches_the_device(tmp_path):
    """slam_load_state used to check the header only; a landmark count beyond L_max (or a state capacity that does
    not match) made the next step kernel index past the instance's slab.  Now the per-instance counters are validated on the
    host and the handle keeps its state."""
    import live_ekf_slam_amd as S
    from live_ekf_slam_amd.scenario import make_scenario
    L, B = 20, 16
    lm, cmds = make_scenario(5, L, 30)
    f = S.BatchedEKF(B, L).readParams(); f.set_map(lm); f.set_seed(3); f.init(0, 0, 0); f.run_sim(cmds[:20])
    good = tmp_path / "good.ckpt"; f.save_state(good)
    raw = bytearray(open(good, "rb").read())
    # locate the landmark-count column: the only run of B int32 equal to landmark_counts()
    M = f.landmark_counts().astype(np.int32)
    pos = bytes(raw).find(M.tobytes())
    assert pos > 0
    bad = bytearray(raw); bad[pos + 4 * 3:pos + 4 * 4] = np.int32(L + 7).tobytes()
    p_bad = tmp_path / "bad.ckpt"; open(p_bad, "wb").write(bad)
    before = f.get_state(3)
    with pytest.raises(S.SlamError, match="landmark count"):
        f.load_state(p_bad)
    neg = bytearray(raw); neg[pos:pos + 4] = np.int32(-1).tobytes()
    p_neg = tmp_path / "neg.ckpt"; open(p_neg, "wb").write(neg)
    with pytest.raises(S.SlamError, match="landmark count"):
        f.load_state(p_neg)
    after = f.get_state(3)
    assert np.array_equal(before["x"], after["x"]) and np.array_equal(before["P"], after["P"])   # nothing was copied
    f.load_state(good); f.run_sim(cmds[20:]); assert np.all(f.status() == 0)
    f.close()
