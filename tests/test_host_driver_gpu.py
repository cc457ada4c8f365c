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
