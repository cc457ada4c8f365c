"""include/slam_filter_ros.hpp - the adapter that derives from the reference's own abstract class (filter.h:54-77) with its exact
virtual signatures (YAML::Node, ros::NodeHandle, Eigen::VectorXd, the ROS message pointers).  ROS / yaml-cpp / Eigen do not exist in
this image, so the adapter is compiled against the TEST stand-ins of tests/ros_stub/ros_stub.hpp and driven by
tests/ros_stub/ros_adapter_driver.cpp exactly as localization_node.cpp drives its filter (factory, readParams(config),
setupStatePublisher(node), init, update + publishState per tick, getStateVector) through std::unique_ptr<Filter>."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

SRC = os.path.join(ROOT, "tests", "ros_stub", "ros_adapter_driver.cpp")
LIBDIR = os.path.join(ROOT, "live_ekf_slam_amd")


def _build(tmp_path):
    exe = str(tmp_path / "ros_adapter_driver")
    out = subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", SRC, "-o", exe, "-L" + LIBDIR, "-lslam_hip", "-Wl,-rpath," + LIBDIR],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return exe


def test_adapter_compiles_against_the_interface_and_fails_loudly_without_gpu(tmp_path):
    """Every override matches a pure virtual of the interface (the driver instantiates both adapters through std::unique_ptr<Filter>:
    a signature that differed would leave the class abstract and the build would fail); without a GPU readParams throws."""
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: the functional test below runs the adapter")
    stream = tmp_path / "s.txt"; stream.write_text("0.1 0.0 0\n")
    out = subprocess.run([exe, "ekf", "2", "20", str(stream), str(tmp_path / "d.bin")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 3 and "slam_batch:" in out.stderr


def _read_dump(path):
    raw = open(path, "rb").read()
    off, recs = 0, []
    while True:
        n = int(np.frombuffer(raw, np.int32, 1, off)[0]); off += 4
        if n < 0:                        # the trailing getStateVector record (float64)
            return recs, np.frombuffer(raw, np.float64, -n, off)
        recs.append(np.frombuffer(raw, np.float32, n, off)); off += 4 * n


@pytest.mark.gpu
@pytest.mark.parametrize("kind,fixture,L_max,T", [("ekf", "sim_seed1_L20_T400.npz", 20, 150), ("ukf", "sim_seed1_L20_T400.npz", 20, 60)])
def test_adapter_publishes_the_oracles_state_every_tick(oracle, tmp_path, kind, fixture, L_max, T):
    """A golden measurement stream of the reference simulator through the adapter, behind the Filter pointer: the message published
    after EVERY tick (EKFState / UKFState: timestep, pose, [id, x, y] triplets, P row by row, float32 wire format) equals the
    oracle's state cast the same way, and getStateVector() / lm_IDs at the end equal it in fp64."""
    exe = _build(tmp_path)
    g = load_golden(fixture)
    stream, dump = tmp_path / "stream.txt", tmp_path / "dump.bin"
    with open(stream, "w") as f:
        for t in range(T):
            k = int(g["meas_count"][t])
            f.write("%.9g %.9g %d " % (g["cmds"][t, 0], g["cmds"][t, 1], k) + " ".join("%.9g" % v for v in g["meas"][t, :k].ravel()) + "\n")
    out = subprocess.run([exe, kind, "5", str(L_max), str(stream), str(dump)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    assert ("topic=/state/%s" % kind) in out.stdout and ("ticks=%d" % T) in out.stdout
    recs, sv = _read_dump(str(dump))
    assert len(recs) == T
    e = (oracle.OracleEKF(L_max=L_max) if kind == "ekf" else oracle.OracleUKF(L_max=L_max)); e.init(0, 0, 0)
    base = 3 if kind == "ekf" else 4
    for t in range(T):
        k = int(g["meas_count"][t])
        e.update(g["cmds"][t, 0], g["cmds"][t, 1], g["meas"][t, :k])
        so = e.state()
        M = so["M"]; n = base + 2 * M
        x = so["x"]
        yaw = x[2] if kind == "ekf" else np.remainder(np.arctan2(x[3], x[2]) + np.pi, 2 * np.pi) - np.pi
        lm = np.stack([so["ids"][:M].astype(np.float32), x[base:n:2].astype(np.float32), x[base + 1:n:2].astype(np.float32)], axis=1).ravel()
        want = np.concatenate([np.array([t + 1, x[0], x[1], yaw, M], dtype=np.float32), lm, so["P"].astype(np.float32).ravel()])
        got = recs[t]
        assert got.size == want.size, (t, got.size, want.size)
        if kind == "ukf":   # atan2 of libm on the host side of the adapter vs numpy: compare the angle loosely, the rest exactly
            assert abs(float(got[3]) - float(want[3])) < 1e-6
            got = got.copy(); got[3] = want[3]
        assert np.array_equal(got, want), (t, np.flatnonzero(got != want)[:5])
    so = e.state()
    if kind == "ekf":
        assert np.array_equal(sv, so["x"])
    else:
        assert sv.size == so["x"].size - 1 and np.array_equal(sv[:2], so["x"][:2]) and np.array_equal(sv[3:], so["x"][4:])
    assert ("lm_IDs=%d" % so["M"]) in out.stdout
