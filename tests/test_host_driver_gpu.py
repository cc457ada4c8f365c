"""The C++ host example over include/slam_filter.hpp (the reference's Filter interface) runs on the GPU."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_filter_driver_runs():
    exe = os.path.join(ROOT, "live_ekf_slam_amd", "filter_driver")
    assert os.path.exists(exe), "build the extension first (__graft_entry__.build())"
    out = subprocess.run([exe, "512", "20", "120"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"mean_avg_err=([0-9.]+) M0=(\d+) timestep=(\d+) P_len=(\d+)", out.stdout)
    assert m, out.stdout
    err, M0, ts, plen = float(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4))
    assert ts == 121 and plen == (3 + 2 * M0) ** 2 and 0.0 < err < 1.0


def test_cpp_pose_graph_driver_runs():
    """iterate() with `filter: pose_graph` + NaiveFilter secondary (localization_node.cpp:124-131) over the C++ mirror."""
    exe = os.path.join(ROOT, "live_ekf_slam_amd", "filter_driver")
    out = subprocess.run([exe, "8", "10", "90", "pose_graph"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"poses=(\d+) solved=(\d) M0=(\d+) result_topic=(\d) x_len=(\d+) conns=(\d+)", out.stdout)
    assert m, out.stdout
    poses, solved, M0, res, xlen, conns = map(int, m.groups())
    assert poses == 90 and solved == 1 and res == 1 and M0 == 3 and xlen == 89 and conns == 30


def test_cpp_ukf_driver_runs():
    exe = os.path.join(ROOT, "live_ekf_slam_amd", "filter_driver")
    out = subprocess.run([exe, "64", "20", "80", "ukf"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"M0=(\d+) timestep=(\d+) P_len=(\d+) X_len=(\d+) sv_len=(\d+)", out.stdout)
    assert m, out.stdout
    M0, ts, plen, xlen, svlen = map(int, m.groups())
    n = 4 + 2 * M0
    assert ts == 81 and plen == n * n and svlen == 3 + 2 * M0 and xlen > 0
