"""CPU-side checks of the C++ Filter mirror: it compiles against the C ABI and reports a missing GPU as an
exception (the reference's error convention), never by computing something else."""
import os
import subprocess

import torch

from conftest import ROOT


def test_driver_builds_and_fails_loudly_without_gpu():
    from live_ekf_slam_amd.build import build_driver
    exe = build_driver()
    assert os.path.exists(exe)
    if torch.cuda.is_available():
        return
    out = subprocess.run([exe, "run", "ekf", "4", "20", "3"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "driver failed" in out.stderr
