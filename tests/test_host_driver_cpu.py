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


def test_shard_plan_of_the_c_library_is_the_python_one():
    """slam_shard_range (include/slam_multi.h, the single-process multi-GPU host) == parallel.shard_range (bench.py's plan):
    contiguous, complete, ragged remainders on the first shards.  No GPU needed."""
    import ctypes as C
    from live_ekf_slam_amd import _lib
    from live_ekf_slam_amd.parallel import shard_range
    L = _lib.lib()
    for B in (1, 7, 8, 65536, 65537, 100003):
        for world in (1, 2, 3, 8):
            if B < world:
                continue
            nxt = 0
            for s in range(world):
                f, c = C.c_int64(-1), C.c_int64(-1)
                assert L.slam_shard_range(B, s, world, C.byref(f), C.byref(c)) == 0
                assert (f.value, c.value) == shard_range(B, s, world) and f.value == nxt
                nxt = f.value + c.value
            assert nxt == B
    f, c = C.c_int64(), C.c_int64()
    assert L.slam_shard_range(8, 8, 8, C.byref(f), C.byref(c)) != 0 and L.slam_shard_range(8, 0, 0, C.byref(f), C.byref(c)) != 0


def test_multi_create_without_gpu_fails_loudly():
    import ctypes as C
    from live_ekf_slam_amd import _lib
    from live_ekf_slam_amd.config import default_config
    if torch.cuda.is_available():
        return
    L = _lib.lib()
    cfg = default_config(); m = C.c_void_p()
    dev = (C.c_int32 * 1)(0)
    assert L.slam_multi_create(C.byref(cfg), 1, 16, 20, 0, dev, 1, C.byref(m)) != 0 and not m.value
    dup = (C.c_int32 * 2)(0, 0)
    assert L.slam_multi_create(C.byref(cfg), 1, 16, 20, 0, dup, 2, C.byref(m)) == -1   # SLAM_ERR_ARG: a device listed twice
