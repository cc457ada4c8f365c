"""The C-ABI library loads, exports every symbol include/slam_batch.h declares, and fails loudly without a GPU.
No compute calls here (no GPU in the CPU test run)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from live_ekf_slam_amd import _lib
from live_ekf_slam_amd.config import SlamConfig, default_config


def _declared_symbols():
    out = set()
    for hdr in ("slam_batch.h", "slam_pgs.h", "slam_multi.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        out |= set(re.findall(r"\b((?:slam|pgs)_[a-z_0-9]+)\s*\(", txt))
    return sorted(out)


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 28
    for s in syms:
        assert hasattr(L, s), f"libslam_hip.so does not export {s}"
    assert set(syms) == set(_lib.SIGNATURES), "python binding and header disagree"
    assert b"gfx950" in L.slam_version()


def test_struct_layout_matches_header(tmp_path):
    """sizeof/offsetof of slam_config as gcc sees the header == the ctypes mirror."""
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "slam_batch.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(slam_config), offsetof(slam_config, W_00),'
                   ' offsetof(slam_config, d_max), offsetof(slam_config, init_yaw), offsetof(slam_config, ukf_float_trig));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(SlamConfig), SlamConfig.W_00.offset, SlamConfig.d_max.offset,
                   SlamConfig.init_yaw.offset, SlamConfig.ukf_float_trig.offset]


def test_config_default_and_yaml_reader(tmp_path):
    L = _lib.lib()
    c = SlamConfig()
    assert L.slam_config_default(C.byref(c)) == 0
    d = default_config()
    for name, _ in SlamConfig._fields_:
        if name == "reserved":
            continue
        assert getattr(c, name) == getattr(d, name), name
    # the quirk switches of round 5 default to the reference's behaviour
    assert (c.ekf_abs_is_int, c.ekf_landmark_from_x_pred, c.ukf_accumulate_zest1, c.ukf_sensing_yaw_from_sigma) == (0, 0, 0, 0)
    y = tmp_path / "params.yaml"
    y.write_text(
        "filter: \"ekf_slam\"\ndt: 0.05\ninit_pose:\n  x: 1.5\n  y: -2.0\n  yaw: 0.25\n"
        "constraints:\n  commands:\n    d_max: 0.2 # comment\n    th_max: 0.1\n  vision:\n    range_max: 4.0\n"
        "    fov_min: -1.0\n    fov_max: 1.0\n  measurements:\n    landmark_id_is_known: false\n"
        "    min_landmark_separation: 0.3\nprocess_noise:\n  mean:\n    v_d: 0.01\n    v_th: 0.0\n  cov:\n"
        "    V_00: 0.02\n    V_11: 0.002\nsensing_noise:\n  mean:\n    w_r: 0.0\n    w_b: 0.02\n  cov:\n"
        "    W_00: 0.03\n    W_11: 0.04\nmap:\n  bound: 10.0\n  min_landmark_separation: 0.05\n")
    assert L.slam_config_load(C.byref(c), str(y).encode()) == 0
    assert (c.init_x, c.init_y, c.init_yaw) == (1.5, -2.0, 0.25)
    assert (c.d_max, c.th_max, c.range_max, c.fov_min, c.fov_max) == (0.2, 0.1, 4.0, -1.0, 1.0)
    assert c.landmark_id_is_known == 0 and abs(c.min_landmark_separation - 0.3) < 1e-7   # not the map: key
    assert (c.V_00, c.V_11, c.W_00, c.W_11) == (0.02, 0.002, 0.03, 0.04)
    assert abs(c.v_d - 0.01) < 1e-9 and abs(c.w_b - 0.02) < 1e-9
    assert L.slam_config_load(C.byref(c), b"/nonexistent/params.yaml") == -5


def test_reference_params_yaml_shape_is_readable():
    """The reader accepts the reference's own key layout (values = params.yaml:19-52)."""
    L = _lib.lib()
    c = SlamConfig(); L.slam_config_default(C.byref(c))
    import tempfile
    txt = ("constraints:\n  commands:\n    d_max: 0.1 # max forward motion\n    th_max: 0.0546\n  vision:\n"
           "    range_max: 3.0\n    fov_min: -1.57 #-3.14\n    fov_max: 1.57\n  measurements:\n"
           "    landmark_id_is_known: true\n    min_landmark_separation: 0.1\n")
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        f.write(txt)
    assert L.slam_config_load(C.byref(c), f.name.encode()) == 0
    assert c.fov_min == -1.57 and c.landmark_id_is_known == 1
    os.unlink(f.name)


def test_bad_arguments_return_error_codes():
    L = _lib.lib()
    assert L.slam_config_default(None) == -1
    h = C.c_void_p()
    c = default_config()
    assert L.slam_create(C.byref(c), 1, 0, 20, 0, 0, C.byref(h)) == -1          # batch 0
    assert L.slam_create(C.byref(c), 1, 4, 1001, 0, 0, C.byref(h)) == -3        # above kernel capacity (EKF fp64: 1000 landmarks, the HBM-streamed class)
    assert L.slam_create(C.byref(c), 1, 4, 1001, 1, 0, C.byref(h)) == -3        # fp32 storage: the same limit (beyond 50 landmarks the streamed class, since round 5)
    assert b"limit" in L.slam_last_error()
    assert L.slam_create(C.byref(c), 3, 4, 20, 1, 0, C.byref(h)) == -3          # f32 storage: EKF only
    assert L.slam_create(C.byref(c), 9, 4, 20, 0, 0, C.byref(h)) == -1          # unknown filter kind
    assert L.slam_step_sim(None, None) == -1
    assert L.slam_destroy(None) == 0
    # the quirk switches live where `reserved` fields were until round 4: anything but 0 / 1 is refused, not read as "switch the quirk off"
    for name in ("ekf_abs_is_int", "ekf_landmark_from_x_pred", "ukf_accumulate_zest1", "ukf_sensing_yaw_from_sigma"):
        c2 = default_config()
        setattr(c2, name, 7)
        assert L.slam_create(C.byref(c2), 1, 4, 20, 0, 0, C.byref(h)) == -1, name
        assert b"quirk switch" in L.slam_last_error()
    # pgs_last_solve_paths keeps its four-double ABI (the eight-entry form is _v2 with an explicit length)
    assert L.pgs_last_solve_paths_v2(None, None, 8) == -1 and L.pgs_last_solve_paths(None, None) == -1


def test_multi_handle_rejects_a_device_listed_twice():
    """include/slam_multi.h: 'a device may appear once' - two shards on one device would share nothing but fight for it; the call must
    fail loudly BEFORE any handle is created (no GPU needed to see it)."""
    L = _lib.lib()
    c = default_config()
    m = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    L.slam_multi_create.restype = C.c_int
    assert L.slam_multi_create(C.byref(c), 1, C.c_int64(64), 20, 0, devs, 2, C.byref(m)) == -1
    assert b"listed twice" in L.slam_last_error()
    assert not m.value
    one = (C.c_int * 1)(0)
    assert L.slam_multi_create(C.byref(c), 1, C.c_int64(0), 20, 0, one, 1, C.byref(m)) == -1   # fewer instances than devices


def test_no_gpu_fails_loudly_not_silently():
    """Without a HIP device the product must raise, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import live_ekf_slam_amd as S
    f = S.BatchedEKF(4, 20)
    with pytest.raises(S.SlamError):
        f.readParams()
    with pytest.raises(S.SlamError):
        f.update((0.1, 0.0), [])


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under live_ekf_slam_amd/ or include/ may mention it."""
    bad = []
    for base in ("live_ekf_slam_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".h", ".hpp", ".cpp", ".hip")):
                    txt = open(os.path.join(dp, fn)).read()
                    if re.search(r"(import\s+oracle|from\s+oracle|#include\s*[\"<][^\">]*oracle/|libslam_oracle)", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
