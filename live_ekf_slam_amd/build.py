"""Build recipe for the HIP extension (libslam_hip.so), in-tree, gfx950 only.

`python -m live_ekf_slam_amd.build` or build_extension() from __graft_entry__.build().
hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the parity contract (bit-identical to the
CPU oracle) relies on unfused IEEE mul/add on both sides.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libslam_hip.so")
SOURCES = ["ekf_inst_103_238.hip", "ekf_inst_103_248.hip", "ekf_inst_103_424.hip", "ekf_inst_103_434.hip", "ekf_inst_103_444.hip", "ekf_inst_203_444.hip", "ekf_inst_103_444_f32.hip", "ekf_inst_103_448.hip", "ekf_inst_103_842.hip", "ekf_inst_43_148.hip", "ekf_inst_43_224.hip", "ekf_inst_43_124.hip", "ekf_inst_43_244.hip", "ekf_inst_43_244_f32.hip", "ekf_inst_43_248.hip", "ekf_inst_43_444.hip", "ekf_kernel.hip", "ukf_kernel.hip", "pgs_kernel.hip", "pgs_capi.cpp", "slam_capi.cpp"]
HEADERS = ["ukf_kernel.h", "ekf_kernel.h", "ekf_kernel_impl.h", "sim_device.h", "slam_math.h", "slam_rng.h", "pgs_kernel.h", "capi_internal.h", "../../include/slam_batch.h", "../../include/slam_pgs.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
         "-Wno-unused-function", "-x", "hip"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_extension(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for src in SOURCES:  # one hipcc per translation unit, all in parallel (the kernel variants dominate)
        obj = os.path.join(CSRC, src + ".o")
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = []
    for src, pr in procs:
        out, _ = pr.communicate()
        if verbose or pr.returncode != 0:
            sys.stderr.write(out)
        if pr.returncode != 0:
            failed.append(src)
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    build_driver()
    return LIB


DRIVER = os.path.join(HERE, "filter_driver")


def build_driver():
    """C++ host example over include/slam_filter.hpp (the reference's Filter interface), plain g++, links the .so."""
    src = os.path.join(CSRC, "host", "filter_driver.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", src, "-o", DRIVER, "-L" + HERE, "-lslam_hip",
                           "-Wl,-rpath,$ORIGIN"])
    return DRIVER


if __name__ == "__main__":
    print(build_extension(force="--force" in sys.argv, verbose="-v" in sys.argv))
