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
SOURCES = ["ekf_kernel.hip", "ekf_big_kernel.hip", "ukf_kernel.hip", "ukf_big_kernel.hip", "pgs_kernel.hip", "pgs_capi.cpp", "slam_capi.cpp", "scenario_capi.cpp", "multi_capi.cpp"]
HEADERS = ["ukf_kernel.h", "ekf_kernel.h", "ekf_kernel_impl.h", "ekf_step_prestep.h", "ekf_step_control.h", "ekf_step_stream.h", "ekf_step_decoupled.h", "ekf_step_lockstep.h", "ekf_inst.hip", "sim_device.h", "slam_math.h", "slam_rng.h", "pgs_kernel.h", "pgs_seg_impl.h", "pgs_factors.h", "pgs_graph.h", "pgs_linearize.h", "pgs_chain.h", "pgs_syrk.h", "pgs_chol.h",
           "pgs_backsolve.h", "pgs_lm_control.h",
           "capi_internal.h", "jacobi_schedule.h", "lds_attr.h", "../../include/slam_batch.h", "../../include/slam_pgs.h", "../../include/slam_scenario.hpp",
           "../../include/slam_filter.hpp", "../../include/slam_multi.h", "host/filter_driver.cpp", "host/config_parse.h", "host/stream_parse.h"]
# Instantiations of the EKF step kernel: (NMAX, W, KG, UNR, f32 storage, PIPE[, KP]); variant code = KP*10000 + PIPE*1000 + W*100 +
# KG*10 + UNR (ekf_kernel.h; KP = 0 / absent: derived from KG).  The release library holds the defaults only (they must match
# SLAM_DEF_* in ekf_kernel.hip) plus the one-wavefront lockstep-only variant the tests force; the sweep set is for tuning
# sessions: SLAM_SWEEP=1 python -m live_ekf_slam_amd.build --force
EKF_DEFAULT_VARIANTS = [(43, 2, 5, 4, 0, 1), (43, 2, 4, 4, 0, 1), (103, 4, 6, 4, 0, 1), (103, 4, 5, 4, 0, 1), (103, 4, 4, 4, 0, 1),
                        (203, 4, 5, 4, 0, 1), (203, 4, 4, 4, 0, 1), (403, 4, 4, 4, 0, 1),
                        (43, 2, 4, 4, 1, 1), (103, 4, 6, 2, 1, 1), (103, 4, 4, 2, 1, 1),
                        (43, 1, 2, 4, 0, 1)]   # one wavefront per filter: no decoupled loop, every step through the synchronised path
EKF_SWEEP_VARIANTS = [(103, 3, 6, 4, 0, 1), (103, 3, 5, 4, 0, 1), (103, 3, 4, 4, 0, 1, 2), (103, 3, 5, 4, 0, 1, 2), (103, 2, 3, 4, 0, 1, 2), (103, 2, 4, 4, 0, 1, 2), (103, 3, 6, 4, 0, 1, 2),
                      (103, 4, 4, 4, 0, 1, 2), (103, 4, 5, 4, 0, 1, 2), (103, 3, 4, 4, 0, 1, 3), (103, 4, 6, 2, 0, 1), (43, 2, 6, 4, 0, 1),
                      (203, 4, 6, 4, 0, 1), (103, 4, 5, 2, 0, 1), (103, 2, 4, 4, 0, 1), (103, 3, 4, 4, 0, 1), (103, 4, 3, 4, 0, 1),
                      (43, 2, 4, 2, 0, 1), (43, 4, 4, 4, 0, 1), (43, 2, 4, 8, 0, 1), (43, 2, 2, 4, 0, 1)]
# EKF_FLAGS, -disable-machine-licm: with the register budget these kernels run at (128 VGPRs at four waves per SIMD), hoisting the
# materialisation of fp64 constants out of loops pins registers the loops need; the compiler then SPILLED the hoisted constants
# (det_atan's hi / lo table) to scratch and every atan2 of the EKF chain waited for three scratch loads.  Without the pass the
# multi-step EKF kernel spills 8 VGPRs instead of 39 (f64 0.925 -> 0.907 ms/step, fp32 0.861 -> 0.821, L=20 202 -> 214 M steps/s).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
         "-Wno-unused-function", "-x", "hip"]
EKF_FLAGS = ["-mllvm", "-disable-machine-licm"]   # the EKF translation units only (pose graph: -2 %, UKF: +-1 %)


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
    extra = ["-DSLAM_ABLATE"] if os.environ.get("SLAM_ABLATE") else []   # timing experiments only (WRONG results)
    extra += os.environ.get("SLAM_EXTRA_FLAGS", "").split()               # A/B tuning builds (tools/gpu_ab.sh)
    jobs = [(src, os.path.join(CSRC, src + ".o"), []) for src in SOURCES]
    # per-file code generation options: the UKF step kernel keeps its MFMA accumulators in VGPRs (the compiler's default put them
    # in AGPRs and copied all of them in and out around every k-block of the covariance contraction)
    per_file = {"ukf_kernel.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}   # (ukf_big_kernel.hip has no MFMA)
    variants = list(EKF_DEFAULT_VARIANTS) + (EKF_SWEEP_VARIANTS if os.environ.get("SLAM_SWEEP") else [])
    # A/B builds: SLAM_EXTRA_VARIANTS="103,4,6,4,1,1;103,4,6,2,1,1" adds instantiations to the defaults (NMAX,W,KG,UNR,f32,PIPE[,KP])
    variants += [tuple(int(x) for x in v.split(",")) for v in os.environ.get("SLAM_EXTRA_VARIANTS", "").split(";") if v.strip()]
    variants = list(dict.fromkeys(variants))
    for v in variants:
        nmax, w, kg, unr, f32, pipe = v[:6]
        kp = v[6] if len(v) > 6 else 0
        tag = f"ekf_inst_{nmax}_{kp if kp else ''}{pipe}{w}{kg}{unr}{'_f32' if f32 else ''}"
        jobs.append((tag, os.path.join(CSRC, tag + ".o"),
                     [f"-DV_NMAX={nmax}", f"-DV_W={w}", f"-DV_KG={kg}", f"-DV_UNR={unr}", f"-DV_F32={f32}", f"-DV_PIPE={pipe}", f"-DV_KP={kp}"]))
    for f in os.listdir(CSRC):   # objects of variants that are no longer part of the build
        if f.startswith("ekf_inst_") and f.endswith(".o") and os.path.join(CSRC, f) not in [j[1] for j in jobs]:
            os.remove(os.path.join(CSRC, f))
    objs, failed = [], []
    maxpar = int(os.environ.get("SLAM_BUILD_JOBS", "8"))   # one hipcc per translation unit, in parallel
    pending, running = list(jobs), []
    while pending or running:
        while pending and len(running) < maxpar:
            name, obj, defs = pending.pop(0)
            src = os.path.join(CSRC, name if not defs else "ekf_inst.hip")
            cmd = [hipcc] + FLAGS + (EKF_FLAGS if (defs or name in ("ekf_kernel.hip", "ekf_big_kernel.hip")) else []) + extra + per_file.get(name, []) + defs + ["-c", src, "-o", obj]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                print(" ".join(cmd), flush=True)
            running.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            objs.append(obj)
        name, pr = running.pop(0)
        out, _ = pr.communicate()
        if verbose or pr.returncode != 0:
            sys.stderr.write(out)
        if pr.returncode != 0:
            failed.append(name)
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
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
