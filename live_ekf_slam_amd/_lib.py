"""ctypes loader for the in-tree HIP extension libslam_hip.so (C ABI: include/slam_batch.h, include/slam_pgs.h).

There is no CPU fallback: if the library is missing or a HIP call fails, an exception is raised.
"""
import ctypes as C
import os

from .config import SlamConfig

HERE = os.path.dirname(os.path.abspath(__file__))
# SLAM_HIP_LIB: another build of the same library, for A/B tuning sessions (tools/gpu_ab.sh); the product path is the
# in-tree libslam_hip.so next to this file
LIB_PATH = os.environ.get("SLAM_HIP_LIB") or os.path.join(HERE, "libslam_hip.so")
_lib = None

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_H = C.c_void_p


class SlamError(RuntimeError):
    """Raised for any non-zero return of the C ABI (the reference signals errors by C++ exceptions)."""


# name -> (restype, argtypes); every symbol include/slam_batch.h and include/slam_pgs.h declare
SIGNATURES = {
    "slam_config_default": (C.c_int, [C.POINTER(SlamConfig)]),
    "slam_config_load": (C.c_int, [C.POINTER(SlamConfig), C.c_char_p]),
    "slam_create": (C.c_int, [C.POINTER(SlamConfig), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_H)]),
    "slam_destroy": (C.c_int, [_H]),
    "slam_set_stream": (C.c_int, [_H, C.c_void_p]),
    "slam_set_instance_offset": (C.c_int, [_H, C.c_int64]),
    "slam_set_seed": (C.c_int, [_H, C.c_uint64]),
    "slam_set_vision": (C.c_int, [_H, C.c_double, C.c_double, C.c_double]),
    "slam_init": (C.c_int, [_H, C.c_float, C.c_float, C.c_float]),
    "slam_set_map": (C.c_int, [_H, _dp, C.c_int]),
    "slam_step": (C.c_int, [_H, _fp, _fp, _ip, C.c_int]),
    "slam_step_dev": (C.c_int, [_H, _fp, C.c_void_p, C.c_void_p, C.c_int]),
    "slam_step_sim": (C.c_int, [_H, _fp]),
    "slam_set_lazy_steps": (C.c_int, [_H, C.c_int]),
    "slam_queued_steps": (C.c_int, [_H]),
    "slam_run_sim": (C.c_int, [_H, _fp, C.c_int]),
    "slam_predict": (C.c_int, [_H, _fp]),
    "slam_update_dev": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int]),
    "slam_get_state": (C.c_int, [_H, C.c_int, _dp, _dp, _ip, _ip, _ip]),
    "slam_get_sigma_points": (C.c_int, [_H, C.c_int, _dp, _ip, _ip]),
    "slam_track_instance": (C.c_int, [_H, C.c_int]),
    "slam_save_state": (C.c_int, [_H, C.c_char_p]),
    "slam_load_state": (C.c_int, [_H, C.c_char_p]),
    "slam_get_poses": (C.c_int, [_H, _dp]),
    "slam_get_landmark_counts": (C.c_int, [_H, _ip]),
    "slam_get_truth": (C.c_int, [_H, _dp]),
    "slam_get_last_meas": (C.c_int, [_H, _fp, _ip, C.c_int]),
    "slam_error_stats": (C.c_int, [_H, _dp]),
    "slam_status": (C.c_int, [_H, _ip]),
    "slam_sync": (C.c_int, [_H]),
    "slam_batch": (C.c_int, [_H]),
    "slam_state_dim_max": (C.c_int, [_H]),
    "slam_algorithmic_bytes": (C.c_int, [_H, _dp]),
    "slam_scenario_make": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint64, C.c_int, C.c_int, _dp, C.c_int, _ip, _fp]),
    "slam_set_run_chunk": (C.c_int, [_H, C.c_int]),
    "slam_set_debug_flags": (C.c_int, [_H, C.c_int]),
    "slam_variant_available": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "slam_k_histogram": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_int]),
    "slam_ukf_sweep_stats": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_int]),
    "slam_traffic_counters": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_int]),
    "slam_reset_counters_async": (C.c_int, [_H]),
    "slam_kernel_info": (C.c_int, [_H, C.c_int, C.c_char_p, C.c_int, _ip]),
    "slam_math_probe": (C.c_int, [_dp, _dp, _dp, C.c_int, C.c_int]),
    # include/slam_pgs.h
    "pgs_create": (C.c_int, [C.POINTER(SlamConfig), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_H)]),
    "pgs_destroy": (C.c_int, [_H]),
    "pgs_set_stream": (C.c_int, [_H, C.c_void_p]),
    "pgs_set_instance_offset": (C.c_int, [_H, C.c_int64]),
    "pgs_set_seed": (C.c_int, [_H, C.c_uint64]),
    "pgs_set_map": (C.c_int, [_H, _dp, C.c_int]),
    "pgs_init": (C.c_int, [_H, C.c_float, C.c_float, C.c_float]),
    "pgs_update": (C.c_int, [_H, _fp, _fp, _ip, C.c_int, _dp]),
    "pgs_update_dev": (C.c_int, [_H, _fp, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pgs_run_sim": (C.c_int, [_H, _fp, C.c_int]),
    "pgs_solve": (C.c_int, [_H]),
    "pgs_set_groups": (C.c_int, [_H, C.c_int]),
    "pgs_adopt_result": (C.c_int, [_H]),
    "pgs_get_graph": (C.c_int, [_H, C.c_int, C.c_int, _dp, _dp, _ip, _ip, _ip]),
    "pgs_get_connections": (C.c_int, [_H, C.c_int, _ip, C.c_int, _ip]),
    "pgs_get_stats": (C.c_int, [_H, _ip, _ip, _ip, _dp, _dp, _dp]),
    "pgs_error_stats": (C.c_int, [_H, C.c_int, _dp]),
    "pgs_last_solve_work": (C.c_int, [_H, _dp, _ip]),
    "pgs_set_profiling": (C.c_int, [_H, C.c_int]),
    "pgs_last_solve_kernel_ms": (C.c_int, [_H, _dp]),
    "pgs_last_solve_paths": (C.c_int, [_H, _dp]),
    "pgs_last_solve_paths_v2": (C.c_int, [_H, _dp, C.c_int]),
    "pgs_set_slots": (C.c_int, [_H, C.c_int]),
    "pgs_run_sim_every_iteration": (C.c_int, [_H, _fp, C.c_int, _ip]),
    "pgs_last_iter_phases": (C.c_int, [_H, _dp]),
    "pgs_last_solve_timeline": (C.c_int, [_H, C.c_int, _ip, C.c_int, _ip, _ip]),
    "pgs_sync": (C.c_int, [_H]),
    "pgs_timestep": (C.c_int, [_H]),
    # include/slam_multi.h
    "slam_shard_range": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "slam_multi_create": (C.c_int, [C.POINTER(SlamConfig), C.c_int, C.c_int64, C.c_int, C.c_int, _ip, C.c_int, C.POINTER(_H)]),
    "slam_multi_destroy": (C.c_int, [_H]),
    "slam_multi_devices": (C.c_int, [_H]),
    "slam_multi_batch": (C.c_int64, [_H]),
    "slam_multi_handle": (_H, [_H, C.c_int]),
    "slam_multi_shard": (C.c_int, [_H, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "slam_multi_set_seed": (C.c_int, [_H, C.c_uint64]),
    "slam_multi_set_vision": (C.c_int, [_H, C.c_double, C.c_double, C.c_double]),
    "slam_multi_set_map": (C.c_int, [_H, _dp, C.c_int]),
    "slam_multi_init": (C.c_int, [_H, C.c_float, C.c_float, C.c_float]),
    "slam_multi_step_sim": (C.c_int, [_H, _fp]),
    "slam_multi_run_sim": (C.c_int, [_H, _fp, C.c_int]),
    "slam_multi_sync": (C.c_int, [_H]),
    "slam_multi_error_stats": (C.c_int, [_H, _dp, C.c_int]),
    "slam_multi_status": (C.c_int, [_H, _ip]),
    "slam_multi_get_state": (C.c_int, [_H, C.c_int64, _dp, _dp, _ip, _ip, _ip]),
    "slam_last_error": (C.c_char_p, []),
    "slam_version": (C.c_char_p, []),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SlamError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -m live_ekf_slam_amd.build, or __graft_entry__.build()). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SlamError(f"libslam_hip error {rc}: {lib().slam_last_error().decode()}")
