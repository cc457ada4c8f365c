"""ctypes mirror of `slam_config` (include/slam_batch.h) + defaults from the reference's params.yaml.

Key names follow ekf_ws/src/base_pkg/config/params.yaml:25-52 (reference), the only config surface the hot
path reads (Filter::readCommonParams, filter.h:105-121; get_cmd, sim_node.py:209-250).
"""
import ctypes as C


class SlamConfig(C.Structure):
    _fields_ = [
        ("v_d", C.c_float), ("v_th", C.c_float),
        ("V_00", C.c_double), ("V_11", C.c_double),
        ("w_r", C.c_float), ("w_b", C.c_float),
        ("W_00", C.c_double), ("W_11", C.c_double),
        ("landmark_id_is_known", C.c_int), ("min_landmark_separation", C.c_float),
        ("d_max", C.c_double), ("th_max", C.c_double),
        ("range_max", C.c_double), ("fov_min", C.c_double), ("fov_max", C.c_double),
        ("init_x", C.c_double), ("init_y", C.c_double), ("init_yaw", C.c_double),
        ("replicate_vw_quirk", C.c_int), ("ukf_float_trig", C.c_int),
        ("reserved", C.c_int * 2),
        # quirk switches of round 5 (include/slam_batch.h): 0 = the reference's behaviour as this build reads it
        ("ekf_abs_is_int", C.c_int), ("ekf_landmark_from_x_pred", C.c_int),
        ("ukf_accumulate_zest1", C.c_int), ("ukf_sensing_yaw_from_sigma", C.c_int),
    ]

    def copy(self):
        c = SlamConfig()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(SlamConfig))
        return c


def default_config() -> SlamConfig:
    """Values committed in the reference's params.yaml (lines 19-52)."""
    c = SlamConfig()
    c.v_d, c.v_th, c.V_00, c.V_11 = 0.0, 0.0, 0.01, 0.001
    c.w_r, c.w_b, c.W_00, c.W_11 = 0.0, 0.0, 0.01, 0.01
    c.landmark_id_is_known, c.min_landmark_separation = 1, 0.1
    c.d_max, c.th_max = 0.1, 0.0546
    c.range_max, c.fov_min, c.fov_max = 3.0, -1.57, 1.57
    c.init_x, c.init_y, c.init_yaw = 0.0, 0.0, 0.0
    c.replicate_vw_quirk, c.ukf_float_trig = 1, 1
    return c


EKF_SLAM, UKF_LOC, UKF_SLAM = 1, 2, 3
F64, F32 = 0, 1
INST_NONFINITE, INST_S_SINGULAR, INST_INDEX_OOR, INST_CAPACITY, INST_SQRT_FAILED = 1, 2, 4, 8, 16
