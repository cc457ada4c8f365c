"""Host-side mirror of the reference's `Filter` interface for the batched engine.

The reference drives ONE filter object through
    readParams(config) -> init(x_0, y_0, yaw_0) -> update(cmdMsg, lmMeasMsg) -> publishState()/getStateVector()
(ekf_ws/src/localization_pkg/include/localization_pkg/filter.h:54-77; caller: localization_node.cpp:33-47,
90-106,108-140).  `BatchedEKF` keeps those names, argument meanings and the exception-on-error convention, for a
batch of B Monte-Carlo instances that share map + commands; everything numeric happens in libslam_hip.so.
The C++ twin of this class (for a C++/ROS host) is include/slam_filter.hpp.
"""
import ctypes as C
import numpy as np

from . import _lib
from .config import SlamConfig, default_config, EKF_SLAM, UKF_LOC, UKF_SLAM, F64


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Command:
    """base_pkg/Command (Command.msg:3-5): float32 fwd, ang."""

    def __init__(self, fwd=0.0, ang=0.0):
        self.fwd = float(np.float32(fwd))
        self.ang = float(np.float32(ang))


class BatchedFilter:
    """Common part of the batched filters (reference: class Filter, filter.h:54-145)."""

    kind = None

    def __init__(self, batch, L_max, device=0, dtype=F64):
        self.batch, self.L_max, self.device, self.dtype = int(batch), int(L_max), int(device), dtype
        self.cfg = default_config()
        self.h = None
        self.isInit = False  # filter.h:68
        self.timestep = 0

    # -- Filter::readParams(YAML::Node) (filter.h:59, readCommonParams filter.h:105-121) --
    def readParams(self, config=None):
        """config: None (reference defaults), a SlamConfig, a params.yaml path, or a nested dict with the
        reference's YAML structure."""
        L = _lib.lib()
        if config is None:
            pass
        elif isinstance(config, SlamConfig):
            self.cfg = config.copy()
        elif isinstance(config, str):
            _lib.check(L.slam_config_load(C.byref(self.cfg), config.encode()))
        elif isinstance(config, dict):
            c = self.cfg
            pn, sn, cons = config.get("process_noise", {}), config.get("sensing_noise", {}), config.get("constraints", {})
            c.v_d = pn.get("mean", {}).get("v_d", c.v_d); c.v_th = pn.get("mean", {}).get("v_th", c.v_th)
            c.V_00 = pn.get("cov", {}).get("V_00", c.V_00); c.V_11 = pn.get("cov", {}).get("V_11", c.V_11)
            c.w_r = sn.get("mean", {}).get("w_r", c.w_r); c.w_b = sn.get("mean", {}).get("w_b", c.w_b)
            c.W_00 = sn.get("cov", {}).get("W_00", c.W_00); c.W_11 = sn.get("cov", {}).get("W_11", c.W_11)
            m = cons.get("measurements", {})
            c.landmark_id_is_known = int(m.get("landmark_id_is_known", c.landmark_id_is_known))
            c.min_landmark_separation = m.get("min_landmark_separation", c.min_landmark_separation)
            cm, v = cons.get("commands", {}), cons.get("vision", {})
            c.d_max = cm.get("d_max", c.d_max); c.th_max = cm.get("th_max", c.th_max)
            c.range_max = v.get("range_max", c.range_max); c.fov_min = v.get("fov_min", c.fov_min); c.fov_max = v.get("fov_max", c.fov_max)
            ip = config.get("init_pose", {})
            c.init_x = ip.get("x", c.init_x); c.init_y = ip.get("y", c.init_y); c.init_yaw = ip.get("yaw", c.init_yaw)
        else:
            raise TypeError("unsupported config type")
        if self.h is not None:
            self.close()
        h = C.c_void_p()
        _lib.check(L.slam_create(C.byref(self.cfg), self.kind, self.batch, self.L_max, self.dtype, self.device, C.byref(h)))
        self.h = h
        self.n_max = L.slam_state_dim_max(self.h)
        return self

    def _need(self):
        if self.h is None:
            raise _lib.SlamError("readParams() must be called before using the filter")

    # -- Filter::init(float x_0, float y_0, float yaw_0) (filter.h:60) --
    def init(self, x_0=0.0, y_0=0.0, yaw_0=0.0):
        self._need()
        _lib.check(_lib.lib().slam_init(self.h, x_0, y_0, yaw_0))
        self.isInit = True
        self.timestep = 0

    # -- batch plumbing that has no counterpart in the single-instance reference --
    def set_stream(self, hip_stream_ptr):
        self._need(); _lib.check(_lib.lib().slam_set_stream(self.h, C.c_void_p(hip_stream_ptr)))

    def set_seed(self, seed):
        self._need(); _lib.check(_lib.lib().slam_set_seed(self.h, int(seed)))

    def set_instance_offset(self, first):
        self._need(); _lib.check(_lib.lib().slam_set_instance_offset(self.h, int(first)))

    def set_vision(self, range_max, fov_min, fov_max):
        self._need(); _lib.check(_lib.lib().slam_set_vision(self.h, range_max, fov_min, fov_max))

    def set_map(self, map_xy):
        """True landmark map [L][2]; `filter->map` of localization_node.cpp:152-156 / sim landmarks."""
        self._need()
        m = np.ascontiguousarray(map_xy, dtype=np.float64)
        _lib.check(_lib.lib().slam_set_map(self.h, _d(m), m.shape[0]))

    # -- Filter::update(Command, Float32MultiArray) (filter.h:61) for every instance --
    def update(self, cmdMsg, lmMeasMsg, meas_count=None):
        """cmdMsg: Command or (fwd, ang).  lmMeasMsg: float32 array [B][k][3] of (id, range, bearing) padded to a
        common k, with meas_count[B] valid detections per instance; or a flat [3k] list (the reference's
        Float32MultiArray.data), which is then applied to EVERY instance."""
        self._need()
        if not self.isInit:
            raise _lib.SlamError("init() must be called before update()")  # localization_node.cpp:109
        cmd = self._cmd(cmdMsg)
        meas = np.asarray(lmMeasMsg, dtype=np.float32)
        if meas.ndim <= 2 and meas_count is None:  # one message for all instances
            one = meas.reshape(-1, 3)
            k = one.shape[0]
            meas = np.broadcast_to(one, (self.batch, k, 3)) if k else np.zeros((self.batch, 1, 3), np.float32)
            meas_count = np.full(self.batch, k, dtype=np.int32)
        meas = np.ascontiguousarray(meas.reshape(self.batch, -1, 3), dtype=np.float32)
        if meas.shape[1] == 0:
            meas = np.zeros((self.batch, 1, 3), np.float32)
        cnt = np.ascontiguousarray(meas_count, dtype=np.int32)
        _lib.check(_lib.lib().slam_step(self.h, _f(cmd), _f(meas), _i(cnt), meas.shape[1]))
        self.timestep += 1

    def update_dev(self, cmdMsg, d_meas_ptr, d_count_ptr, k_stride):
        self._need()
        cmd = self._cmd(cmdMsg)
        _lib.check(_lib.lib().slam_step_dev(self.h, _f(cmd), C.c_void_p(d_meas_ptr), C.c_void_p(d_count_ptr), k_stride))
        self.timestep += 1

    def update_sim(self, cmdMsg):
        """One step with the device-side generator (get_cmd, sim_node.py:209-250) feeding the filter."""
        self._need()
        cmd = self._cmd(cmdMsg)
        _lib.check(_lib.lib().slam_step_sim(self.h, _f(cmd)))
        self.timestep += 1

    def run_sim(self, cmds):
        self._need()
        c = np.ascontiguousarray(cmds, dtype=np.float32).reshape(-1, 2)
        _lib.check(_lib.lib().slam_run_sim(self.h, _f(c), c.shape[0]))
        self.timestep += c.shape[0]

    @staticmethod
    def _cmd(cmdMsg):
        if isinstance(cmdMsg, Command):
            return np.array([cmdMsg.fwd, cmdMsg.ang], dtype=np.float32)
        return np.ascontiguousarray(cmdMsg, dtype=np.float32).reshape(2)

    # -- Filter::getStateVector() (filter.h:76; ekf.cpp:181-184) --
    def getStateVector(self, instance=0):
        return self.get_state(instance)["x"]

    def get_state(self, instance=0):
        self._need()
        x = np.zeros(self.n_max); P = np.zeros(self.n_max * self.n_max); ids = np.zeros(self.L_max, dtype=np.int32)
        M = C.c_int32(0); ts = C.c_int32(0)
        _lib.check(_lib.lib().slam_get_state(self.h, int(instance), _d(x), _d(P), C.byref(M), _i(ids), C.byref(ts)))
        n = self._n(M.value)
        return dict(x=x[:n].copy(), P=P[:n * n].reshape(n, n).copy(), M=M.value, ids=ids[:M.value].copy(), timestep=ts.value)

    def track_instance(self, instance):
        """publishState(instance) every tick without running the batch's queued steps (slam_track_instance); -1 = off."""
        self._need(); _lib.check(_lib.lib().slam_track_instance(self.h, int(instance)))

    def save_state(self, path):
        """Checkpoint of the whole batch (slam_save_state)."""
        self._need(); _lib.check(_lib.lib().slam_save_state(self.h, str(path).encode()))

    def load_state(self, path):
        """Resume from a checkpoint written by a handle of the same kind / batch / L_max / dtype (slam_load_state)."""
        self._need(); _lib.check(_lib.lib().slam_load_state(self.h, str(path).encode()))
        self.isInit = True

    def poses(self):
        self._need(); out = np.zeros((self.batch, 3)); _lib.check(_lib.lib().slam_get_poses(self.h, _d(out))); return out

    def landmark_counts(self):
        self._need(); out = np.zeros(self.batch, dtype=np.int32); _lib.check(_lib.lib().slam_get_landmark_counts(self.h, _i(out))); return out

    def truth(self):
        self._need(); out = np.zeros((self.batch, 3)); _lib.check(_lib.lib().slam_get_truth(self.h, _d(out))); return out

    def status(self):
        self._need(); out = np.zeros(self.batch, dtype=np.int32); _lib.check(_lib.lib().slam_status(self.h, _i(out))); return out

    def error_stats(self):
        """Per-instance average position error (compute_average_error, plotting_node.py:195-218)."""
        self._need(); out = np.zeros(self.batch); _lib.check(_lib.lib().slam_error_stats(self.h, _d(out))); return out

    def last_meas(self, k_stride):
        self._need()
        meas = np.zeros((self.batch, k_stride, 3), dtype=np.float32); cnt = np.zeros(self.batch, dtype=np.int32)
        _lib.check(_lib.lib().slam_get_last_meas(self.h, _f(meas), _i(cnt), k_stride))
        return meas, cnt

    def algorithmic_bytes(self):
        self._need(); v = C.c_double(0); _lib.check(_lib.lib().slam_algorithmic_bytes(self.h, C.byref(v))); return v.value

    def set_run_chunk(self, steps_per_launch):
        """Timesteps per kernel launch of run_sim (0 = the whole call)."""
        self._need(); _lib.check(_lib.lib().slam_set_run_chunk(self.h, int(steps_per_launch)))

    def set_debug_flags(self, flags):
        self._need(); _lib.check(_lib.lib().slam_set_debug_flags(self.h, int(flags)))

    def step_stamps(self, steps):
        """(wall-clock ticks at 100 MHz, detections) of every workgroup at the end of each of the first `steps` (<= 128)
        timesteps of the LAST multi-step launch (needs set_debug_flags(32) before that launch): two [batch][steps] arrays."""
        self._need()
        L = _lib.lib()
        L.slam_debug_read_prof_raw.argtypes = [C.c_void_p, C.c_void_p]
        L.slam_debug_read_prof_raw.restype = C.c_int
        buf = np.zeros((self.batch, 128), dtype=np.uint64)
        _lib.check(L.slam_debug_read_prof_raw(self.h, buf.ctypes.data_as(C.c_void_p)))
        return (buf[:, :steps] >> np.uint64(4)).astype(np.int64), (buf[:, :steps] & np.uint64(15)).astype(np.int64)

    def k_histogram(self, reset=False):
        """Instance-steps by detections per message (k = 0..6, >= 7) since creation / the last reset (EKF and UKF step kernels)."""
        self._need(); out = np.zeros(8, dtype=np.uint64)
        _lib.check(_lib.lib().slam_k_histogram(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64)), int(bool(reset))))
        return out

    def sweep_stats(self, reset=False):
        """UKF: (Jacobi sweeps that rotated something, eigen-decompositions) since creation / the last reset."""
        self._need(); out = np.zeros(2, dtype=np.uint64)
        _lib.check(_lib.lib().slam_ukf_sweep_stats(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64)), int(bool(reset))))
        return out

    def reset_counters_async(self):
        """Zero the detection-count histogram and the traffic counters in stream order (no host synchronisation)."""
        _lib.check(_lib.lib().slam_reset_counters_async(self.h))

    def traffic_counters(self, reset=False):
        """EKF: device-counted (P-stream bytes read + written by passes, other global bytes, passes, updates applied by passes)
        since creation / the last reset (slam_traffic_counters)."""
        self._need(); out = np.zeros(4, dtype=np.uint64)
        _lib.check(_lib.lib().slam_traffic_counters(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64)), int(bool(reset))))
        return out

    def kernel_info(self, multi_step=True):
        """EKF: the step-kernel instantiation this handle launches and what the runtime reports about it."""
        self._need(); name = C.create_string_buffer(128); out = np.zeros(5, dtype=np.int32)
        _lib.check(_lib.lib().slam_kernel_info(self.h, int(bool(multi_step)), name, 128, _i(out)))
        return dict(name=name.value.decode(), lds_bytes=int(out[0]), vgprs=int(out[1]), threads=int(out[2]),
                    workgroups_per_cu=int(out[3]), cus=int(out[4]))

    def set_lazy_steps(self, n):
        self._need(); _lib.check(_lib.lib().slam_set_lazy_steps(self.h, int(n)))

    def sync(self):
        self._need(); _lib.check(_lib.lib().slam_sync(self.h))

    def close(self):
        if self.h is not None:
            _lib.lib().slam_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchedEKF(BatchedFilter):
    """EKF-SLAM (reference: class EKF, filter.h:148-174, ekf.cpp)."""

    kind = EKF_SLAM

    def _n(self, M):
        return 3 + 2 * M

    # -- EKF::publishState payload (ekf.cpp:192-220, EKFState.msg) --
    def publishState(self, instance=0):
        s = self.get_state(instance)
        x, M = s["x"], s["M"]
        lm = np.empty(3 * M, dtype=np.float32)  # [id, x, y] triplets (ekf.cpp:203-208)
        lm[0::3] = s["ids"]; lm[1::3] = x[3::2]; lm[2::3] = x[4::2]
        return dict(timestep=s["timestep"], x_v=np.float32(x[0]), y_v=np.float32(x[1]), yaw_v=np.float32(x[2]),
                    M=M, landmarks=lm, P=s["P"].astype(np.float32).ravel())  # P row-major (ekf.cpp:211-217)


class BatchedUKF(BatchedFilter):
    """UKF-SLAM (reference: class UKF, filter.h:177-223, ukf.cpp).  State x = [x, y, cos(yaw), sin(yaw), landmarks]."""

    kind = UKF_SLAM

    def _n(self, M):
        return 4 + 2 * M

    # -- UKF::getStateVector (ukf.cpp:47-53; the reference's fixed-size Vector3d bug is not replicated) --
    def getStateVector(self, instance=0):
        x = self.get_state(instance)["x"]
        return np.concatenate([[x[0], x[1], np.remainder(np.arctan2(x[3], x[2]) + np.pi, 2 * np.pi) - np.pi], x[4:]])

    # -- UKF::predictionStage / UKF::updateStage (filter.h:187-188, ukf.cpp:197-291) --
    def predictionStage(self, cmdMsg):
        self._need()
        _lib.check(_lib.lib().slam_predict(self.h, _f(self._cmd(cmdMsg))))

    def updateStage(self, d_meas_ptr=None, d_count_ptr=None, k_stride=0):
        """Measurements as DEVICE pointers ([B][k_stride][3] float32, [B] int32); none = empty message."""
        self._need()
        _lib.check(_lib.lib().slam_update_dev(self.h, C.c_void_p(d_meas_ptr), C.c_void_p(d_count_ptr), int(k_stride)))
        self.timestep += 1

    def sigma_points(self, instance=0):
        """X of the last prediction stage, shape (n, 2n+1) (ukf.cpp:214-219)."""
        self._need()
        r = C.c_int32(0); c = C.c_int32(0)
        X = np.zeros(self.n_max * (2 * self.n_max + 1))
        _lib.check(_lib.lib().slam_get_sigma_points(self.h, int(instance), _d(X), C.byref(r), C.byref(c)))
        return X[:r.value * c.value].reshape(c.value, r.value).T.copy()

    # -- UKF::publishState payload (ukf.cpp:60-104, UKFState.msg); X column by column as the reference pushes it --
    def publishState(self, instance=0):
        import math
        s = self.get_state(instance)
        x, M = s["x"], s["M"]
        lm = np.empty(3 * M, dtype=np.float32)
        lm[0::3] = s["ids"]; lm[1::3] = x[4::2]; lm[2::3] = x[5::2]
        yaw = math.remainder(math.atan2(x[3], x[2]), 2 * 3.14159265358979323846)
        return dict(timestep=s["timestep"], x_v=np.float32(x[0]), y_v=np.float32(x[1]), yaw_v=np.float32(yaw),
                    M=M, landmarks=lm, P=s["P"].astype(np.float32).ravel(),
                    X=self.sigma_points(instance).T.astype(np.float32).ravel())


class BatchedUKFLoc(BatchedUKF):
    """UKF localisation against the known map (FilterChoice::UKF_LOC, localization_node.cpp:39-41, ukf.cpp:146-154):
    the state is the vehicle only; every detection updates against `set_map`'s landmark of the same id."""

    kind = UKF_LOC

    def __init__(self, batch, device=0):
        super().__init__(batch, 1, device)
