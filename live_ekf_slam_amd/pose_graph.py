"""Host-side mirror of the reference's `PoseGraph : Filter` (GTSAM implementation) for a batch of graphs.

The reference drives ONE PoseGraph through (localization_node.cpp:90-140, pose_graph.cpp)
    readParams(config) -> init(x_0, y_0, yaw_0) -> per tick: updateNaiveVehPoseEstimate(secondary state) ->
    update(cmdMsg, lmMeasMsg) [solves + publishes when timestep+1 >= num_iterations, or every tick] -> publishState()
`BatchedPoseGraph` keeps those names, the stop/solve logic of PoseGraph::update (pose_graph.cpp:199-267) and the
exception-on-error convention for B Monte-Carlo instances; everything numeric happens in libslam_hip.so behind
include/slam_pgs.h.  `NaiveFilter` is the reference's secondary filter of params.yaml:60 (filter.h:325-369).
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from .config import SlamConfig, default_config
from .filters import Command, _d, _f, _i


class NaiveFilter:
    """NaiveFilter (filter.h:325-369): propagate the commands, ignore the measurements.  Identical for every
    instance of a batch (it never sees instance-specific data), so one copy serves the whole batch."""

    def __init__(self):
        self.x_t = np.zeros(3)
        self.timestep = 0
        self.isInit = False
        self.lm_IDs = []

    def readParams(self, config=None):
        return self

    def init(self, x_0=0.0, y_0=0.0, yaw_0=0.0):
        self.timestep = 0
        self.x_t = np.array([float(np.float32(x_0)), float(np.float32(y_0)), float(np.float32(yaw_0))])
        self.isInit = True

    def update(self, cmdMsg, lmMeasMsg=None):
        fwd, ang = (cmdMsg.fwd, cmdMsg.ang) if isinstance(cmdMsg, Command) else (float(np.float32(cmdMsg[0])), float(np.float32(cmdMsg[1])))
        self.timestep += 1
        x = self.x_t
        self.x_t = np.array([x[0] + fwd * math.cos(x[2]), x[1] + fwd * math.sin(x[2]),
                             math.remainder(x[2] + ang, 2 * 3.14159265358979323846)])

    def getStateVector(self):
        return self.x_t.copy()


class BatchedPoseGraph:
    """Pose-graph SLAM (reference: class PoseGraph, filter.h:232-322, pose_graph.cpp) for `batch` instances."""

    def __init__(self, batch, num_iterations=1000, L_max=20, k_per_pose=8, device=0):
        self.batch, self.L_max, self.k_per_pose, self.device = int(batch), int(L_max), int(k_per_pose), int(device)
        self.num_iterations_total = int(num_iterations)      # config["num_iterations"], pose_graph.cpp:40
        self.solve_graph_every_iteration = False             # params.yaml:64 (the reference's file says true)
        self.cfg = default_config()
        self.h = None
        self.isInit = False
        self.solved_pose_graph = False
        self.timestep = 0

    # -- PoseGraph::readParams (pose_graph.cpp:12-66) --
    def readParams(self, config=None, solve_graph_every_iteration=None):
        L = _lib.lib()
        if isinstance(config, SlamConfig):
            self.cfg = config.copy()
        elif isinstance(config, str):
            # a params.yaml path: the common keys through the C reader, plus the three keys PoseGraph::readParams itself takes
            # from the file (pose_graph.cpp:28-47): num_iterations, pose_graph.implementation, .solve_graph_every_iteration
            _lib.check(L.slam_config_load(C.byref(self.cfg), config.encode()))
            section = None
            with open(config) as fh:
                for line in fh:
                    body = line.split("#", 1)[0].rstrip()
                    if not body.strip():
                        continue
                    if not body[0].isspace():
                        section = body.split(":", 1)[0].strip()
                    key, _, val = body.strip().partition(":")
                    val = val.strip().strip('"').strip("'")
                    if not val:
                        continue
                    if section == "num_iterations" and key == "num_iterations":
                        self.num_iterations_total = int(val)
                    elif section == "pose_graph" and key == "implementation" and val != "gtsam":   # pose_graph.cpp:31-40
                        raise _lib.SlamError("pose_graph.implementation must be gtsam (the reference's sesync/custom are incomplete)")
                    elif section == "pose_graph" and key == "solve_graph_every_iteration":
                        self.solve_graph_every_iteration = val.lower() == "true"
        elif isinstance(config, dict):
            pg = config.get("pose_graph", {})
            if pg.get("implementation", "gtsam") != "gtsam":   # pose_graph.cpp:31-40: sesync / custom throw
                raise _lib.SlamError("pose_graph.implementation must be gtsam (the reference's sesync/custom are incomplete)")
            self.solve_graph_every_iteration = bool(pg.get("solve_graph_every_iteration", self.solve_graph_every_iteration))
            self.num_iterations_total = int(config.get("num_iterations", self.num_iterations_total))
        elif config is not None:
            raise TypeError("unsupported config type")
        if solve_graph_every_iteration is not None:
            self.solve_graph_every_iteration = bool(solve_graph_every_iteration)
        if self.h is not None:
            self.close()
        h = C.c_void_p()
        _lib.check(L.pgs_create(C.byref(self.cfg), self.batch, self.num_iterations_total, self.L_max, self.k_per_pose, self.device, C.byref(h)))
        self.h = h
        return self

    def _need(self):
        if self.h is None:
            raise _lib.SlamError("readParams() must be called before using the filter")

    # -- PoseGraph::init (pose_graph.cpp:68-95) --
    def init(self, x_0=0.0, y_0=0.0, yaw_0=0.0):
        self._need()
        _lib.check(_lib.lib().pgs_init(self.h, x_0, y_0, yaw_0))
        self.isInit, self.solved_pose_graph, self.timestep = True, False, 0
        self._sec = None

    def set_stream(self, ptr):
        self._need(); _lib.check(_lib.lib().pgs_set_stream(self.h, C.c_void_p(ptr)))

    def set_seed(self, seed):
        self._need(); _lib.check(_lib.lib().pgs_set_seed(self.h, int(seed)))

    def set_instance_offset(self, first):
        self._need(); _lib.check(_lib.lib().pgs_set_instance_offset(self.h, int(first)))

    def set_map(self, map_xy):
        self._need()
        m = np.ascontiguousarray(map_xy, dtype=np.float64)
        _lib.check(_lib.lib().pgs_set_map(self.h, _d(m), m.shape[0]))

    # -- PoseGraph::updateNaiveVehPoseEstimate (pose_graph.cpp:97-119) --
    def updateNaiveVehPoseEstimate(self, state_vector, landmark_ids=None):
        """state_vector: [>=3] (one estimate for every instance, e.g. the NaiveFilter) or [B][>=3] (per instance, e.g.
        BatchedEKF.poses()); only (x, y, yaw) is used (update_landmarks_after_adding = false)."""
        sv = np.asarray(state_vector, dtype=np.float64)
        sv = np.broadcast_to(sv[:3], (self.batch, 3)) if sv.ndim == 1 else sv[:, :3]
        self._sec = np.ascontiguousarray(sv, dtype=np.float64)

    # -- PoseGraph::update (pose_graph.cpp:199-267) --
    def update(self, cmdMsg, lmMeasMsg, meas_count=None):
        self._need()
        if not self.isInit:
            raise _lib.SlamError("init() must be called before update()")
        if self.solved_pose_graph and not self.solve_graph_every_iteration:
            return                                            # :201-205
        if self.timestep + 1 >= self.num_iterations_total:    # :208-214
            self.solvePoseGraph()
            return
        cmd = np.array([cmdMsg.fwd, cmdMsg.ang], dtype=np.float32) if isinstance(cmdMsg, Command) else np.ascontiguousarray(cmdMsg, dtype=np.float32).reshape(2)
        meas = np.asarray(lmMeasMsg, dtype=np.float32)
        if meas.ndim <= 2 and meas_count is None:             # one message for all instances
            one = meas.reshape(-1, 3)
            k = one.shape[0]
            meas = np.broadcast_to(one, (self.batch, k, 3))
            meas_count = np.full(self.batch, k, dtype=np.int32)
        meas = np.ascontiguousarray(meas.reshape(self.batch, -1, 3), dtype=np.float32)
        cnt = np.ascontiguousarray(meas_count, dtype=np.int32)
        ks = meas.shape[1]
        sec = self._sec
        _lib.check(_lib.lib().pgs_update(self.h, _f(cmd), _f(meas) if ks else None, _i(cnt) if ks else None, ks,
                                         _d(sec) if sec is not None else None))
        self.timestep += 1
        if self.solve_graph_every_iteration:                  # :258-264
            self.solvePoseGraph()
            _lib.check(_lib.lib().pgs_adopt_result(self.h))

    def run_sim(self, cmds):
        """All of `cmds` with the device-side measurement generator and the NaiveFilter as the secondary filter."""
        self._need()
        c = np.ascontiguousarray(cmds, dtype=np.float32).reshape(-1, 2)
        _lib.check(_lib.lib().pgs_run_sim(self.h, _f(c), c.shape[0]))
        self.timestep += c.shape[0]

    def run_sim_every_iteration(self, cmds):
        """solve_graph_every_iteration (params.yaml:64; pose_graph.cpp:258-264) with the simulator on the device: per command one tick of
        run_sim, solvePoseGraph, initial_estimate = result.  Returns [batch][2]: LM iterations / lambda trials summed over the ticks."""
        self._need()
        c = np.ascontiguousarray(cmds, dtype=np.float32).reshape(-1, 2)
        counts = np.zeros((self.batch, 2), dtype=np.int32)
        _lib.check(_lib.lib().pgs_run_sim_every_iteration(self.h, _f(c), c.shape[0], _i(counts)))
        self.timestep += c.shape[0]
        self.solved_pose_graph = True
        return counts

    def last_iter_phases(self):
        self._need(); o = np.zeros(6); _lib.check(_lib.lib().pgs_last_iter_phases(self.h, _d(o)))
        return dict(sim_append_ms=o[0], solve_ms=o[1], adopt_ms=o[2], trials_launched=int(o[3]), syrk_flop=o[4], chol_flop=o[5])

    # -- PoseGraph::solvePoseGraph (pose_graph.cpp:269-300) --
    def solvePoseGraph(self):
        self._need()
        _lib.check(_lib.lib().pgs_solve(self.h))
        self.solved_pose_graph = True

    def set_groups(self, groups):
        """Number of concurrently solved sub-batches (separate HIP streams); 0 = automatic."""
        self._need(); _lib.check(_lib.lib().pgs_set_groups(self.h, int(groups)))

    def set_slots(self, slots):
        """Streaming solve: at most `slots` graphs of the batch in flight, the others wait for a running slot (0 = lockstep)."""
        self._need(); _lib.check(_lib.lib().pgs_set_slots(self.h, int(slots)))

    def last_solve_timeline(self):
        """Running slots per trial of the last solve, one array per solve group."""
        self._need()
        out, g, ng = [], 0, C.c_int32(1)
        while g < ng.value:
            n = C.c_int32(0)
            _lib.check(_lib.lib().pgs_last_solve_timeline(self.h, g, None, 0, C.byref(n), C.byref(ng)))
            a = np.zeros(max(n.value, 1), dtype=np.int32)
            _lib.check(_lib.lib().pgs_last_solve_timeline(self.h, g, _i(a), n.value, C.byref(n), C.byref(ng)))
            out.append(a[:n.value].copy())
            g += 1
        return out

    def adopt_result(self):
        self._need(); _lib.check(_lib.lib().pgs_adopt_result(self.h))

    def get_graph(self, instance=0, which=None):
        self._need()
        which = int(self.solved_pose_graph) if which is None else int(which)
        N = self.timestep + 1
        poses = np.zeros((N, 3)); lms = np.zeros((self.L_max, 2)); ids = np.zeros(self.L_max, dtype=np.int32)
        ts = C.c_int32(0); M = C.c_int32(0)
        _lib.check(_lib.lib().pgs_get_graph(self.h, int(instance), which, _d(poses), _d(lms), C.byref(ts), C.byref(M), _i(ids)))
        return dict(poses=poses, landmarks=lms[:M.value].copy(), timestep=ts.value, M=M.value, ids=ids[:M.value].copy())

    def connections(self, instance=0):
        self._need()
        cap = (self.timestep + 1) * self.k_per_pose
        c = np.zeros((max(cap, 1), 2), dtype=np.int32); n = C.c_int32(0)
        _lib.check(_lib.lib().pgs_get_connections(self.h, int(instance), _i(c), cap, C.byref(n)))
        return c[:n.value].copy()

    # -- PoseGraph::publishState payload (pose_graph.cpp:302-387, PoseGraphState.msg) --
    def publishState(self, instance=0):
        g = self.get_graph(instance)
        ts = g["timestep"]
        p = g["poses"][:ts]                                   # the reference's loop is `i < timestep` (:325)
        return dict(timestep=ts, M=g["M"], x_v=p[:, 0].astype(np.float32), y_v=p[:, 1].astype(np.float32),
                    yaw_v=p[:, 2].astype(np.float32), landmarks=g["landmarks"].astype(np.float32).ravel(),
                    meas_connections=self.connections(instance).ravel(),
                    topic="/state/pose_graph/result" if self.solved_pose_graph else "/state/pose_graph/initial")

    def stats(self):
        self._need()
        B = self.batch
        it = np.zeros(B, dtype=np.int32); tr = np.zeros(B, dtype=np.int32); fl = np.zeros(B, dtype=np.int32)
        e0 = np.zeros(B); e1 = np.zeros(B); lam = np.zeros(B)
        _lib.check(_lib.lib().pgs_get_stats(self.h, _i(it), _i(tr), _i(fl), _d(e0), _d(e1), _d(lam)))
        return dict(iterations=it, trials=tr, flags=fl, err_init=e0, err_final=e1, lam=lam)

    def error_stats(self, which=1):
        self._need(); out = np.zeros(self.batch); _lib.check(_lib.lib().pgs_error_stats(self.h, int(which), _d(out))); return out

    def last_solve_work(self):
        self._need(); f = C.c_double(0); t = C.c_int32(0)
        _lib.check(_lib.lib().pgs_last_solve_work(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def set_profiling(self, on=True):
        self._need(); _lib.check(_lib.lib().pgs_set_profiling(self.h, int(on)))

    def last_solve_kernel_ms(self):
        """{kernel: total ms in the last solve} from HIP events on the handle's stream (set_profiling(True))."""
        self._need(); ms = np.zeros(6); _lib.check(_lib.lib().pgs_last_solve_kernel_ms(self.h, _d(ms)))
        return dict(zip(("linearize", "chain", "syrk", "chol", "backsolve", "evaluate"), ms.tolist()))

    def last_solve_paths(self):
        """The last profiled solve by path: algorithmic SYRK FLOP and ms of the separate SYRK launches / of the fused chain + SYRK launches."""
        self._need(); o = np.zeros(8); _lib.check(_lib.lib().pgs_last_solve_paths_v2(self.h, _d(o), 8))
        return dict(flop_separate=o[0], flop_fused=o[1], ms_separate_syrk=o[2], ms_fused=o[3], flop_segmented=o[4], ms_segmented_syrk=o[5],
                    segmented=bool(o[6]), segment_length=int(o[7]))

    def sync(self):
        self._need(); _lib.check(_lib.lib().pgs_sync(self.h))

    def close(self):
        if self.h is not None:
            _lib.lib().pgs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
