#!/usr/bin/env python3
"""Generate golden scenario / measurement fixtures by IMPORTING the reference simulator.

Runs ONLY in the build container (needs /root/reference); the GPU box never runs this.
What it does:
  * registers stub modules for rospy / rospkg / cv2 / cv_bridge / ROS message packages,
  * imports  ekf_ws/src/base_pkg/src/sim_node.py  from the reference tree (nothing is copied),
  * seeds the global Mersenne Twister the simulator draws from (`from random import random`),
  * runs generate_landmarks('random'), generate_full_trajectory(), then get_cmd() per command,
    emulating the float32 ROS wire format of Command / Float32MultiArray,
  * records every uniform draw consumed, and writes small .npz fixtures next to this file.

Fixture contents (data only: inputs + expected outputs):
  map[L,2] f64, cmds[T,2] f32 (wire values), cmds64[T,2] f64 (pre-wire), truth[T,3] f64,
  meas_count[T] i32, meas[T,KMAX,3] f32 (wire values: id, r, beta), meas64[T,KMAX,3] f64,
  draws_map / draws_traj / draws_step (the uniform draws, in consumption order; draws_step is
  [T, 2+2*KMAX] padded with NaN), avg_err_case (est/truth lists + compute_average_error output).
"""
import importlib.util, math, os, random, sys, types
import numpy as np
import yaml

REF = "/root/reference/ekf_ws/src/base_pkg"
HERE = os.path.dirname(os.path.abspath(__file__))


class _Msg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Capture:
    def __init__(self):
        self.msgs = []

    def publish(self, m):
        self.msgs.append(m)


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Rate:
        def __init__(self, hz):
            pass

        def sleep(self):
            pass

    mod("rospy", ROSInterruptException=Exception, Rate=_Rate, is_shutdown=lambda: False,
        logerr=print, loginfo=lambda *a, **k: None, logwarn=lambda *a, **k: None)
    mod("rospkg")
    mod("cv2")
    mod("cv_bridge", CvBridge=object)
    mod("base_pkg")
    mod("base_pkg.msg", Command=_Msg, EKFState=_Msg, UKFState=_Msg, NaiveState=_Msg, PoseGraphState=_Msg)
    mod("std_msgs")
    mod("std_msgs.msg", Float32MultiArray=_Msg)
    mod("geometry_msgs")
    mod("geometry_msgs.msg", Vector3=_Msg)
    mod("sensor_msgs")
    mod("sensor_msgs.msg", Image=_Msg)


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class _RecordingRandom:
    """Wraps random.random so every draw the simulator consumes is logged."""

    def __init__(self):
        self.log = []

    def __call__(self):
        u = random.random()
        self.log.append(u)
        return u


def f32(x):
    return float(np.float32(x))


def run_scenario(seed, L, T, map_type="random"):
    _install_stubs()
    sim = _load(REF + "/src/sim_node.py", "ref_sim_node")
    with open(REF + "/config/params.yaml") as f:
        cfg = yaml.safe_load(f)
    cfg["map"]["num_landmarks"] = L
    cfg["num_iterations"] = T
    sim.config = cfg
    sim.config_shift = cfg["map"]["occ_map_size"] / 2
    sim.config_scale = cfg["map"]["bound"] / sim.config_shift
    sim.display_region = [cfg["map"]["bound"] * cfg["plotter"]["display_region_mult"] * s for s in (-1, 1)]
    sim.occ_map = np.ones((cfg["map"]["occ_map_size"], cfg["map"]["occ_map_size"]))  # blank map: all free
    sim.dt = cfg["dt"]
    sim.lm_pub, sim.true_pose_pub, sim.true_map_pub, sim.cmd_pub = _Capture(), _Capture(), _Capture(), _Capture()
    sim.x_v = [cfg["init_pose"]["x"], cfg["init_pose"]["y"], cfg["init_pose"]["yaw"]]
    rec = _RecordingRandom()
    sim.random = rec  # the simulator did `from random import random`
    random.seed(seed)

    # --- map (sim_node.py:177-188)
    sim.generate_landmarks(map_type)
    draws_map = list(rec.log); rec.log.clear()
    L = len(sim.landmarks)          # the fixed maps (demo / grid / igvc1) set their own landmark count
    lm = np.array([sim.landmarks[i] for i in range(L)], dtype=np.float64)

    # --- trajectory (sim_node.py:63-152). The publish loop ends by itself at t == num_iterations.
    sim.generate_full_trajectory()
    draws_traj = list(rec.log); rec.log.clear()
    cmds64 = np.array([[m.fwd, m.ang] for m in sim.cmd_pub.msgs], dtype=np.float64)
    assert cmds64.shape == (T, 2)
    cmds = cmds64.astype(np.float32)  # ROS wire: Command.msg float32 fwd, ang

    # --- per-step truth + measurements (sim_node.py:209-250)
    truth = np.zeros((T, 3)); counts = np.zeros(T, dtype=np.int32)
    meas_list, draw_list = [], []
    for t in range(T):
        sim.lm_pub.msgs.clear(); rec.log.clear()
        sim.get_cmd(_Msg(fwd=float(cmds[t, 0]), ang=float(cmds[t, 1])))  # subscriber sees float32 values
        truth[t] = sim.x_v
        data = sim.lm_pub.msgs[-1].data
        counts[t] = len(data) // 3
        meas_list.append(np.array(data, dtype=np.float64).reshape(-1, 3))
        draw_list.append(list(rec.log))
    kmax = max(1, int(counts.max()))
    meas64 = np.zeros((T, kmax, 3)); draws_step = np.full((T, 2 + 2 * kmax), np.nan)
    for t in range(T):
        meas64[t, :counts[t]] = meas_list[t]
        draws_step[t, :len(draw_list[t])] = draw_list[t]
        assert len(draw_list[t]) == 2 + 2 * counts[t]
    meas = meas64.astype(np.float32)  # ROS wire: Float32MultiArray
    return dict(map=lm, cmds=cmds, cmds64=cmds64, truth=truth, meas_count=counts, meas=meas, meas64=meas64,
                draws_map=np.array(draws_map), draws_traj=np.array(draws_traj), draws_step=draws_step,
                seed=np.int64(seed), L=np.int64(L), T=np.int64(T))  # noqa


def avg_err_case():
    """compute_average_error (plotting_node.py:195-218) on a synthetic estimate list."""
    _install_stubs()
    # plotting_node imports matplotlib at module import; stub what it needs.
    for name in ["matplotlib", "matplotlib.pyplot", "matplotlib.patches", "matplotlib.lines",
                 "matplotlib.backend_bases", "matplotlib.legend_handler"]:
        m = types.ModuleType(name); sys.modules[name] = m
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.modules["matplotlib"].patches = sys.modules["matplotlib.patches"]
    sys.modules["matplotlib"].__path__ = []
    sys.modules["matplotlib.backend_bases"].MouseButton = object
    sys.modules["matplotlib.legend_handler"].HandlerPatch = object
    sys.modules["matplotlib.lines"].Line2D = object
    plt = _load(REF + "/src/plotting_node.py", "ref_plotting_node")
    rng = np.random.default_rng(7)
    T = 200
    tx, ty = np.cumsum(rng.normal(0, .1, T)), np.cumsum(rng.normal(0, .1, T))
    ex, ey = tx + rng.normal(0, .05, T), ty + rng.normal(0, .05, T)
    plt.true_poses = [_Msg(x=float(tx[i]), y=float(ty[i]), z=0.0) for i in range(T)]
    plt.avg_errs = {}
    stamps = list(range(1, T + 1))  # filter timestep t pairs with true_poses[t-1]
    plt.compute_average_error("ekf", [float(v) for v in ex], [float(v) for v in ey], stamps)
    return dict(true_x=tx, true_y=ty, est_x=ex, est_y=ey, avg_err=np.float64(plt.avg_errs["ekf"]))


if __name__ == "__main__":
    for seed, L, T in [(0, 20, 1000), (1, 20, 400), (2, 50, 1000), (1234, 50, 400)]:
        out = run_scenario(seed, L, T)
        path = os.path.join(HERE, f"sim_seed{seed}_L{L}_T{T}.npz")
        np.savez_compressed(path, **out)
        c = out["meas_count"]
        print(path, "mean k %.3f max k %d" % (c.mean(), c.max()), "bytes", os.path.getsize(path))
    for map_type in ("demo", "grid", "igvc1"):   # fixed maps: map + trajectory + measurement stream, 200 steps
        out = run_scenario(5, 0, 200, map_type)
        path = os.path.join(HERE, f"sim_{map_type}_seed5_T200.npz")
        np.savez_compressed(path, **out)
        print(path, "L", int(out["L"]), "mean k %.3f" % out["meas_count"].mean(), "bytes", os.path.getsize(path))
    np.savez_compressed(os.path.join(HERE, "avg_err_case.npz"), **avg_err_case())
    print("avg_err_case written")
