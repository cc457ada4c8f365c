"""Host-side scenario generators: landmark maps and the precomputed command trajectory.

Reference behaviour (ekf_ws/src/base_pkg/src/sim_node.py): generate_landmarks (lines 155-206) and
generate_full_trajectory (lines 63-152, greedy nearest-neighbour tour with command clamps).  The reference is
Python, so this side is Python too; it consumes a `random.Random(seed)` stream in the same draw order as the
reference consumes the global `random.random` (map draws, then 2 draws per landmark for the planner's noisy
map), so a seed reproduces the reference's scenario exactly (tests/test_scenario.py, golden fixtures).
Only the blank occupancy map is supported (every cell free), i.e. no rejection by obstacles.
"""
import math
import random

import numpy as np

# params.yaml:68-72, 88-91, 25-28, 15 (reference defaults)
DEFAULTS = dict(bound=10.0, min_landmark_separation=0.05, grid_step=4, landmark_noise=0.2,
                visitation_threshold=3.0, d_max=0.1, th_max=0.0546, display_region_mult=1.0)


def _dist(a, b):
    return ((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2) ** (1 / 2)


def generate_landmarks(map_type, num_landmarks, rng, bound=DEFAULTS["bound"],
                       min_sep=DEFAULTS["min_landmark_separation"], grid_step=DEFAULTS["grid_step"]):
    """Landmark map as float64 array [L][2]; id = row.  map_type: 'random' | 'grid' | 'demo' | 'igvc1'
    (sim_node.py:163-199; the two fixed maps are data files, `num_landmarks` is ignored for them and for 'grid')."""
    if map_type in ("demo", "igvc1"):
        import json, os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "fixed_maps.json")) as fh:
            return np.array(json.load(fh)[map_type], dtype=np.float64)
    if map_type in ("random", "rand"):
        pts = []
        while len(pts) < num_landmarks:
            cand = (2 * bound * rng.random() - bound, 2 * bound * rng.random() - bound)
            if any(_dist(q, cand) < min_sep for q in pts):
                continue
            pts.append(cand)
        return np.array(pts, dtype=np.float64)
    if map_type == "grid":
        half = grid_step / 2
        ticks = np.arange(-bound + half, bound, grid_step)
        return np.array([(float(r), float(c)) for r in ticks for c in ticks], dtype=np.float64)
    raise ValueError(f"unsupported map_type {map_type!r} (random | grid | demo | igvc1)")


def generate_full_trajectory(landmarks, num_iterations, rng, x0=(0.0, 0.0, 0.0), **kw):
    """Commands [T][2] float64 (fwd, ang) of the nearest-neighbour tour (sim_node.py:63-138).

    The caller rounds to float32 for the Command wire format.  The pose is integrated WITHOUT noise."""
    o = dict(DEFAULTS); o.update(kw)
    L = len(landmarks)
    lo, hi = -o["bound"] * o["display_region_mult"] + 1, o["bound"] * o["display_region_mult"] - 1
    rough = []
    for i in range(L):  # planner's noisy copy of the map: 2 draws per landmark, x then y
        nx = landmarks[i][0] + 2 * o["landmark_noise"] * rng.random() - o["landmark_noise"]
        ny = landmarks[i][1] + 2 * o["landmark_noise"] * rng.random() - o["landmark_noise"]
        rough.append((max(lo, min(nx, hi)), max(lo, min(ny, hi))))
    pose = list(x0)
    start = 0
    best = _dist(rough[0], pose)
    for i in range(L):
        if _dist(rough[i], pose) < best:
            start, best = i, _dist(rough[i], pose)
    tour, todo, cur = [start], [i for i in range(L) if i != start], start
    while todo:
        nxt, nd = None, -1.0
        for i in todo:
            dd = _dist(rough[i], rough[cur])
            if nd < 0 or dd < nd:
                nxt, nd = i, dd
        tour.append(nxt); todo.remove(nxt); cur = nxt
    cmds = np.zeros((num_iterations, 2), dtype=np.float64)
    for t in range(num_iterations):
        if _dist(pose, rough[tour[0]]) < o["visitation_threshold"]:
            tour = tour[1:] + [tour[0]]
        goal = rough[tour[0]]
        d = min(_dist(goal, pose), o["d_max"])
        hdg = math.remainder(math.atan2(goal[1] - pose[1], goal[0] - pose[0]) - pose[2], math.tau)
        if abs(hdg) > o["th_max"]:
            hdg = o["th_max"] * float(np.sign(hdg))
        pose = [pose[0] + d * math.cos(pose[2]), pose[1] + d * math.sin(pose[2]), pose[2] + hdg]
        cmds[t] = (d, hdg)
    return cmds


def make_scenario(seed, num_landmarks, num_iterations, map_type="random", **kw):
    """Map + float32 command sequence for one scenario seed (the reference's draw order: map, then planner)."""
    rng = random.Random(seed)
    lm = generate_landmarks(map_type, num_landmarks, rng,
                            **{k: v for k, v in kw.items() if k in ("bound", "min_sep", "grid_step")})
    cmds = generate_full_trajectory(lm, num_iterations, rng,
                                    **{k: v for k, v in kw.items() if k not in ("min_sep", "grid_step")})
    return lm, cmds.astype(np.float32)


def make_scenario_native(seed, num_landmarks, num_iterations, map_type="random"):
    """The same scenario from the C++ generators of the host library (include/slam_scenario.hpp through the C ABI
    slam_scenario_make): what a C++ host uses; bit-identical to make_scenario (tests/test_scenario.py)."""
    import ctypes as C
    import os
    from . import _lib
    L = _lib.lib()
    fixed = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "fixed_maps.json").encode()
    n = C.c_int32(0)
    _lib.check(L.slam_scenario_make(map_type.encode(), fixed, int(seed), int(num_landmarks), 0, None, 0, C.byref(n), None))
    lm = np.zeros((n.value, 2)); cmds = np.zeros((int(num_iterations), 2), dtype=np.float32)
    _lib.check(L.slam_scenario_make(map_type.encode(), fixed, int(seed), int(num_landmarks), int(num_iterations),
                                    lm.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n),
                                    cmds.ctypes.data_as(C.POINTER(C.c_float))))
    return lm, cmds
