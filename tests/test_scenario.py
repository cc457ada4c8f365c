"""Host-side scenario generators vs fixtures captured from the reference simulator (sim_node.py:63-206)."""
import numpy as np

from live_ekf_slam_amd.scenario import generate_landmarks, make_scenario
import random


def test_random_map_and_tsp_trajectory_match_reference(golden_files):
    for f in golden_files:
        g = np.load(f)
        lm, cmds = make_scenario(int(g["seed"]), int(g["L"]), int(g["T"]))
        assert np.array_equal(lm, g["map"])
        assert cmds.dtype == np.float32 and np.array_equal(cmds, g["cmds"])


def test_draw_consumption_matches_reference(golden_files):
    g = np.load(golden_files[0])
    rng = random.Random(int(g["seed"]))
    generate_landmarks("random", int(g["L"]), rng)
    # after the map, the next draws must be the planner's (2 per landmark)
    nxt = [rng.random() for _ in range(2 * int(g["L"]))]
    assert np.array_equal(np.array(nxt), g["draws_traj"])


def test_command_constraints_and_grid_map():
    lm, cmds = make_scenario(3, 30, 500)
    assert np.all(cmds[:, 0] >= 0) and np.all(cmds[:, 0] <= np.float32(0.1))
    assert np.all(np.abs(cmds[:, 1]) <= np.float32(0.0546))
    d = np.linalg.norm(lm[:, None] - lm[None], axis=-1) + np.eye(30)
    assert d.min() >= 0.05 and np.abs(lm).max() <= 10.0
    grid = generate_landmarks("grid", 0, random.Random(0))
    assert grid.shape == (25, 2) and grid[0].tolist() == [-8.0, -8.0] and grid[-1].tolist() == [8.0, 8.0]


def test_fixed_and_grid_maps_match_reference():
    """map_type demo / grid / igvc1 (sim_node.py:163-199): landmark map and the TSP command sequence planned over it,
    bit for bit against fixtures captured from the imported reference simulator (tests/golden/make_golden.py)."""
    import os
    from conftest import GOLDEN
    for map_type, L in (("demo", 20), ("grid", 25), ("igvc1", 37)):
        g = np.load(os.path.join(GOLDEN, f"sim_{map_type}_seed5_T200.npz"))
        lm, cmds = make_scenario(5, 0, 200, map_type=map_type)
        assert lm.shape == (L, 2) and np.array_equal(lm, g["map"])
        assert np.array_equal(cmds, g["cmds"])


def test_cpp_generators_match_reference_fixtures(golden_files):
    """generate_landmarks / generate_full_trajectory as host C++ (include/slam_scenario.hpp, exported as
    slam_scenario_make): map and float32 command sequence bit for bit against the fixtures captured from the imported reference
    simulator, for the random maps of every golden seed and for the grid / demo / igvc1 maps.  Host code: no GPU needed."""
    import os
    from conftest import GOLDEN
    from live_ekf_slam_amd.scenario import make_scenario_native
    for f in golden_files:
        g = np.load(f)
        lm, cmds = make_scenario_native(int(g["seed"]), int(g["L"]), int(g["T"]))
        assert np.array_equal(lm, g["map"]) and np.array_equal(cmds, g["cmds"])
    for map_type, L in (("demo", 20), ("grid", 25), ("igvc1", 37)):
        g = np.load(os.path.join(GOLDEN, f"sim_{map_type}_seed5_T200.npz"))
        lm, cmds = make_scenario_native(5, 0, 200, map_type=map_type)
        assert lm.shape == (L, 2) and np.array_equal(lm, g["map"]) and np.array_equal(cmds, g["cmds"])
    lm, cmds = make_scenario_native(1234, 50, 800)       # the bench scenario
    lm2, cmds2 = make_scenario(1234, 50, 800)
    assert np.array_equal(lm, lm2) and np.array_equal(cmds, cmds2)
