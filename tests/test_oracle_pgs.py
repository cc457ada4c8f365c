"""CPU tests of the pose-graph oracle (oracle/slam_oracle_pgs.cpp).  GTSAM is not in the image, so the
oracle restates its published algorithm; these tests pin that restatement from independent sides."""
import numpy as np
import pytest

from live_ekf_slam_amd.config import default_config
from live_ekf_slam_amd.scenario import make_scenario


def _small_graph(oracle, T=60, L=12, seed=3, KP=8, N_max=None):
    lm, cmds = make_scenario(50 + seed, L, T)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, 1, L, KP=KP, seed=seed, cfg=cfg, want_streams=True)
    g = oracle.OraclePoseGraph(cfg, N_max=N_max or T + 1, L_max=L, KP=KP)
    g.init(0.0, 0.0, 0.0)
    nv = np.zeros(3)
    for t in range(T):
        nv = np.array([nv[0] + float(cmds[t, 0]) * np.cos(nv[2]), nv[1] + float(cmds[t, 0]) * np.sin(nv[2]), nv[2] + float(cmds[t, 1])])
        g.updateNaiveVehPoseEstimate(r["pose_init"][0, t + 1])   # the runner's naive secondary
        g.update(cmds[t, 0], cmds[t, 1], r["meas"][0, t, :min(r["cnt"][0, t], KP)])
    return g, r, lm, cmds


def test_incremental_api_equals_batch_runner(oracle):
    g, r, _, _ = _small_graph(oracle)
    st = g.solve()
    v = g.values(1)
    assert v["M"] == r["M"][0] and np.array_equal(v["ids"], r["ids"][0, :v["M"]])
    assert np.array_equal(v["poses"], r["pose_res"][0]) and np.array_equal(v["landmarks"], r["lm_res"][0, :v["M"]])
    assert st["iterations"] == r["iterations"][0] and st["err_final"] == r["err_final"][0]
    assert st["err_final"] < st["err_init"]
    assert np.array_equal(g.values(0)["poses"], r["pose_init"][0])


def test_schur_elimination_equals_dense_cholesky(oracle):
    """Poses-first block elimination vs ONE dense Cholesky of the whole damped system: same LM path, same minimiser."""
    g1, _, _, _ = _small_graph(oracle, T=80, L=15, seed=5)
    g2, _, _, _ = _small_graph(oracle, T=80, L=15, seed=5)
    s1, s2 = g1.solve(oracle.LIN_SCHUR), g2.solve(oracle.LIN_DENSE)
    assert s1["iterations"] == s2["iterations"] and s1["trials"] == s2["trials"]
    assert abs(s1["err_final"] - s2["err_final"]) < 1e-9 * max(1.0, s1["err_final"])
    assert np.abs(g1.values(1)["poses"] - g2.values(1)["poses"]).max() < 1e-8
    assert np.abs(g1.values(1)["landmarks"] - g2.values(1)["landmarks"]).max() < 1e-8


def test_minimiser_matches_scipy_least_squares(oracle):
    """Independent optimiser (scipy trust-region, numerical Jacobian) on the same whitened residuals reaches the same
    minimum.  GTSAM's default Between/Prior Jacobians drop the rotation of the residual pose (no
    GTSAM_SLOW_BUT_CORRECT_BETWEENFACTOR), so the LM fixed point sits within O(|e_theta|*|e|) of the true minimiser."""
    from scipy.optimize import least_squares
    g, _, _, _ = _small_graph(oracle, T=40, L=10, seed=2)
    st = g.solve()
    v = g.values(1)
    N, M = len(v["poses"]), v["M"]

    def fun(z):
        return g.residuals(z[:3 * N].reshape(N, 3), z[3 * N:].reshape(M, 2))

    z0 = np.concatenate([v["poses"].ravel(), v["landmarks"].ravel()])
    assert abs(0.5 * np.sum(fun(z0) ** 2) - st["err_final"]) < 1e-12 * max(1.0, st["err_final"])
    sol = least_squares(fun, z0, method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12)
    assert sol.cost <= st["err_final"] * (1 + 1e-12)
    assert st["err_final"] - sol.cost < 1e-4 * st["err_final"]          # we are at the bottom of the same valley
    assert np.abs(sol.x - z0).max() < 5e-3
    # and from the INITIAL estimate scipy needs to walk to the same place
    v0 = g.values(0)
    zi = np.concatenate([v0["poses"].ravel(), v0["landmarks"].ravel()])
    sol2 = least_squares(fun, zi, method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12)
    assert abs(sol2.cost - sol.cost) < 1e-6 * sol.cost


def test_gradient_is_derivative_of_cost_along_the_retraction(oracle):
    g, _, _, _ = _small_graph(oracle, T=30, L=8, seed=4)
    v = g.values(0)
    N, M = len(v["poses"]), v["M"]
    gp, gl = g.gradient(v["poses"], v["landmarks"])
    rng = np.random.default_rng(0)
    for _ in range(5):
        dp, dl = rng.normal(size=(N, 3)), rng.normal(size=(M, 2))
        eps = 1e-6
        pp, lp = g.retract(v["poses"], v["landmarks"], eps * dp, eps * dl)
        pm, lmm = g.retract(v["poses"], v["landmarks"], -eps * dp, -eps * dl)
        num = (0.5 * np.sum(g.residuals(pp, lp) ** 2) - 0.5 * np.sum(g.residuals(pm, lmm) ** 2)) / (2 * eps)
        ana = np.sum(gp * dp) + np.sum(gl * dl)
        assert abs(num - ana) < 2e-2 * abs(ana) + 1e-6, (num, ana)


def test_graph_building_follows_pose_graph_cpp(oracle):
    """Keys / nodes / connections as PoseGraph::update builds them (pose_graph.cpp:122-178,199-256)."""
    cfg = default_config()
    g = oracle.OraclePoseGraph(cfg, N_max=8, L_max=4, KP=2)
    g.init(1.0, 2.0, 0.5)
    g.updateNaiveVehPoseEstimate([1.1, 2.0, 0.5])
    g.update(0.1, 0.0, [[7, 2.0, 0.25], [3, 1.0, -0.5]])
    v = g.values(0)
    assert v["timestep"] == 1 and v["M"] == 2 and list(v["ids"]) == [7, 3]
    assert np.allclose(v["poses"][1], [1.1, 2.0, 0.5])
    # landmark initial estimate from the secondary filter's pose (pose_graph.cpp:162), float32 wire values
    assert np.allclose(v["landmarks"][0], [1.1 + 2.0 * np.cos(0.75), 2.0 + 2.0 * np.sin(0.75)], atol=1e-7)
    g.updateNaiveVehPoseEstimate([1.2, 2.0, 0.5])
    flags = g.update(0.1, 0.0, [[3, 1.0, -0.4], [9, 1.5, 0.0], [11, 1.0, 0.0]])   # 9 is new, 11 exceeds KP=2
    assert flags & 4
    c = g.connections()
    # first detections are recorded with landmark index -1 (getLandmarkIndexFromID returns -1 for a new id)
    assert c.tolist() == [[1, -1], [1, -1], [2, 1], [2, -1]]
    assert g.values(0)["M"] == 4      # the landmark node exists, its third factor of the step did not fit
    for _ in range(5):
        flags = g.update(0.1, 0.0, [])
    assert g.values(0)["timestep"] == 7
    assert g.update(0.1, 0.0, []) & 1   # pose capacity


def test_solve_every_iteration_mode(oracle):
    """solve_graph_every_iteration (params.yaml:64): solve after every update and adopt the result as the next initial
    estimate (pose_graph.cpp:258-264); the final estimate is as good as the one-shot solve's."""
    lm, cmds = make_scenario(60, 10, 50)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, 1, 10, KP=8, seed=9, cfg=cfg, want_streams=True)
    g = oracle.OraclePoseGraph(cfg, N_max=51, L_max=10, KP=8)
    g.init(0.0, 0.0, 0.0)
    for t in range(50):
        g.updateNaiveVehPoseEstimate(r["pose_init"][0, t + 1])
        g.update(cmds[t, 0], cmds[t, 1], r["meas"][0, t, :r["cnt"][0, t]])
        st = g.solve()
        assert st["flags"] == 0
        g.adopt()
    assert abs(g.cost(1) - r["err_final"][0]) < 1e-3 * r["err_final"][0]


def test_segmented_elimination_order_gives_the_same_solve(oracle):
    """LIN_SEG (round 5; what the GPU path does): the poses eliminated segment by segment - interiors, then the separator poses - instead
    of 0, 1, 2, ...  Any exact elimination order solves the same damped normal equations, so the LM path (iterations, lambda trials)
    is identical and the minimiser agrees to rounding; segment lengths that put a separator next to either end of the chain included."""
    from live_ekf_slam_amd.scenario import make_scenario
    for (L, T, KP, B, SL) in [(15, 80, 8, 4, 8), (20, 150, 8, 4, 32), (40, 400, 16, 4, 16), (8, 33, 8, 3, 32), (8, 34, 8, 3, 32),
                              (8, 3, 8, 3, 2), (8, 1, 8, 2, 32), (12, 40, 8, 3, 7)]:
        lm, cmds = make_scenario(100 + T, L, T)
        r0 = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=5, nthreads=4)
        r1 = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=5, nthreads=4, lin_mode=oracle.LIN_SEG | (SL << 8))
        assert np.array_equal(r0["iterations"], r1["iterations"]) and np.array_equal(r0["trials"], r1["trials"]), (L, T, SL)
        assert np.array_equal(r0["flags"], r1["flags"])
        assert np.abs(r0["pose_res"] - r1["pose_res"]).max() < 1e-9 and np.abs(r0["lm_res"] - r1["lm_res"]).max() < 1e-9, (L, T, SL)
        assert np.allclose(r0["err_final"], r1["err_final"], rtol=1e-10, atol=0)
