"""GPU parity of the batched pose-graph solver (include/slam_pgs.h) against the CPU oracle (oracle/slam_oracle_pgs.cpp).

Graph building is integer / copy work plus one sincos per new landmark: compared BIT-EXACT.  The LM solve is an
iterative fp64 computation whose Schur complement is accumulated by MFMA in a different order than the oracle's
sequential loops: identical iteration and lambda-trial counts, 1e-9 relative on the objective, and on poses / landmarks
    max(1e-7 m, 10 x the instance's own ROUNDING SPREAD):
the largest distance between the results the ORACLE reaches with its exact elimination orders (sequential Schur / segmented
with 32, 16, 8 poses per segment; they differ in rounding only).  For all but a handful of instances in ten thousand that spread is
below 1e-9 m and the bound is the fixed 1e-7 m; an instance whose LM path is long (15+ iterations, lambda walking up and down)
amplifies one ulp in a linear solve to 1e-8 ... 1e-7 m, for ANY implementation - test_the_ill_conditioned_instance_of_the_round_4_soak
pins one (found by tools/gpu_soak_pgs.py: GPU at 2.75e-7 m with equal counts and objective)."""
import numpy as np
import pytest

from live_ekf_slam_amd.config import default_config
from live_ekf_slam_amd.scenario import make_scenario

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-7
OBJ_RTOL = 1e-9


def _compare(pg, r, B, check_init=True):
    st = pg.stats()
    assert np.array_equal(st["flags"], r["flags"])
    assert np.array_equal(st["iterations"], r["iterations"]), (st["iterations"], r["iterations"])
    assert np.array_equal(st["trials"], r["trials"]), (st["trials"], r["trials"])
    assert np.allclose(st["err_init"], r["err_init"], rtol=1e-12, atol=0)
    assert np.allclose(st["err_final"], r["err_final"], rtol=OBJ_RTOL, atol=0)
    assert np.allclose(st["lam"], r["lam"], rtol=1e-12)
    for b in range(B):
        g0, g1 = pg.get_graph(b, 0), pg.get_graph(b, 1)
        M = r["M"][b]
        assert g0["M"] == M and np.array_equal(g0["ids"], r["ids"][b, :M])
        if check_init:
            assert np.array_equal(g0["poses"], r["pose_init"][b])          # bit-exact graph building
        assert np.abs(g1["poses"] - r["pose_res"][b]).max() < POSE_TOL
        assert np.abs(g1["landmarks"] - r["lm_res"][b, :M]).max() < POSE_TOL


@pytest.mark.parametrize("L,T,B,KP", [(20, 300, 16, 8), (20, 999, 4, 8), (100, 250, 6, 24),
                                      (200, 999, 3, 32)])   # last: BASELINE configs[4] size, 1000 poses x 200 landmarks
def test_sim_build_and_solve_match_oracle(oracle, L, T, B, KP):
    import live_ekf_slam_amd as S
    lm, cmds = make_scenario(321 + L, L, T)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=11, cfg=cfg, nthreads=8)
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
    pg.set_map(lm); pg.set_seed(11); pg.init(0.0, 0.0, 0.0)
    pg.run_sim(cmds)
    pg.solvePoseGraph()
    _compare(pg, r, B)
    assert np.allclose(pg.error_stats(0), r["avg_err_init"], rtol=1e-12)
    assert np.allclose(pg.error_stats(1), r["avg_err_result"], rtol=1e-6)
    assert np.all(pg.stats()["err_final"] < pg.stats()["err_init"])


def test_update_api_with_naive_secondary_matches_oracle(oracle):
    """The reference's call sequence (localization_node.cpp:124-131): secondary filter update ->
    updateNaiveVehPoseEstimate -> update; the last update call triggers the solve (pose_graph.cpp:208-214)."""
    import live_ekf_slam_amd as S
    L, T, B, KP = 12, 80, 5, 6
    lm, cmds = make_scenario(77, L, T)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=5, cfg=cfg, want_streams=True)
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
    naive = S.NaiveFilter()
    pg.init(0.0, 0.0, 0.0); naive.init(0.0, 0.0, 0.0)
    for t in range(T):
        naive.update(cmds[t])
        pg.updateNaiveVehPoseEstimate(naive.getStateVector(), naive.lm_IDs)
        pg.update(cmds[t], r["meas"][:, t], np.minimum(r["cnt"][:, t], KP))
        assert not pg.solved_pose_graph
    pg.update(cmds[-1], np.zeros((B, 1, 3), np.float32), np.zeros(B, np.int32))   # timestep+1 >= num_iterations: solve
    assert pg.solved_pose_graph and pg.timestep == T
    # the host NaiveFilter uses libm cos/sin, the oracle runner the shared deterministic ones: 1-ulp differences
    _compare(pg, r, B, check_init=False)
    assert np.abs(pg.get_graph(0, 0)["poses"] - r["pose_init"][0]).max() < 1e-12
    msg = pg.publishState(0)
    assert msg["topic"].endswith("result") and len(msg["x_v"]) == T and msg["x_v"].dtype == np.float32
    g = oracle.OraclePoseGraph(cfg, N_max=T + 1, L_max=L, KP=KP)
    g.init(0.0, 0.0, 0.0)
    for t in range(T):
        g.updateNaiveVehPoseEstimate(r["pose_init"][0, t + 1])
        g.update(cmds[t, 0], cmds[t, 1], r["meas"][0, t, :min(r["cnt"][0, t], KP)])
    assert np.array_equal(msg["meas_connections"].reshape(-1, 2), g.connections())


def test_solve_every_iteration_mode(oracle):
    import live_ekf_slam_amd as S
    L, T, B, KP = 10, 40, 3, 6
    lm, cmds = make_scenario(60, L, T)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=9, cfg=cfg, want_streams=True)
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg, solve_graph_every_iteration=True)
    pg.init(0.0, 0.0, 0.0)
    gs = []
    for b in range(B):
        g = oracle.OraclePoseGraph(cfg, N_max=T + 1, L_max=L, KP=KP); g.init(0.0, 0.0, 0.0); gs.append(g)
    for t in range(T):
        pg.updateNaiveVehPoseEstimate(r["pose_init"][:, t + 1])
        pg.update(cmds[t], r["meas"][:, t], np.minimum(r["cnt"][:, t], KP))
        for b, g in enumerate(gs):
            g.updateNaiveVehPoseEstimate(r["pose_init"][b, t + 1])
            g.update(cmds[t, 0], cmds[t, 1], r["meas"][b, t, :min(r["cnt"][b, t], KP)])
            st = g.solve(); g.adopt()
    st = pg.stats()
    for b, g in enumerate(gs):
        v = g.values(1)
        assert np.abs(pg.get_graph(b, 1)["poses"] - v["poses"]).max() < 1e-6
        assert np.abs(pg.get_graph(b, 1)["landmarks"] - v["landmarks"]).max() < 1e-6
        assert st["flags"][b] == 0


@pytest.mark.parametrize("async_ticks", [1, 0])
@pytest.mark.parametrize("L,T,B,KP", [(10, 60, 6, 6), (20, 300, 5, 8), (60, 200, 3, 16)])
def test_solve_every_iteration_on_the_device_matches_oracle(monkeypatch, oracle, L, T, B, KP, async_ticks):
    """pgs_run_sim_every_iteration: the reference's default mode (params.yaml:64; pose_graph.cpp:258-264) with the simulator on the device -
    per tick one step of the simulator + NaiveFilter + append, the solve, `initial_estimate = result`.  Against the oracle run in the same
    mode: the LM iteration and lambda-trial counts SUMMED over all ticks are equal, the final result within the one-shot bar.
    async_ticks = 1 (default): no batch-wide barrier per tick - every graph walks through its ticks at its own pace, one LM trial of all
    unfinished graphs per round of launches, converged graphs advanced on a second stream; 0: the lockstep tick loop."""
    import live_ekf_slam_amd as S
    monkeypatch.setenv("SLAM_PGS_ITER_ASYNC", str(async_ticks))
    lm, cmds = make_scenario(61 + L, L, T)
    cfg = default_config()
    for lin in (oracle.LIN_SCHUR, oracle.LIN_SEG):
        r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=9, cfg=cfg, every_iteration=True, lin_mode=lin, nthreads=8)
        if lin == oracle.LIN_SCHUR:
            r0 = r
    assert np.array_equal(r0["iterations"], r["iterations"]) and np.array_equal(r0["trials"], r["trials"])   # the oracle's two orders agree
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg, solve_graph_every_iteration=True)
    pg.set_map(lm); pg.set_seed(9); pg.init(0.0, 0.0, 0.0)
    counts = pg.run_sim_every_iteration(cmds)
    assert pg.timestep == T and pg.solved_pose_graph
    assert np.array_equal(counts[:, 0], r["iterations"]), (counts[:, 0], r["iterations"])
    assert np.array_equal(counts[:, 1], r["trials"]), (counts[:, 1], r["trials"])
    assert np.all(counts[:, 1] >= T)          # at least one lambda trial per tick
    st = pg.stats()
    assert np.all(st["flags"] == 0)
    for b in range(B):
        g1 = pg.get_graph(b, 1)
        M = r["M"][b]
        assert g1["M"] == M and np.array_equal(g1["ids"], r["ids"][b, :M])
        assert np.abs(g1["poses"] - r["pose_res"][b]).max() < POSE_TOL
        assert M == 0 or np.abs(g1["landmarks"] - r["lm_res"][b, :M]).max() < POSE_TOL
        assert np.abs(pg.get_graph(b, 0)["poses"] - r["pose_res"][b]).max() < POSE_TOL     # adopted: initial_estimate = result
    assert np.allclose(pg.error_stats(1), r["avg_err_result"], rtol=1e-6)
    ph = pg.last_iter_phases()
    assert ph["trials_launched"] >= T and ph["chol_flop"] >= 0 and ph["syrk_flop"] > 0
    pg.close()


def test_capacity_flags_and_errors():
    import live_ekf_slam_amd as S
    pg = S.BatchedPoseGraph(2, num_iterations=4, L_max=2, k_per_pose=2).readParams()
    with pytest.raises(S.SlamError):
        pg.update([0.1, 0.0], [])          # init first
    pg.init(0.0, 0.0, 0.0)
    pg.updateNaiveVehPoseEstimate([0.1, 0.0, 0.0])
    pg.update([0.1, 0.0], [[1, 1.0, 0.1], [2, 1.0, 0.2], [3, 1.0, 0.3]])   # third landmark exceeds L_max = 2
    assert np.all(pg.stats()["flags"] & 2)
    pg.update([0.1, 0.0], [[1, 1.0, 0.1], [2, 1.0, 0.2], [1, 1.1, 0.1]])   # third detection exceeds k_per_pose = 2
    assert np.all(pg.stats()["flags"] & 4)
    pg.update([0.1, 0.0], [])
    assert not pg.solved_pose_graph
    pg.update([0.1, 0.0], [])              # timestep+1 >= num_iterations -> solve instead of growing
    assert pg.solved_pose_graph and pg.timestep == 3
    g = pg.get_graph(0, 1)
    assert g["M"] == 2 and np.all(np.isfinite(g["poses"]))
    cfg = default_config(); cfg.landmark_id_is_known = 0
    with pytest.raises(S.SlamError):       # pose_graph.cpp:137
        S.BatchedPoseGraph(1, 10, 2).readParams(cfg)


def test_shard_invariance_and_determinism():
    """Instances are keyed by GLOBAL index (pgs_set_instance_offset): a batch of 8 == two shards of 4, bit for bit."""
    import live_ekf_slam_amd as S
    L, T, KP = 20, 200, 8
    lm, cmds = make_scenario(5, L, T)

    def run(B, off):
        pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams()
        pg.set_map(lm); pg.set_seed(3); pg.set_instance_offset(off); pg.init(0.0, 0.0, 0.0)
        pg.run_sim(cmds); pg.solvePoseGraph()
        out = [pg.get_graph(b, 1) for b in range(B)], pg.stats(), pg.error_stats(1)
        pg.close()
        return out

    whole, a, b = run(8, 0), run(4, 0), run(4, 4)
    again = run(8, 0)
    for i in range(8):
        part = a if i < 4 else b
        for key in ("poses", "landmarks", "ids"):
            assert np.array_equal(whole[0][i][key], part[0][i % 4][key])
            assert np.array_equal(whole[0][i][key], again[0][i][key])
    assert np.array_equal(whole[2], np.concatenate([a[2], b[2]]))
    assert np.array_equal(whole[1]["trials"], np.concatenate([a[1]["trials"], b[1]["trials"]]))


def test_degenerate_graphs(oracle):
    """Prior only (one pose), odometry only (no landmark ever seen), and a single landmark: same answers as the oracle."""
    import live_ekf_slam_amd as S
    cfg = default_config()
    for T, meas_at in ((0, {}), (5, {}), (6, {2: [[4, 2.0, 0.3]], 4: [[4, 1.9, 0.25]]})):
        pg = S.BatchedPoseGraph(2, num_iterations=T + 2, L_max=3, k_per_pose=2).readParams(cfg)
        g = oracle.OraclePoseGraph(cfg, N_max=T + 2, L_max=3, KP=2)
        naive = S.NaiveFilter()
        pg.init(0.5, -0.25, 0.1); g.init(0.5, -0.25, 0.1); naive.init(0.5, -0.25, 0.1)
        for t in range(T):
            cmd = [0.1, 0.02 * (t % 3 - 1)]
            naive.update(cmd)
            sv = naive.getStateVector() + np.array([0.01 * t, -0.005 * t, 0.002 * t])   # a drifting secondary estimate
            pg.updateNaiveVehPoseEstimate(sv); g.updateNaiveVehPoseEstimate(sv)
            m = np.asarray(meas_at.get(t, []), dtype=np.float32).reshape(-1, 3)
            pg.update(cmd, m); g.update(cmd[0], cmd[1], m)
        pg.solvePoseGraph(); so = g.solve()
        st = pg.stats()
        assert st["flags"].tolist() == [so["flags"]] * 2 and st["iterations"].tolist() == [so["iterations"]] * 2
        v = g.values(1)
        for b in range(2):
            gg = pg.get_graph(b, 1)
            assert gg["M"] == v["M"] and np.abs(gg["poses"] - v["poses"]).max() < 1e-9
            if v["M"]:
                assert np.abs(gg["landmarks"] - v["landmarks"]).max() < 1e-9
        pg.close()


def test_update_dev_equals_update():
    """pgs_update_dev (device buffers, e.g. another engine's measurement dump) == pgs_update (host buffers)."""
    import ctypes as C
    import live_ekf_slam_amd as S
    from live_ekf_slam_amd import _lib
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    B, T, L, KP = 3, 25, 6, 4
    rng = np.random.default_rng(4)
    a = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(); a.init(0.0, 0.0, 0.0)
    b = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(); b.init(0.0, 0.0, 0.0)
    d_meas, d_cnt, d_sec = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_meas), B * KP * 3 * 4) == 0 and hip.hipMalloc(C.byref(d_cnt), B * 4) == 0 and hip.hipMalloc(C.byref(d_sec), B * 3 * 8) == 0
    pose = np.zeros((B, 3))
    for t in range(T):
        cmd = np.array([0.1, 0.03 * np.sin(0.3 * t)], dtype=np.float32)
        pose = pose + np.array([0.1 * np.cos(pose[:, 2]), 0.1 * np.sin(pose[:, 2]), np.full(B, float(cmd[1]))]).T + rng.normal(0, 1e-3, (B, 3))
        cnt = rng.integers(0, KP + 1, B).astype(np.int32)
        meas = np.zeros((B, KP, 3), dtype=np.float32)
        meas[:, :, 0] = rng.integers(0, L, (B, KP)); meas[:, :, 1] = rng.uniform(1, 3, (B, KP)); meas[:, :, 2] = rng.uniform(-1, 1, (B, KP))
        a.updateNaiveVehPoseEstimate(pose); a.update(cmd, meas, cnt)
        sec = np.ascontiguousarray(pose)
        for dst, src in ((d_meas, meas), (d_cnt, cnt), (d_sec, sec)):
            assert hip.hipMemcpy(dst, src.ctypes.data_as(C.c_void_p), src.nbytes, 1) == 0
        _lib.check(_lib.lib().pgs_update_dev(b.h, cmd.ctypes.data_as(C.POINTER(C.c_float)), d_meas, d_cnt, KP, d_sec))
        b.timestep += 1
    a.solvePoseGraph(); b.solvePoseGraph()
    for i in range(B):
        for which in (0, 1):
            ga, gb = a.get_graph(i, which), b.get_graph(i, which)
            assert ga["M"] == gb["M"] and np.array_equal(ga["ids"], gb["ids"])
            assert np.array_equal(ga["poses"], gb["poses"]) and np.array_equal(ga["landmarks"], gb["landmarks"])
        assert np.array_equal(a.connections(i), b.connections(i))
    a.close(); b.close()


def test_solve_groups_do_not_change_results():
    """pgs_set_groups only changes which stream / host thread drives an instance's LM loop."""
    import live_ekf_slam_amd as S
    L, T, KP, B = 20, 150, 8, 10
    lm, cmds = make_scenario(9, L, T)
    out = []
    for G in (1, 3, 16):
        pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams()
        pg.set_groups(G)
        pg.set_map(lm); pg.set_seed(4); pg.init(0.0, 0.0, 0.0)
        pg.run_sim(cmds); pg.solvePoseGraph()
        out.append(([pg.get_graph(b, 1) for b in range(B)], pg.stats()))
        pg.close()
    for graphs, st in out[1:]:
        for b in range(B):
            assert np.array_equal(graphs[b]["poses"], out[0][0][b]["poses"])
            assert np.array_equal(graphs[b]["landmarks"], out[0][0][b]["landmarks"])
        for key in ("iterations", "trials", "flags", "err_final", "lam"):
            assert np.array_equal(st[key], out[0][1][key])


@pytest.mark.parametrize("range_max,expect_sl", [(3.0, 32), (5.4, 16), (5.8, 8), (6.1, 0)])
def test_segment_length_is_chosen_per_solve(oracle, range_max, expect_sl):
    """a wide sensor on a dense map (200 landmarks, field of view +-3 rad) makes the interior poses of a 32-pose segment see
    more than the 63 landmarks the segment kernels hold - until round 5 the whole solve then fell back to the sequential chain.  Now the solve
    takes the longest segment length of 32 / 16 / 8 that fits (the segment-count-dependent arrays are re-made on demand); only a sensor that
    sees more than 63 landmarks from 8 poses still takes the sequential chain.  Whatever the order, the LM path equals the oracle's, the result
    agrees with the SAME-order oracle (LIN_SEG with that segment length) to 1e-9 m and with the sequential oracle within the usual bar."""
    import live_ekf_slam_amd as S
    L, T, B, KP = 200, 300, 3, 64
    lm, cmds = make_scenario(700, L, T)
    cfg = default_config(); cfg.range_max = range_max; cfg.fov_min = -3.0; cfg.fov_max = 3.0
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=3, cfg=cfg, nthreads=8)
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
    pg.set_map(lm); pg.set_seed(3); pg.init(0.0, 0.0, 0.0)
    pg.run_sim(cmds)
    pg.set_profiling(True)      # (last_solve_paths reports the order the solve ran)
    pg.solvePoseGraph()
    paths = pg.last_solve_paths()
    assert paths["segmented"] == (expect_sl > 0) and (expect_sl == 0 or paths["segment_length"] == expect_sl), paths
    _compare(pg, r, B)
    if expect_sl:
        rs = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=3, cfg=cfg, nthreads=8, lin_mode=oracle.LIN_SEG | (expect_sl << 8))
        for b in range(B):
            assert np.abs(pg.get_graph(b, 1)["poses"] - rs["pose_res"][b]).max() < 1e-9
    # a second solve of the same handle starts from the length the first one found
    pg.set_profiling(False)
    pg.solvePoseGraph()
    _compare(pg, r, B)
    pg.close()


@pytest.mark.parametrize("KP", [64, 100])
def test_messages_of_more_than_64_detections_on_the_device_simulator(oracle, KP):
    """Found by tools/gpu_soak_pgs.py `wide` (round 6): the device simulator cut a message at 64 detections BEFORE the graph append, so the
    landmarks of the detections beyond were not created at that pose (the reference's loop creates the landmark, then drops the factor that
    does not fit: pose_graph.cpp:249-256) - same counts, same flags, landmarks metres apart.  Now every detection reaches the append."""
    import live_ekf_slam_amd as S
    L, T, B = 200, 196, 2
    lm, cmds = make_scenario(1050712621, L, T)
    cfg = default_config(); cfg.range_max = 6.413; cfg.fov_min = -3.068; cfg.fov_max = 3.068
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=602889364, cfg=cfg, nthreads=4, want_streams=True)
    assert r["cnt"].max() > 64 and (KP == 100 or np.all(r["flags"] & 4))
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
    pg.set_map(lm); pg.set_seed(602889364); pg.init(0.0, 0.0, 0.0)
    pg.run_sim(cmds); pg.solvePoseGraph()
    _compare(pg, r, B)
    pg.close()


@pytest.mark.parametrize("G", [1, 2])
def test_streaming_slots_do_not_change_results(oracle, G):
    """pgs_set_slots (round 6): only `slots` graphs are in flight, the others wait and take over the running slots of converged ones
    on the device; trials are enqueued ahead of the host.  A graph's LM sequence depends on nothing but the graph: poses, landmarks,
    iteration / trial counts, objective and lambda are BIT-identical to the lockstep solve for every slot count and pipeline depth,
    and equal to the oracle's within the usual bar.  The timeline shows the refill: full lists while graphs wait."""
    import live_ekf_slam_amd as S
    L, T, KP, B = 20, 150, 8, 45
    lm, cmds = make_scenario(9, L, T)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=4, cfg=cfg, nthreads=8)
    out = []
    for slots in (0, 8, 7, 44, 64):
        pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
        pg.set_groups(G); pg.set_slots(slots)
        pg.set_map(lm); pg.set_seed(4); pg.init(0.0, 0.0, 0.0)
        pg.run_sim(cmds); pg.solvePoseGraph()
        if slots == 0:
            _compare(pg, r, B)
        out.append(([pg.get_graph(b, 1) for b in range(B)], pg.stats(), pg.last_solve_timeline()))
        # a second solve of the same handle re-uses the counters and cursors
        pg.solvePoseGraph()
        st2 = pg.stats()
        for key in ("iterations", "trials", "flags", "err_final", "lam"):
            assert np.array_equal(st2[key], out[-1][1][key]), (slots, key)
        pg.close()
    for (graphs, st, tl), slots in zip(out[1:], (8, 7, 44, 64)):
        for b in range(B):
            assert np.array_equal(graphs[b]["poses"], out[0][0][b]["poses"]), (slots, b)
            assert np.array_equal(graphs[b]["landmarks"], out[0][0][b]["landmarks"]), (slots, b)
        for key in ("iterations", "trials", "flags", "err_final", "lam"):
            assert np.array_equal(st[key], out[0][1][key]), (slots, key)
        assert len(tl) == G
        cap = -(-slots // G)
        per = -(-B // G)
        if cap < per:      # streaming: never more than the group's share in flight, and the list stays full while graphs wait
            for g, a in enumerate(tl):
                assert a.max() <= max(cap, 1) * 4 and a[0] == cap, (slots, g, a)   # (x lanes once the lockstep tail runs its lambda lanes)
                total = int(st["trials"][g * per:(g + 1) * per].sum())
                assert int(a.sum()) >= total, (slots, g)          # every consumed trial ran in some slot (speculative lanes add more)
                full = int((a[:max(1, len(a) // 3)] == cap).sum())
                assert full >= max(1, len(a) // 3) - 1, (slots, g, a)


def test_streaming_trial_cap_is_per_graph(monkeypatch, oracle):
    """Found by tools/gpu_soak_pgs.py big (round 6): with few running slots a batch needs many more LAUNCHES than any graph needs trials;
    the host's cap on launches (SLAM_PGS_MAX_TRIALS, default 400) cut the queue off and left the graphs at its end unsolved.  The cap is a
    graph's own trial count now (pgs_decide_kernel applies it while graphs stream)."""
    import live_ekf_slam_amd as S
    monkeypatch.setenv("SLAM_PGS_MAX_TRIALS", "8")
    L, T, KP, B = 20, 150, 8, 45
    lm, cmds = make_scenario(9, L, T)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=4, cfg=cfg, nthreads=8)
    assert r["trials"].max() < 8 and r["trials"].sum() > 2 * 6 * 8    # no graph at the cap, the batch through 2 slots far beyond it
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
    pg.set_groups(1); pg.set_slots(2)
    pg.set_map(lm); pg.set_seed(4); pg.init(0.0, 0.0, 0.0)
    pg.run_sim(cmds); pg.solvePoseGraph()
    _compare(pg, r, B)
    assert len(pg.last_solve_timeline()[0]) > 6 * 8
    pg.close()


@pytest.mark.parametrize("L,T", [(20, 150), (60, 400)])
def test_syrk_variants_agree_with_the_oracle(monkeypatch, oracle, L, T):
    """The Schur complement has tile kernels (few active instances) and one with instance-resident accumulators and Y
    streamed through LDS (many); the Cholesky has a 1024- and a 256-thread build.  Forced in combinations
    (SLAM_PGS_SYRK_TILE, SLAM_PGS_CHOL_THREADS), all reproduce the oracle's LM path (same iteration and trial counts,
    _compare) and its result within the usual tolerance."""
    import live_ekf_slam_amd as S
    KP, B = 8, 12
    lm, cmds = make_scenario(21, L, T)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=6, cfg=cfg, nthreads=8)
    # SYRK: 32x32 tiles / 64x64 tiles / instance-resident accumulators; Cholesky: 1024 or 256 threads per instance (the
    # 256-thread one is otherwise chosen only above 256 active instances, i.e. by no other test)
    monkeypatch.setenv("SLAM_PGS_SEG", "0")     # the sequential chain of rounds 1-4 (the default is the segmented elimination, tested below)
    monkeypatch.setenv("SLAM_PGS_FUSED", "0")   # chain and SYRK as two launches (few slots would otherwise run the fused kernel)
    for tile, chol in (("32", "1024"), ("1", "1024"), ("64", "256"), ("1", "256")):
        monkeypatch.setenv("SLAM_PGS_SYRK_TILE", tile)
        monkeypatch.setenv("SLAM_PGS_CHOL_THREADS", chol)
        pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
        pg.set_map(lm); pg.set_seed(6); pg.init(0.0, 0.0, 0.0)
        pg.run_sim(cmds); pg.solvePoseGraph()
        _compare(pg, r, B)
        pg.close()


@pytest.mark.parametrize("L,T,KP,B", [(20, 150, 8, 12), (100, 250, 24, 6), (200, 999, 32, 5)])
def test_fused_chain_syrk_and_slot_list_agree_with_the_oracle(monkeypatch, oracle, L, T, KP, B):
    """Round 3: the trial kernels are launched over the compacted list of running slots (SLAM_PGS_LIST, default on) and
    chain + SYRK run as ONE launch with Y in LDS on 2, 3 or 4 workgroups per instance (SLAM_PGS_FUSED = 2 | 3 | 4 forces a
    variant, 0 = two launches, default = chosen per trial from the number of running slots).  Every combination follows the
    oracle's LM path (same iteration / trial counts) to the usual tolerance; the last size is BASELINE configs[4]."""
    import live_ekf_slam_amd as S
    lm, cmds = make_scenario(55 + L, L, T)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=8, cfg=cfg, nthreads=8)
    monkeypatch.setenv("SLAM_PGS_SEG", "0")     # these are the launch shapes of the sequential chain
    res = {}
    for fused, lst in (("0", "1"), ("0", "0"), ("2", "1"), ("3", "1"), ("4", "1"), ("4", "0"), ("-1", "1")):
        monkeypatch.setenv("SLAM_PGS_FUSED", fused)
        monkeypatch.setenv("SLAM_PGS_LIST", lst)
        pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
        pg.set_map(lm); pg.set_seed(8); pg.init(0.0, 0.0, 0.0)
        pg.run_sim(cmds); pg.solvePoseGraph()
        _compare(pg, r, B)
        res[(fused, lst)] = [pg.get_graph(b, 1)["poses"] for b in range(B)]
        pg.close()
    for b in range(B):   # the slot list only changes which workgroup serves which instance; the fused variants share their arithmetic
        assert np.array_equal(res[("0", "1")][b], res[("0", "0")][b])
        assert np.array_equal(res[("4", "1")][b], res[("4", "0")][b])
        assert np.array_equal(res[("2", "1")][b], res[("4", "1")][b]) and np.array_equal(res[("3", "1")][b], res[("4", "1")][b])


@pytest.mark.parametrize("L,T,KP,B", [(20, 150, 8, 12), (8, 31, 8, 5), (8, 32, 8, 5), (8, 33, 8, 5), (8, 34, 8, 5), (8, 65, 8, 5), (8, 16, 8, 3), (8, 7, 8, 3), (100, 250, 24, 6),
                                      (60, 400, 16, 9), (200, 999, 32, 5)])
def test_segmented_elimination_agrees_with_the_oracle(monkeypatch, oracle, L, T, KP, B):
    """Round 5: the pose chain is eliminated segment by segment (pgs_seg_impl.h: interiors of the segments side by side, then the
    separator poses, then the landmarks) instead of pose by pose - the reference's solve (pose_graph.cpp:273-300) in another exact
    elimination order.  (i) Against the oracle's SEQUENTIAL order: the usual bar, identical LM iteration / trial counts and 1e-7 m.
    (ii) Against the oracle restating the SAME order (LIN_SEG): 1e-9 m - what is left is the MFMA's fused accumulation and the
    reciprocal-square-root pivots.  Segment lengths 32 (default), 16 and 7, with and without the slot list; the pose counts around 33 / 34 /
    66 put the last separator next to the end of the chain; the last size is BASELINE configs[4]."""
    import live_ekf_slam_amd as S
    lm, cmds = make_scenario(155 + L, L, T)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=8, cfg=cfg, nthreads=8)
    for sl, lst in ((32, "1"), (16, "1"), (7, "0")):
        if L == 200 and sl == 7:
            continue   # 142 separators: beyond the separator kernel's staging (the host would fall back to the sequential chain)
        monkeypatch.setenv("SLAM_PGS_SEG", str(sl))
        monkeypatch.setenv("SLAM_PGS_LIST", lst)
        monkeypatch.setenv("SLAM_PGS_SEG_BACK_GLOBAL", "1" if sl == 16 else "0")   # the pose step's chains from global memory / out of LDS
        pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
        pg.set_map(lm); pg.set_seed(8); pg.init(0.0, 0.0, 0.0)
        pg.run_sim(cmds); pg.solvePoseGraph()
        paths = pg.last_solve_paths()
        assert paths["segmented"] and paths["segment_length"] == sl, paths   # the path under test really ran
        _compare(pg, r, B)
        rs = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=8, cfg=cfg, nthreads=8, lin_mode=oracle.LIN_SEG | (sl << 8))
        assert np.array_equal(rs["iterations"], r["iterations"]) and np.array_equal(rs["trials"], r["trials"])
        for b in range(B):
            g1 = pg.get_graph(b, 1); M = r["M"][b]
            assert np.abs(g1["poses"] - rs["pose_res"][b]).max() < 1e-9
            assert np.abs(g1["landmarks"] - rs["lm_res"][b, :M]).max() < 1e-9
        pg.close()


def test_segmented_elimination_falls_back_when_a_segment_sees_too_many_landmarks(monkeypatch, oracle):
    """A map of 64 landmarks all in view all the time (64 factor slots per pose, the device simulator's message limit): every segment's
    column set would hold 64 landmarks (> 63, the segment kernels' limit), so the solve runs the sequential chain - and says so."""
    import live_ekf_slam_amd as S
    L, T, KP, B = 64, 70, 64, 3
    rng = np.random.default_rng(5)
    lm = rng.uniform(-1.0, 1.0, (L, 2)) + np.array([1.5, 0.0])
    _, cmds = make_scenario(9, 20, T)
    cfg = default_config(); cfg.range_max = 50.0; cfg.fov_min = -3.2; cfg.fov_max = 3.2
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=3, cfg=cfg, nthreads=4)
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
    pg.set_map(lm); pg.set_seed(3); pg.init(0.0, 0.0, 0.0)
    pg.run_sim(cmds); pg.solvePoseGraph()
    assert not pg.last_solve_paths()["segmented"]
    _compare(pg, r, B)
    pg.close()


def _rounding_spread(oracle, lm, cmds, B, L, KP, seed, cfg, r):
    """Per instance: the largest distance between the oracle's sequential-Schur result `r` and its results with the segmented
    elimination at 32, 16 and 8 poses per segment - the same exact solve, rounded differently (module docstring)."""
    sp = np.zeros(B)
    for sl in (32, 16, 8):
        v = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=seed, cfg=cfg, nthreads=8, lin_mode=oracle.LIN_SEG | (sl << 8))
        assert np.array_equal(v["trials"], r["trials"]) and np.array_equal(v["iterations"], r["iterations"])
        for b in range(B):
            sp[b] = max(sp[b], np.abs(v["pose_res"][b] - r["pose_res"][b]).max(), np.abs(v["lm_res"][b] - r["lm_res"][b]).max())
    return sp


@pytest.mark.parametrize("seg", ["32", "0"])
def test_the_ill_conditioned_instance_of_the_round_4_soak(monkeypatch, oracle, seg):
    """profiles/r04b/soak_final: L=40 T=846 KP=32 B=11 seed=1058182634 scenario=1043562854 fused=3 list=1 groups=3 lanes=4 - instance 9
    (19 iterations, 36 lambda trials) came out 2.75e-7 m from the oracle with identical counts, flags and objective: beyond the fixed
    1e-7 m.  Its LM path amplifies rounding: the oracle's OWN elimination orders end up to 7e-8 m apart on it (2.5e-9 m Schur vs dense,
    7.2e-8 m Schur vs 8-pose segments) while the other ten instances agree to 1e-10 m.  The bar of this module - max(1e-7 m, 10 x that
    spread) - is what the instance is held to, on the launch shape that reported it (seg = 0: the sequential chain, fused on three
    workgroups) and on the default segmented path."""
    import live_ekf_slam_amd as S
    L, T, KP, B, seed, sc = 40, 846, 32, 11, 1058182634, 1043562854
    lm, cmds = make_scenario(sc, L, T)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=seed, cfg=cfg, nthreads=8)
    spread = _rounding_spread(oracle, lm, cmds, B, L, KP, seed, cfg, r)
    assert spread[9] > 1e-8 and np.all(np.delete(spread, 9) < 1e-8), spread     # the premise: ONE instance is that sensitive
    monkeypatch.setenv("SLAM_PGS_SEG", seg)
    monkeypatch.setenv("SLAM_PGS_FUSED", "3"); monkeypatch.setenv("SLAM_PGS_LIST", "1"); monkeypatch.setenv("SLAM_PGS_LANES", "4")
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
    pg.set_groups(3)
    pg.set_map(lm); pg.set_seed(seed); pg.init(0.0, 0.0, 0.0)
    pg.run_sim(cmds); pg.solvePoseGraph()
    st = pg.stats()
    assert np.array_equal(st["flags"], r["flags"]) and np.array_equal(st["iterations"], r["iterations"]) and np.array_equal(st["trials"], r["trials"])
    assert np.allclose(st["err_final"], r["err_final"], rtol=OBJ_RTOL, atol=0)
    for b in range(B):
        g1 = pg.get_graph(b, 1); M = r["M"][b]
        err = max(np.abs(g1["poses"] - r["pose_res"][b]).max(), np.abs(g1["landmarks"] - r["lm_res"][b, :M]).max())
        assert err < max(POSE_TOL, 10.0 * spread[b]), (b, err, spread[b])
    pg.close()


@pytest.mark.parametrize("L,T,KP,B,seed,sc,seg,groups", [(200, 278, 4, 10, 898228925, 220903737, "32", 0), (200, 359, 8, 14, 41288564, 1035020105, "16", 0),
                                                         (150, 274, 4, 8, 111986297, 1010396802, "5", 3)])
def test_segmented_elimination_on_graphs_that_drop_detections(monkeypatch, oracle, L, T, KP, B, seed, sc, seg, groups):
    """Found by tools/gpu_soak_pgs.py (round 5): with 4 - 8 factor slots per pose on a dense map most detections are dropped
    (PGS_FLAG_MEAS_CAP) while their landmarks are still created, so a landmark's first FACTOR can come long after those of landmarks
    numbered after it - the per-landmark first separator was not monotone in the landmark index, which the tile SYRK's k trimming assumes
    (steps wrong by metres, LM paths of 50 - 100 iterations).  The plan now takes the suffix minimum."""
    import live_ekf_slam_amd as S
    lm, cmds = make_scenario(sc, L, T)
    cfg = default_config()
    r = oracle.run_pgs_batch(lm, cmds, B, L, KP=KP, seed=seed, cfg=cfg, nthreads=8)
    assert np.all(r["flags"] & 4)     # the premise: every instance dropped detections
    monkeypatch.setenv("SLAM_PGS_SEG", seg)
    pg = S.BatchedPoseGraph(B, num_iterations=T + 1, L_max=L, k_per_pose=KP).readParams(cfg)
    if groups:
        pg.set_groups(groups)
    pg.set_map(lm); pg.set_seed(seed); pg.init(0.0, 0.0, 0.0)
    pg.run_sim(cmds); pg.solvePoseGraph()
    assert pg.last_solve_paths()["segmented"]
    _compare(pg, r, B)
    pg.close()
