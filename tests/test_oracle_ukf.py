"""CPU tests of the UKF oracle: Appendix-E known answers, an INDEPENDENT numpy transliteration of ukf.cpp (LAPACK
eigh for the matrix square root, numpy matrix products), and invariants."""
import math

import numpy as np
import pytest

from conftest import load_golden

F32 = np.float32
TWO_PI = 2 * 3.14159265358979323846


def _rem(x):
    return math.remainder(x, TWO_PI)


class NumpyUKF:
    """Straight transliteration of ukf.cpp:3-45,106-372 with numpy (float casts of SURVEY.md Appendix B, float
    overload of cos/sin).  Deliberately shares nothing with the C++ oracle."""

    def __init__(self, V00=0.01, V11=0.01, W00=1.0, W11=1.0, v_d=0.0, v_th=0.0, w_r=0.0, w_b=0.0):
        self.V = np.diag([V00, V11]); self.W = np.diag([W00, W11])   # V/W quirk on: V = (W_00, W_11), W = I
        self.v_d, self.v_th, self.w_r, self.w_b = F32(v_d), F32(v_th), F32(w_r), F32(w_b)
        self.W0 = F32(0.2)
        self.M = 0; self.ids = []
        self.P = np.diag([1e-4, 1e-4, 2.5e-5, 2.5e-5])

    def init(self, x0, y0, yaw0):
        yaw0 = F32(yaw0)
        self.x = np.array([F32(x0), F32(y0), F32(math.cos(yaw0)), F32(math.sin(yaw0))], dtype=np.float64)

    @staticmethod
    def _yaw(v):
        return F32(_rem(math.atan2(v[3], v[2])))

    def _motion(self, x, u_d, u_th):
        xp = x.copy()
        yaw = self._yaw(x)
        dd = F32(u_d + self.v_d)
        xp[0] = x[0] + float(F32(dd * F32(math.cos(yaw))))
        xp[1] = x[1] + float(F32(dd * F32(math.sin(yaw))))
        ny = F32(_rem(float(F32(F32(yaw + u_th) + self.v_th))))
        xp[2] = float(F32(math.cos(ny))); xp[3] = float(F32(math.sin(ny)))
        return xp

    def _sense(self, x, li):
        yaw = self._yaw(self.x)
        dx, dy = x[li] - x[0], x[li + 1] - x[1]
        return np.array([math.sqrt(dx * dx + dy * dy) + float(self.w_r),
                         _rem(math.atan2(dy, dx) - float(yaw) + float(self.w_b))])

    def update(self, fwd, ang, meas):
        u_d, u_th = F32(fwd), F32(ang)
        n = 2 * self.M + 4
        w = float(F32((F32(1) - self.W0) / F32(2 * n)))
        Wts = np.full(2 * n + 1, w); Wts[0] = float(self.W0)
        yaw = self._yaw(self.x)
        Q = np.zeros((n, n))
        Q[0, 0] = self.V[0, 0] * float(F32(math.cos(yaw))); Q[1, 1] = self.V[0, 0] * float(F32(math.sin(yaw)))
        Q[2, 2] = self.V[1, 1] * float(F32(math.cos(yaw))); Q[3, 3] = self.V[1, 1] * float(F32(math.sin(yaw)))
        Y = 0.5 * (self.P + self.P.T) * float(F32(F32(2 * self.M + 4) / (F32(1) - self.W0)))
        D, Qv = np.linalg.eigh(Y)
        sq = (Qv * np.sqrt(np.maximum(D, 1e-8))) @ Qv.T
        X = np.zeros((n, 2 * n + 1))
        X[:, 0] = self.x
        for i in range(1, n + 1):
            X[:, i] = self.x + sq[:, i - 1]; X[:, i + n] = self.x - sq[:, i - 1]
        Xp = np.stack([self._motion(X[:, i], u_d, u_th) for i in range(2 * n + 1)], axis=1)
        xp = np.zeros(n)
        for i in range(2 * n + 1):
            xp = xp + Wts[i] * Xp[:, i]
        Pp = np.zeros((n, n))
        for i in range(2 * n + 1):
            d = Xp[:, i] - xp
            Pp = Pp + np.outer(Wts[i] * d, d)
        Pp = Pp + Q
        fresh = []
        for (idf, r, b) in meas:
            idn = int(idf)
            if idn in self.ids:
                li = 2 * self.ids.index(idn) + 4
                Z = np.stack([self._sense(Xp[:, i], li) for i in range(2 * n + 1)], axis=1)
                z_est = np.array([sum(Wts[i] * Z[0, i] for i in range(2 * n + 1)), 0.0])
                S = np.zeros((2, 2)); Cm = np.zeros((n, 2))
                for i in range(2 * n + 1):
                    d = Z[:, i] - z_est; d[1] = _rem(d[1])
                    S = S + np.outer(Wts[i] * d, d)
                    Cm = Cm + np.outer(Wts[i] * (Xp[:, i] - xp), d)
                S = S + self.W
                K = Cm @ np.linalg.inv(S)
                inn = np.array([float(F32(r)), float(F32(b))]) - z_est; inn[1] = _rem(inn[1])
                xp = xp + K @ inn
                Pp = Pp - K @ S @ K.T
            else:
                fresh.append((idn, F32(r), F32(b)))
        for idn, r, b in fresh:
            nn = len(xp)
            yw = self._yaw(xp)
            a = F32(yw + b)
            xp = np.concatenate([xp, [xp[0] + float(F32(r * F32(math.cos(a)))), xp[1] + float(F32(r * F32(math.sin(a))))]])
            Pn = np.zeros((nn + 2, nn + 2)); Pn[:nn, :nn] = Pp; Pn[nn:, nn:] = self.W
            Pp = Pn
            self.ids.append(idn); self.M += 1
        self.x, self.P = xp, Pp


KAT = [  # SURVEY.md Appendix E, UKF-SLAM table (float-overload variant), tolerance 1e-6 absolute
    ((0.1, 0.02, []), [9.99987537e-02, 0.0, 9.99787536e-01, 1.99984163e-02], [1.01000000e-02, 1.00249970e-04, 1.00000106e-02, 2.49868782e-05]),
    ((0.1, 0.02, [[3, 2.0, 0.5]]), [0.199976856, 0.002020882, 0.999176887, 0.040199062, 1.915177757, 1.030653880],
     [2.00980005e-02, 3.01032414e-04, 9.99804852e-03, 2.29545033e-04, 1, 1]),
    ((0.1, -0.01, [[3, 1.92, 0.53]]), [0.304516584, 0.006072063, 0.999400853, 0.030643361, 1.703221149, 1.085030494],
     [2.99070255e-02, 7.06575105e-04, 9.99222711e-03, 6.50766055e-04, 7.47483819e-01, 8.57351816e-01]),
]


@pytest.mark.parametrize("math_policy", [0, 1])
def test_ukf_known_answer_table(oracle, math_policy):
    u = oracle.OracleUKF(L_max=5, math=math_policy); u.init(0, 0, 0)
    for (fwd, ang, meas), x, dP in KAT:
        u.update(fwd, ang, meas)
        s = u.state()
        np.testing.assert_allclose(s["x"], x, atol=1e-6)
        np.testing.assert_allclose(np.diag(s["P"]), dP, atol=1e-6)
    s = u.state()
    assert s["P"][0, 4] == pytest.approx(6.46262797e-03, abs=1e-8) and s["P"][0, 5] == pytest.approx(2.07608233e-03, abs=1e-8)
    # quirk 8 signature: the landmark jumps on a consistent re-observation because the predicted bearing is 0
    assert abs(s["x"][4] - 1.703221149) < 1e-6


def test_ukf_oracle_matches_independent_numpy_transliteration(oracle):
    """120 steps of a reference measurement stream: C++ oracle (Jacobi sqrt, libm policy) vs numpy (LAPACK eigh)."""
    g = load_golden("sim_seed1_L20_T400.npz")
    u = oracle.OracleUKF(L_max=20, math=oracle.MATH_LIBM); u.init(0, 0, 0)
    ref = NumpyUKF(); ref.init(0, 0, 0)
    worst = 0.0
    for t in range(120):
        k = int(g["meas_count"][t]); m = g["meas"][t, :k]
        u.update(g["cmds"][t, 0], g["cmds"][t, 1], m)
        ref.update(g["cmds"][t, 0], g["cmds"][t, 1], [tuple(r) for r in m])
        s = u.state()
        assert s["M"] == ref.M and list(s["ids"]) == ref.ids
        worst = max(worst, np.abs(s["x"] - ref.x).max(), np.abs(s["P"] - ref.P).max())
    assert ref.M >= 1
    assert worst < 1e-8, worst


def test_jacobi_sqrt_against_lapack(oracle):
    rng = np.random.default_rng(3)
    for n in (4, 8, 44, 104):
        A = rng.normal(size=(n, n)); P = A @ A.T / n + np.diag(rng.uniform(1e-6, 1.0, n))
        P[0, 1] += 1e-13   # slightly asymmetric input, as the filter's P is
        out, sweeps = oracle.ukf_sqrt_probe(P, 7.5)
        Y = 0.5 * (P + P.T) * 7.5
        D, Q = np.linalg.eigh(Y)
        ref = (Q * np.sqrt(np.maximum(D, 1e-8))) @ Q.T
        assert 0 < sweeps < 20
        assert np.abs(out - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
        assert np.array_equal(out, out.T)
        assert np.abs(out @ out - Y).max() < 1e-11 * np.abs(Y).max()
    # indefinite input (signed Q can make P_pred indefinite, ukf.cpp:183-186): eigenvalues clamp at 1e-8
    P = np.diag([1.0, -0.5, 2.0, 1e-12])
    out, _ = oracle.ukf_sqrt_probe(P, 1.0)
    np.testing.assert_allclose(np.diag(out), [1.0, 1e-4, math.sqrt(2.0), 1e-4], rtol=1e-12)


def test_ukf_trajectory_invariants(oracle):
    g = load_golden("sim_seed0_L20_T1000.npz")
    u = oracle.OracleUKF(L_max=20); u.init(0, 0, 0)
    for t in range(300):
        k = int(g["meas_count"][t])
        fl = u.update(g["cmds"][t, 0], g["cmds"][t, 1], g["meas"][t, :k])
    s = u.state()
    assert fl == 0 and s["timestep"] == 300 and s["M"] >= 3
    assert np.all(np.isfinite(s["x"])) and np.all(np.isfinite(s["P"]))
    assert abs(math.hypot(s["x"][2], s["x"][3]) - 1.0) < 0.2      # (cos, sin) never renormalised, stays near 1
    assert np.abs(s["P"] - s["P"].T).max() < 1e-9


def test_ukf_double_trig_switch_is_close(oracle):
    """cos/sin(float) resolved to the double function instead of the float overload: <= ~1e-6 after 3 steps."""
    cfg = oracle.default_config(); cfg.ukf_float_trig = 0
    a = oracle.OracleUKF(L_max=5); b = oracle.OracleUKF(cfg=cfg, L_max=5)
    for u in (a, b):
        u.init(0, 0, 0)
        for (fwd, ang, meas), _, _ in KAT:
            u.update(fwd, ang, meas)
    d = np.abs(a.state()["x"] - b.state()["x"]).max()
    assert 0 < d < 1e-6


def test_jacobi_settles_on_clusters_of_equal_eigenvalues(oracle):
    """L=50 with every landmark mapped at step 0: P holds dozens of identical W blocks, so the scaled matrix has clusters of
    EQUAL diagonal entries with off-diagonals at the rounding level between them.  A rotation-sign convention that follows
    a_pq at a_pp == a_qq cycles there (30 % of the instances ran out of sweeps within 131 steps: SLAM_INST_SQRT_FAILED, a stale
    sqtP); the reference's `tau >= 0 -> +1` does not.  Guards the convention: no instance may flag."""
    from live_ekf_slam_amd.scenario import make_scenario
    L, T, B = 50, 131, 8
    lm, cmds = make_scenario(1234, L, T)
    vis = np.tile([3.0, -1.57, 1.57], (T, 1)); vis[0] = [1e9, -4.0, 4.0]
    r = oracle.run_ukf_batch(lm, cmds[:T], B, L, nthreads=8, want_P=False, vision=vis)
    assert not r["flags"].any(), r["flags"]
    assert (r["M"] == L).all()


def test_device_order_and_reference_order_arithmetic_stay_together(oracle):
    """the UKF oracle evaluates the eigen-iteration and the weighted covariance the way the kernel does (fused
    products in MFMA order, tau-free rotation parameters, warm start) so that GPU == oracle can be bit-exact.  Its
    REFERENCE-ORDER mode keeps the independent statement of ukf.cpp / Eigen semantics: no FMA (the reference's build has no
    contraction), textbook tau / t / c / s, a cold start every step.  The two must stay within rounding-level distance over a
    long trajectory, so that a device-driven change of the default mode cannot drift from the reference semantics unnoticed."""
    from live_ekf_slam_amd.scenario import make_scenario
    L, B, T = 20, 6, 300
    lm, cmds = make_scenario(77, L, T)
    a = oracle.run_ukf_batch(lm, cmds, B, L, seed=5, nthreads=6)
    b = oracle.run_ukf_batch(lm, cmds, B, L, seed=5, nthreads=6, ref_order=True)
    assert not a["flags"].any() and not b["flags"].any() and np.array_equal(a["M"], b["M"]) and a["M"].max() >= 5
    dx = np.abs(a["x"] - b["x"]).max(); dP = np.abs(a["P"] - b["P"]).max()
    assert 0.0 < dx < 2e-5 and dP < 2e-5, (dx, dP)      # not the same bits (different rounding), the same filter
    assert np.abs(a["avg_err"] - b["avg_err"]).max() < 1e-6
