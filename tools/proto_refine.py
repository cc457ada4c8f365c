"""Prototype (numpy): iterative refinement of a symmetric eigen-decomposition (Ogita & Aishima 2018) as the UKF's
nearestSPD + sqrt (ukf.cpp:106-123,208), warm-started from the previous timestep's eigenvectors - everything is GEMM-shaped.
Measures, on the P sequence of oracle UKF runs fed by the golden measurement streams: iterations to convergence, failures,
accuracy of sqtP against LAPACK."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

def refine(A, X, iters=8, tol=1e-13, verbose=False):
    n = A.shape[0]
    I = np.eye(n)
    anorm = np.abs(A).sum(axis=1).max()
    hist = []
    for it in range(iters):
        W = A @ X
        S = X.T @ W
        R = I - X.T @ X
        lam = np.diag(S) / (1.0 - np.diag(R))
        D = np.diag(lam)
        # delta = 2 (||S - D||_2 + ||A||_2 ||R||_2), norms bounded by the max row sum (inf-norm >= 2-norm for symmetric)
        delta = 2.0 * (np.abs(S - D).sum(axis=1).max() + anorm * np.abs(R).sum(axis=1).max())
        dl = lam[None, :] - lam[:, None]          # lam_j - lam_i at [i, j]
        far = np.abs(dl) > delta
        E = np.where(far, (S + lam[None, :] * R) / np.where(far, dl, 1.0), R / 2.0)
        np.fill_diagonal(E, np.diag(R) / 2.0)
        X = X + X @ E
        e = np.abs(E).max()
        hist.append(e)
        if verbose: print("   it", it, "max|E|", e, "delta", delta)
        if e < tol:
            break
    W = A @ X
    lam = np.einsum("ij,ij->j", X, W) / np.einsum("ij,ij->j", X, X)
    return X, lam, hist

def run(fixture, L, T=None):
    from tests.conftest import load_golden
    g = load_golden(fixture)
    cmds, meas, cnt = g["cmds"], g["meas"], g["meas_count"]
    T = T or len(cmds)
    f = O.OracleUKF(L_max=L); f.init(0, 0, 0)
    Xprev = None; nprev = 0
    its, fails, errs = [], 0, []
    for t in range(T):
        s = f.state()
        P = s["P"]; n = P.shape[0]; M = s["M"]
        scale = float(np.float32((2 * M + 4) / np.float32(1 - np.float32(0.2))))
        Y = 0.5 * (P + P.T) * scale
        w, V = np.linalg.eigh(Y)
        ref = (V * np.sqrt(np.maximum(w, 1e-8))) @ V.T
        if Xprev is not None:
            X0 = np.eye(n); X0[:nprev, :nprev] = Xprev
            X, lam, hist = refine(Y, X0)
            sq = (X * np.sqrt(np.maximum(lam, 1e-8))) @ X.T
            err = np.abs(sq - ref).max() / max(np.abs(ref).max(), 1e-300)
            ok = hist[-1] < 1e-13 and err < 1e-9
            if not ok:
                fails += 1
                if fails <= 6:
                    print(f"  t={t} n={n} nprev={nprev} hist={['%.1e' % h for h in hist]} err={err:.2e} min_eig={w.min():.3e} max={w.max():.3e}")
                X = V
            its.append(len(hist)); errs.append(err if ok else 0.0)
            Xprev = X
        else:
            Xprev = V
        nprev = n
        k = int(cnt[t])
        f.update(cmds[t, 0], cmds[t, 1], meas[t, :k])
    its = np.array(its)
    print(f"{fixture} L={L} T={T}: steps {len(its)}, fails {fails}, iterations mean {its.mean():.2f} max {its.max()}, hist {np.bincount(its)}, max rel err of sqtP {max(errs):.2e}")

if __name__ == "__main__":
    import numpy as np
    d = np.load("tests/golden/sim_seed0_L20_T1000.npz"); print(list(d.keys()))
    run("sim_seed0_L20_T1000.npz", 20, 400)
    run("sim_seed2_L50_T1000.npz", 50, 200)
