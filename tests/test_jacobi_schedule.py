"""The Jacobi schedule the UKF oracle and the kernels share (oracle/slam_oracle_ukf.cpp jacobi_pair, csrc/jacobi_schedule.h):
a sweep is n - 1 rounds of n / 2 disjoint pairs and visits every pair once; for n divisible by four two consecutive rounds
stay inside quadruples of indices (what the sqrt kernels keep in registers); the device header lists the same pairs."""
import os, subprocess, sys
import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sweep_pairs(O, n):
    return [[O.ukf_jacobi_pair(k, t, n) for k in range(n // 2)] for t in range(n - 1)]


@pytest.mark.parametrize("n", [4, 6, 8, 10, 12, 22, 42, 44, 102, 104, 204])
def test_a_sweep_visits_every_pair_once_in_disjoint_rounds(oracle, n):
    seen = set()
    for rnd in sweep_pairs(oracle, n):
        idx = [i for pq in rnd for i in pq]
        assert sorted(idx) == list(range(n))                # disjoint, and every index rotates in every round
        for p, q in rnd:
            assert 0 <= p < q < n and (p, q) not in seen
            seen.add((p, q))
    assert len(seen) == n * (n - 1) // 2


@pytest.mark.parametrize("n", [4, 8, 12, 44, 104])
def test_two_rounds_of_a_pass_stay_inside_quadruples(oracle, n):
    """n = 0 (mod 4): round 0 rotates inside the blocks (2B, 2B+1); rounds 1 + 2T and 2 + 2T pair the same two blocks
    (a, b | c, d) as (a, c) (b, d), then (a, d) (b, c); pairs 2 kb and 2 kb + 1 belong to quadruple kb; round 0 is indexed
    by the quadruples of block round 0."""
    rounds = sweep_pairs(oracle, n)
    for k, (p, q) in enumerate(rounds[0]):
        assert p % 2 == 0 and q == p + 1
    for T in range(n // 2 - 1):
        r1, r2 = rounds[1 + 2 * T], rounds[2 + 2 * T]
        for kb in range(n // 4):
            (a, c), (b, d) = r1[2 * kb], r1[2 * kb + 1]
            assert a % 2 == 0 and b == a + 1 and c % 2 == 0 and d == c + 1 and a < c
            assert r2[2 * kb] == (a, d) and r2[2 * kb + 1] == (b, c)
            if T == 0:
                assert rounds[0][2 * kb] == (a, b) and rounds[0][2 * kb + 1] == (c, d)


@pytest.mark.parametrize("n", [6, 10, 42])
def test_other_sizes_use_the_circle_method_over_the_indices(oracle, n):
    """(What the HBM-streamed class runs at n = 2 mod 4; the LDS classes pad such a size to n + 2 and walk the quadruples -
    test_padded_sizes_give_the_square_root below.)"""
    for t, rnd in enumerate(sweep_pairs(oracle, n)):
        at = lambda k: 0 if k == 0 else 1 + ((k - 1 + t) % (n - 1))
        for k, (p, q) in enumerate(rnd):
            a, b = at(k), at(n - 1 - k)
            assert (p, q) == (min(a, b), max(a, b))


def test_device_header_lists_the_same_pairs(oracle, tmp_path):
    """csrc/jacobi_schedule.h compiled for the host (hipcc, no GPU needed) against the oracle's function."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    src = tmp_path / "sched.cpp"
    src.write_text('#include <cstdio>\n#include "jacobi_schedule.h"\nint main() { for (int n = 4; n <= 204; n += 2) for (int t = 0; t < n - 1; ++t) '
                   'for (int k = 0; k < n / 2; ++k) { int p, q; slam::jacobi_pair(k, t, n, p, q); std::printf("%d %d\\n", p, q); } return 0; }\n')
    exe = tmp_path / "sched"
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-I", os.path.join(REPO, "live_ekf_slam_amd", "csrc"),
                           str(src), "-o", str(exe)])
    out = np.array(subprocess.check_output([str(exe)]).split(), dtype=np.int64).reshape(-1, 2)
    ref = [oracle.ukf_jacobi_pair(k, t, n) for n in range(4, 206, 2) for t in range(n - 1) for k in range(n // 2)]
    assert np.array_equal(out, np.array(ref, dtype=np.int64))


@pytest.mark.parametrize("n", [6, 10, 22, 42, 102])
def test_padded_sizes_give_the_square_root(oracle, n):
    """n = 2 (mod 4) runs padded by a decoupled 2 x 2 zero block (oracle and kernels alike): the n x n result is the principal square
    root of the clamped matrix all the same (LAPACK, 1e-12)."""
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n)); P = A @ A.T / n + np.diag(rng.uniform(1e-6, 1.0, n))
    scale = float(np.float32(n) / np.float32(0.8))
    out, sweeps = oracle.ukf_sqrt_probe(P, scale)
    Y = 0.5 * (P + P.T) * scale
    D, Q = np.linalg.eigh(Y)
    ref = (Q * np.sqrt(np.maximum(D, 1e-8))) @ Q.T
    assert sweeps >= 0 and np.abs(out.reshape(n, n) - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
