// jacobi_schedule.h — which index pairs (p < q) rotate together in round t = 0 .. n - 2 of a sweep of the parallel-order Jacobi
// eigen-iteration of the UKF (pair k = 0 .. n / 2 - 1; the pairs of a round are disjoint, a sweep visits every pair once).  The
// oracle restates the same schedule (oracle/slam_oracle_ukf.cpp): GPU == oracle is a bit-exact statement, so the order is shared.
#pragma once
#include <hip/hip_runtime.h>

namespace slam {

// the circle method over the n indices: position 0 fixed, the others move one slot per round
__host__ __device__ inline void rr_pair(int k, int t, int n, int& p, int& q) {
    const int nm1 = n - 1;
    int x = k - 1 + t;
    if (x >= nm1) x -= nm1;
    const int a = (k == 0) ? 0 : 1 + x;
    int y = nm1 - k - 1 + t;
    if (y >= nm1) y -= nm1;
    const int bq = 1 + y;
    p = a < bq ? a : bq;
    q = a < bq ? bq : a;
}

// n divisible by four: the circle method over the n / 2 BLOCKS of two consecutive indices.  Block round T = 0 .. n / 2 - 2 pairs the
// blocks into quadruples (a, b | c, d) = (2X, 2X+1 | 2Y, 2Y+1), X < Y, and takes two rounds: t = 1 + 2T: (a, c) (b, d); t = 2 + 2T:
// (a, d) (b, c).  Round 0 rotates inside the blocks, (a, b) (c, d), indexed by the quadruples of block round 0.  Pairs 2 kb and
// 2 kb + 1 belong to quadruple kb.  Two consecutive rounds stay inside the same 4 x 4 blocks of A and the same four rows of V^T:
// the sqrt kernels of the LDS size classes keep them in registers across both (half the passes over LDS, half the barriers).
// Those kernels (and the oracle for those classes) PAD a state size n = 2 (mod 4) by two all-zero rows to n + 2 and call this with the padded
// size; n = 2 (mod 4) itself - the circle method over the indices - is what the HBM-streamed class (ukf_big_kernel.hip) still runs.
__host__ __device__ inline void jacobi_pair(int k, int t, int n, int& p, int& q) {
    if (n & 2) { rr_pair(k, t, n, p, q); return; }
    const int kb = k >> 1, u = k & 1;
    int X, Y;
    rr_pair(kb, t == 0 ? 0 : (t - 1) >> 1, n >> 1, X, Y);
    if (t == 0) { p = 2 * (u ? Y : X); q = p + 1; return; }
    const int s = (t - 1) & 1;
    p = 2 * X + u;
    q = 2 * Y + (s ? 1 - u : u);
}

}  // namespace slam
