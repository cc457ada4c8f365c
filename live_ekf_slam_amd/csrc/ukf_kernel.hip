// ukf_kernel.hip — UKF-SLAM step for gfx950 (MI355X): UKF::update of the reference
// (ekf_ws/src/localization_pkg/src/ukf.cpp:161-372) as two kernels per timestep, one workgroup per instance.
//
//  ukf_sqrt_kernel  nearestSPD + principal matrix square root (ukf.cpp:106-123,208).  The reference calls Eigen's
//                   SelfAdjointEigenSolver and MatrixFunctions sqrt; here: cyclic Jacobi in PARALLEL ORDER entirely in LDS
//                   - A packed lower-triangular (exact symmetry by construction), V^T full - n/2 disjoint rotations per
//                   round (jacobi_schedule.h), every pair-block B' = R_i^T B R_j and every V row-pair an independent work
//                   item.  The state size, padded to a multiple of four, walks the schedule in PASSES of two rounds that stay inside
//                   quadruples of indices: the 4 x 4 blocks of A and four rows of V^T in registers across both, one barrier
//                   per pass.  This O(n^3 * sweeps) fp64 phase dominates the UKF (70 % of its GPU time); it is bound by
//                   VALU issue (82 % utilisation at six workgroups per CU, profiles/r04_ukf/).
//  ukf_step_kernel  sigma points through the motion model, weighted mean and covariance (sequential in the sigma
//                   index exactly like the reference's accumulation loops), all landmark updates (the reference never
//                   redraws sigma points, so K and S of every update are independent of P), insertions, and ONE
//                   pass that forms P_pred, adds Q, subtracts the K S K^T terms and writes P_t in its final packed
//                   leading dimension.  sqtP lives in LDS; P is never staged.
//
// Arithmetic: plain IEEE fp64 (-ffp-contract=off), same operation order as the CPU oracle
// (oracle/slam_oracle_ukf.cpp) so results are bit-identical; the reference's float truncations are real fp32 ops.
#include "ukf_kernel.h"

#include <stdlib.h>

#include <vector>

#include "../../include/slam_batch.h"
#include "jacobi_schedule.h"
#include "sim_device.h"
#include "slam_math.h"
#include "slam_rng.h"

typedef double dbl4_t __attribute__((ext_vector_type(4)));   // C/D of v_mfma_f64_16x16x4_f64

#ifndef SLAM_UKF_PRIO
#define SLAM_UKF_PRIO 2
#endif
#ifndef SLAM_UKF_SQRT_WG
#define SLAM_UKF_SQRT_WG 6   // workgroups of ukf_sqrt_kernel<44, 256> the compiler must leave room for on a CU (25 KB of LDS each allow six; 80 VGPRs)
#endif

namespace slam {

namespace {

constexpr float kW0 = 0.2f;  // filter.h:207

__device__ __forceinline__ bool inv2x2_lu_ukf(const double S[4], double Si[4]) {  // MatrixXd::inverse() (ukf.cpp:339)
    const bool sw = fabs(S[2]) > fabs(S[0]);
    const double a00 = sw ? S[2] : S[0], a01 = sw ? S[3] : S[1];
    const double a10 = sw ? S[0] : S[2], a11 = sw ? S[1] : S[3];
    const double l = a10 / a00;
    const double u11 = a11 - l * a01;
    const bool ok = (a00 != 0.0) && (u11 != 0.0);
    {
        const double r0 = sw ? 0.0 : 1.0, r1 = sw ? 1.0 : 0.0;
        const double y1 = r1 - l * r0;
        const double x1 = y1 / u11;
        Si[0] = (r0 - a01 * x1) / a00;
        Si[2] = x1;
    }
    {
        const double r0 = sw ? 1.0 : 0.0, r1 = sw ? 0.0 : 1.0;
        const double y1 = r1 - l * r0;
        const double x1 = y1 / u11;
        Si[1] = (r0 - a01 * x1) / a00;
        Si[3] = x1;
    }
    return ok;
}

// unqualified cos / sin on a float argument (ukf.cpp:39-42,129-133,183-186,358-359)
__device__ __forceinline__ void tsincos(float a, int float_trig, double* s, double* c) {
    double ss, cc;
    det_sincos((double)a, &ss, &cc);
    *s = float_trig ? (double)(float)ss : ss;
    *c = float_trig ? (double)(float)cc : cc;
}

__device__ __forceinline__ float yaw_of(double c, double s) {  // (float) remainder(atan2(x3, x2), 2 pi)
    return (float)remainder(det_atan2(s, c), kTwoPi);
}

// a double from another lane of the same group of four (DPP quad_perm; every lane of the group must be active)
template <int CTRL>
__device__ __forceinline__ double dpp_quad(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double block_max(double v, double* s_red, int tid, int tpb) {
    for (int o = 32; o > 0; o >>= 1) {
        const double w = __shfl_down(v, o);
        v = v > w ? v : w;
    }
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    double r = s_red[0];
    for (int i = 1; i < tpb / 64; ++i) r = r > s_red[i] ? r : s_red[i];
    return r;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Pass table (round 4; one per padded state size nj = 4, 8, .., 44).  jacobi_schedule.h pairs the nj / 2 blocks of two consecutive
// indices by the circle method; a block round ("pass") T holds n / 4 QUADRUPLES (a, b | c, d) and two rounds of the schedule
// - (a, c) (b, d), then (a, d) (b, c); pass 0 also the in-block round (a, b) (c, d) before them - which touch nothing outside the
// quadruple's four rows / columns.  The kernel therefore reads every element of A and V once per PASS, not per round:
//   * the 4 x 4 block of A between quadruples I > J belongs to four adjacent lanes, lane 2 i + j holding rows (pair i of I) x
//     columns (pair j of J); between the two rounds the lanes swap three of their four elements by DPP quad_perm;
//   * a V item is (quadruple, 16-byte pair of columns k): four rows of V^T through all the rotations of the pass;
//   * the diagonal 4 x 4 block of a quadruple belongs to two PARAMETER lanes, one per pair of the round: parameters from its pair's
//     three elements, the pair's diagonal block, then (lane 1) the cross block with both rotations, round after round through LDS
//     (one wavefront's LDS accesses execute in order).  This runs a pass ahead, in wavefront 0,
//     which also owns the "critical" blocks - the ones that hold the next pass's pivots: the cross block of next quadruple (X, Y)
//     lies in the block between the current quadruples of X and of Y.  One barrier per pass.
// Entry (32 bytes = w[0..7]) of thread tid in pass T at size n:
//   w7 = kind (0 none, 1 block lane) | I << 8 | J << 16 | i << 24 | j << 25
//   block lane: w0 w1 = LDS byte offsets of its four elements in the first round of the pass (e00 | e01 << 16, e10 | e11 << 16),
//               w2 w3 = in the second round, w4 w5 = in the in-block round (pass 0)
//   w6 = blocks X | Y << 8 of the thread's two V items (second item << 16), 0xff = none (threads 64 ..); lanes 2 q + u < n / 2 of
//        wavefront 0: the blocks of quadruple q, whose parameter lanes they are
// ------------------------------------------------------------------------------------------------------------------
hipError_t launch_ukf_quad_table(uint4* tab, hipStream_t stream) {
    std::vector<uint32_t> h(kUkfQuadTabEntries * 4, 0u);
    auto idx = [](int r, int c) { return (unsigned)(8 * (r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r)); };
    for (int n = 4; n <= 4 * (kUkfQuadSizes - 1); n += 4) {
        const int m = n / 2, mq = n / 4;
        int owner0[kUkfQuadSizes][kUkfQuadSizes];
        for (int T = 0; T < m - 1; ++T) {
            int X[kUkfQuadSizes], Y[kUkfQuadSizes], quad_of[2 * kUkfQuadSizes];
            for (int q = 0; q < mq; ++q) { rr_pair(q, T, m, X[q], Y[q]); quad_of[X[q]] = q; quad_of[Y[q]] = q; }
            // owner lanes of the off-diagonal blocks: critical ones in wavefront 0 (lanes 16 + 4 c), the others from thread 64 on
            int owner[kUkfQuadSizes][kUkfQuadSizes];
            for (int i = 0; i < mq; ++i) for (int j = 0; j < mq; ++j) owner[i][j] = -1;
            const int Tn = T + 1 < m - 1 ? T + 1 : 0;
            int ncrit = 0;
            for (int q = 0; q < mq && mq > 1; ++q) {
                int xn, yn;
                rr_pair(q, Tn, m, xn, yn);
                int I = quad_of[xn], J = quad_of[yn];
                if (I < J) { const int t = I; I = J; J = t; }
                if (I != J && owner[I][J] < 0) owner[I][J] = 4 * ncrit++;
            }
            int nother = 0;
            for (int I = 1; I < mq; ++I) for (int J = 0; J < I; ++J) if (owner[I][J] < 0) owner[I][J] = 64 + 4 * nother++;
            if (4 * ncrit > 64 || 64 + 4 * nother > kUkfRotThreads) return hipErrorInvalidValue;
            // the critical blocks sit at the same quadruple POSITIONS in every pass (circle method: next pair k = top of k + 1, bottom of k - 1),
            // so a lane owns the same block (I, J) throughout: the kernel reads w7 once per launch
            for (int I = 1; I < mq; ++I) for (int J = 0; J < I; ++J) {
                if (T == 0) owner0[I][J] = owner[I][J];
                else if (owner0[I][J] != owner[I][J]) return hipErrorInvalidValue;
            }
            uint32_t* const base = h.data() + ((size_t)(n / 4) * kUkfQuadPasses + T) * kUkfRotThreads * 8;
            for (int tid = 0; tid < kUkfRotThreads; ++tid) { base[8 * tid + 6] = 0xffffffffu; }
            for (int q = 0; q < mq; ++q)   // parameter lanes 2 q + u (they are block lanes as well): the blocks of their quadruple
                for (int u = 0; u < 2; ++u) base[8 * (2 * q + u) + 6] = 0xffff0000u | (unsigned)X[q] | ((unsigned)Y[q] << 8);
            for (int I = 1; I < mq; ++I)
                for (int J = 0; J < I; ++J) {
                    const int row[4] = {2 * X[I], 2 * X[I] + 1, 2 * Y[I], 2 * Y[I] + 1};
                    const int col[4] = {2 * X[J], 2 * X[J] + 1, 2 * Y[J], 2 * Y[J] + 1};
                    for (int i = 0; i < 2; ++i)
                        for (int j = 0; j < 2; ++j) {
                            uint32_t* w = base + 8 * (owner[I][J] + 2 * i + j);
                            auto four = [&](int r0, int r1, int c0, int c1, uint32_t* o) {
                                o[0] = idx(row[r0], col[c0]) | (idx(row[r0], col[c1]) << 16);
                                o[1] = idx(row[r1], col[c0]) | (idx(row[r1], col[c1]) << 16);
                            };
                            four(i, i + 2, j, j + 2, w + 0);
                            four(i, 3 - i, j, 3 - j, w + 2);
                            four(2 * i, 2 * i + 1, 2 * j, 2 * j + 1, w + 4);
                            w[7] = 1u | ((unsigned)I << 8) | ((unsigned)J << 16) | ((unsigned)i << 24) | ((unsigned)j << 25);
                        }
                }
            for (int tid = 64; tid < kUkfRotThreads; ++tid) {   // V items: (quadruple, pair of columns), n / 2 pairs per quadruple
                uint32_t v = 0xffffffffu;
                for (int u = 1; u >= 0; --u) {
                    const int it = tid - 64 + (kUkfRotThreads - 64) * u, Q = it / m;
                    v = (v << 16) | (Q < mq ? ((unsigned)X[Q] | ((unsigned)Y[Q] << 8)) : 0xffffu);
                }
                base[8 * tid + 6] = v;
            }
        }
    }
    hipError_t e = hipMemcpyAsync(tab, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);   // h is a pageable host array that goes out of scope
    return e;
}

// ------------------------------------------------------------------------------------------------------------------
// nearestSPD + sqrt
// ------------------------------------------------------------------------------------------------------------------
// PROF = true compiles the phase timers in (a separate instantiation, launched only when the debug buffer is attached:
// as a run-time option they cost the production kernel 20 VGPRs = one wavefront per SIMD of occupancy, -9 % steps/s).
template <int NMAX, int TPB, bool PROF = false>
__global__ __launch_bounds__(TPB, (NMAX == 44 && TPB == 256 && !PROF) ? SLAM_UKF_SQRT_WG : 1) void ukf_sqrt_kernel(const UkfStepParams p) {
    constexpr int MMAX = NMAX / 2;
    __shared__ double sA[NMAX * (NMAX + 1) / 2];   // packed lower triangle: A(r,c), r >= c, at r(r+1)/2 + c
    __shared__ __attribute__((aligned(16))) double sVt[NMAX * NMAX];   // V transposed: Vt[p*n + k] = V(k, p)
    __shared__ double s_cs[MMAX], s_sn[MMAX], s_tn[MMAX], s_sd[NMAX];
    __shared__ int s_pp[MMAX], s_qq[MMAX];
    constexpr bool kQuadLds = (MMAX * (MMAX + 1) / 2 + TPB - 1) / TPB <= 2;   // the variants that walk the schedule in passes of two rounds (kFast below)
    __shared__ double2 s_csn[(kQuadLds ? 6 : 2) * MMAX];   // (c, s) of the rotations, one 16-byte read per consumer: [pass parity][round of the pass][pair] (round-by-round variants: [pair])
    __shared__ int s_qflag[kQuadLds ? 2 * ((MMAX + 1) / 2) : 1];   // passes: [pass parity][quadruple] any rotation of the pass that is not the identity
    __shared__ int s_xy[kQuadLds ? 2 * ((MMAX + 1) / 2) : 1];      // passes without the table: [pass parity][quadruple] its blocks X | Y << 8
    __shared__ int s_pass_flag;                                    // passes without the table: wavefront 1 -> wavefront 0, "critical blocks of pass # written"

    const int b = blockIdx.x + p.b_off, tid = threadIdx.x;
    const int M = p.M[b];
    const int n = 4 + 2 * M, m = n / 2;
    const double* __restrict__ Pb = p.P + (size_t)b * p.pstride;
    double* __restrict__ Sq = p.sqtP + (size_t)b * p.pstride;
    auto AT = [&](int r, int c) -> double& { return r >= c ? sA[r * (r + 1) / 2 + c] : sA[c * (c + 1) / 2 + r]; };

    // phase timers (debug, SLAM_DEBUG_FLAGS & 4): slots 10..15 of the [B][16] buffer whose slots 0..9 the step kernel fills.
    // tid 0 stamps right after a barrier, so a delta is one whole barrier-to-barrier phase as wavefront 0 sees it.
    unsigned long long sacc[PROF ? 6 : 1] = {0}, sprev = 0ull;
    if constexpr (PROF) sprev = wall_clock64();
#define SQ_STAMP(i) do { if constexpr (PROF) { if (tid == 0) { const unsigned long long now_ = wall_clock64(); sacc[i] += now_ - sprev; sprev = now_; } } } while (0)
    const float scale_f = (float)(2 * M + 4) / (1 - kW0);   // ukf.cpp:114, evaluated in float
    const double scale = (double)scale_f;
    // Warm start: the eigenvectors of the previous timestep (extended by the identity for landmarks inserted since)
    // make V0^T A V0 nearly diagonal, so 4 sweeps instead of 8 converge.  Same arithmetic as the oracle's warm path.
    constexpr int kWarmMaxAge = 100;
    const int age = p.v_age[b], n_v = p.n_sq[b];
    const bool warm = age >= 0 && age < kWarmMaxAge && n_v > 0 && n_v <= n;
    double* Vs = p.Vt_store + (size_t)b * p.pstride;   // V0^T on entry; scratch for T once V0 sits in LDS; V^T on exit
    {   // the global loads of four elements are issued before the first value is used (a quarter of the memory round trips
        // of one element at a time: the ISA had vmcnt(0) after each pair of loads).  Four, not all eight: the whole batch
        // made this prologue the register peak of the kernel (100 VGPRs) and cost the 44/256 variant its fifth wavefront.
        constexpr int NE = (NMAX * NMAX + TPB - 1) / TPB, NB4 = 4;
        // e / n for e < n^2 <= 10 816 as a multiplication: floor(e ceil(2^20 / n) / 2^20) is exact while e < 2^20 / n (one division per launch
        // instead of two per element: the index arithmetic of this loop was 2 % of the <44, 256> kernel's VALU instructions)
        const unsigned ninv = ((1u << 20) + (unsigned)n - 1u) / (unsigned)n;
#pragma unroll 1
        for (int u0 = 0; u0 < NE; u0 += NB4) {
            double pa[NB4], pb[NB4], pv[NB4];
#pragma unroll
            for (int u = 0; u < NB4; ++u) {
                const int e = tid + TPB * (u0 + u);
                const bool in = e < n * n;
                const int r = in ? (int)(((unsigned)e * ninv) >> 20) : 0, c = in ? e - r * n : 0;   // e / n (see ninv)
                const bool lower = in && c <= r, wv = in && warm && r < n_v && c < n_v;
                pa[u] = lower ? Pb[(size_t)r * n + c] : 0.0;
                pb[u] = lower ? Pb[(size_t)c * n + r] : 0.0;
                pv[u] = wv ? Vs[(size_t)r * n_v + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < NB4; ++u) {
                const int e = tid + TPB * (u0 + u);
                if (e < n * n) {
                    const int r = (int)(((unsigned)e * ninv) >> 20), c = e - r * n;
                    if (c <= r) sA[r * (r + 1) / 2 + c] = (0.5 * (pa[u] + pb[u])) * scale;
                    sVt[e] = (warm && r < n_v && c < n_v) ? pv[u] : ((r == c) ? 1.0 : 0.0);   // row r = eigenvector r
                }
            }
        }
    }
    __syncthreads();
    if (warm) {
        // T = A V0 (n x n, through the V slab in HBM/L2 as scratch: its content is in sVt now, and the stale sqtP must
        // survive a failed decomposition, ukf.cpp:209-211), then B = V0^T T (lower triangle) into sA.  Round 3: both products
        // on v_mfma_f64_16x16x4_f64 (they were 14.5 % of this kernel as 2 n^3 scalar multiply-adds out of LDS): one 16 x 16
        // tile per wavefront at a time, k in steps of four; per output element the instruction chain is
        // acc = fma(a_k, b_k, acc) in ascending k (tools/ubench_mfma_f64.hip), which is what the oracle evaluates; rows,
        // columns and k beyond n give zero operands (fma(0, 0, acc) = acc).
        const int nt = (n + 15) >> 4, nk = (n + 3) >> 2;
        const int wv = tid >> 6, ln = tid & 63, kq = ln >> 4, cl = ln & 15;
#pragma unroll 1
        for (int t = wv; t < nt * nt; t += TPB / 64) {
            const int tr = t / nt, tc = t - tr * nt;
            const int ar = 16 * tr + cl, bc = 16 * tc + cl;     // A-operand row (of A), B-operand column (of V0)
            const bool va = ar < n, vb = bc < n;
            const int arc = va ? ar : 0, bcc = vb ? bc : 0;
            dbl4_t acc = dbl4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
            for (int ks = 0; ks < nk; ++ks) {
                const int k = 4 * ks + kq;
                const bool vk = k < n;
                const int kc = vk ? k : 0;
                double a = arc >= kc ? sA[arc * (arc + 1) / 2 + kc] : sA[kc * (kc + 1) / 2 + arc];   // A(ar, k), symmetric
                double bv = sVt[bcc * n + kc];                                                       // V0(k, bc)
                a = (va && vk) ? a : 0.0;
                bv = (vb && vk) ? bv : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {   // C/D layout: row = (lane >> 4) + 4 * reg, column = lane & 15
                const int r = 16 * tr + kq + 4 * r4, c = 16 * tc + cl;
                if (r < n && c < n) Vs[(size_t)r * n + c] = acc[r4];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int t = wv; t < nt * nt; t += TPB / 64) {
            const int tr = t / nt, tc = t - tr * nt;
            if (tc > tr) continue;                              // lower triangle of B only (wave-uniform)
            const int ar = 16 * tr + cl, bc = 16 * tc + cl;     // A-operand: column ar of V0 (row of V0^T); B-operand: column bc of T
            const bool va = ar < n, vb = bc < n;
            const int arc = va ? ar : 0, bcc = vb ? bc : 0;
            dbl4_t acc = dbl4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int ks = 0; ks < nk; ++ks) {
                const int k = 4 * ks + kq;
                const bool vk = k < n;
                const int kc = vk ? k : 0;
                double a = sVt[arc * n + kc];                   // V0(k, ar)
                double bv = Vs[(size_t)kc * n + bcc];           // T(k, bc): back from L2, several k-steps in flight
                a = (va && vk) ? a : 0.0;
                bv = (vb && vk) ? bv : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = 16 * tr + kq + 4 * r4, c = 16 * tc + cl;
                if (r < n && c <= r) sA[r * (r + 1) / 2 + c] = acc[r4];
            }
        }
        __syncthreads();
    }
    const int tiny_from = warm ? 0 : 3;
    // The Jacobi iteration runs on the state size padded to a multiple of four, nj = n or n + 2, so that every size walks the schedule over
    // quadruples (jacobi_schedule.h; the oracle pads likewise): two more rows of A (packed: the elements n (n + 1) / 2 ..) and of V^T, all zero.
    // Every rotation with one of the two extra indices is the identity (a_pq = 0), their partners rest in those rounds, the arithmetic on the
    // n x n part is that of the oracle's padded matrix; the columns >= n of V^T are never formed (they would stay zero / one).
    const int nj = (n + 3) & ~3, mj = nj >> 1;
    if (nj != n) {
        for (int e = n * (n + 1) / 2 + tid; e < nj * (nj + 1) / 2; e += TPB) sA[e] = 0.0;
        for (int e = n * n + tid; e < nj * n; e += TPB) sVt[e] = 0.0;
        __syncthreads();
    }
    SQ_STAMP(0);   // load, symmetrise, warm-start transform

    const int nb = mj * (mj - 1) / 2;       // (the item loop of the variants without passes: pair-blocks, diagonal blocks, V rows of the PADDED size)
    const int items = nb + mj + mj * n;
    // The work items a thread owns are the same in every round: decode them once.
    //   kind 0: pair-block (i, j), i > j      B' = R_i^T B R_j
    //   kind 1: diagonal block of pair i
    //   kind 2: row k of V for pair i         V <- V J
    constexpr int IT = (MMAX * (MMAX - 1) / 2 + MMAX + MMAX * NMAX + TPB - 1) / TPB;
    int desc[IT];
#pragma unroll
    for (int u = 0; u < IT; ++u) {
        const int it = tid + TPB * u;
        int d = -1;
        if (it < nb) {
            int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)it)) * 0.5f);
            while (i * (i - 1) / 2 > it) --i;
            while ((i + 1) * i / 2 <= it) ++i;
            d = (i << 8) | (it - i * (i - 1) / 2);
        } else if (it < nb + mj) {
            d = (1 << 16) | ((it - nb) << 8);
        } else if (it < items) {
            const int e = it - nb - mj;
            const int i = e / n;
            d = (2 << 16) | (i << 8) | (e - i * n);
        }
        desc[u] = d;
    }
    // The variants with at most two pair-block items per thread walk the schedule in PASSES of two rounds (below); <44, 256> takes the LDS
    // addresses of a pass from the pass table, the others derive them from the quadruples' block numbers.
    constexpr int ITB = (MMAX * (MMAX + 1) / 2 + TPB - 1) / TPB;
    constexpr bool kFast = ITB <= 2;
    constexpr bool kTab = ITB <= 2 && NMAX == 44 && TPB == kUkfRotThreads;
    constexpr int VT0 = kFast ? 64 : 0;      // first thread that has V items (wavefront 0 carries the parameter chain)
    bool converged = false;
    int pass_seq = 1;                       // passes without the table: number of the pass (s_pass_flag)
    if (tid == 0) s_pass_flag = 0;          // (the barriers of the prologue come before its first use)
    uint4 te_next = make_uint4(0u, 0u, 0u, 0u);
    const uint4* const qtab = kTab ? p.quad_tab + ((size_t)(nj >> 2) * kUkfQuadPasses * kUkfRotThreads + tid) * 2 : nullptr;   // + 2 * 256 * pass
    unsigned qnz = 0u;   // pass table: the words of the NEXT pass's entry that every pass needs - (te_next = w0..w3, qnz = w6) - requested a pass ahead
    unsigned qw7 = 0u;   // w7 (the same in every pass)
    if constexpr (kTab) { te_next = qtab[0]; qnz = qtab[1].z; qw7 = qtab[1].w; }
    int par = 0;                            // passes: parity of the parameter buffers the current pass reads
    // rotation parameters (c, s, t = tan) of the pair (pidx, qidx) from the current A; sw: the sweep the rotation belongs to
    auto jacobi_param = [&](double app, double aqq, double apq, int sw, double& c, double& s, double& tt) {
        c = 1.0; s = 0.0; tt = 0.0;
        // small-element rule (classical Jacobi): after three sweeps an off-diagonal element that cannot change
        // either diagonal neighbour in fp64 is set to zero instead of being rotated away
        const double g = 100.0 * fabs(apq);
        const bool tiny = sw >= tiny_from && (fabs(app) + g == fabs(app)) && (fabs(aqq) + g == fabs(aqq));
        if (apq != 0.0 && !tiny) {
            // t = tan(theta) of the rotation that annihilates a_pq: the smaller root of t^2 + 2 tau t - 1 = 0 with
            // tau = (a_qq - a_pp) / (2 a_pq), written without tau so that the dependent chain is sqrt, div, sqrt instead
            // of div, sqrt, div, sqrt, div: with d = a_qq - a_pp, h = hypot(d, 2 a_pq), w = |d| + h:
            // t = 2 a_pq / (+-w) (sign of d), c = 1 / sqrt(1 + t^2) = sqrt(w / (2 h)), s = t c.
            const double d = aqq - app, b2 = 2.0 * apq;
            const double h = sqrt(fma(d, d, b2 * b2));
            if (h > 0.0) {   // h == 0: d and a_pq below 1e-154, nothing to rotate (the element is zeroed)
                const double w = fabs(d) + h;
                // sign of t = sign of tau = d / (2 a_pq), +1 at d = 0 exactly (equal diagonal entries are common: every
                // landmark enters P with the same W block; letting the sign follow a_pq there made clusters of equal
                // eigenvalues cycle at the rounding level instead of settling)
                const bool pos = (d == 0.0) || ((d > 0.0) == (b2 > 0.0));
                tt = (pos ? fabs(b2) : -fabs(b2)) / w;
                c = sqrt(w / (2.0 * h));
                s = tt * c;
            }
        }
    };
    constexpr int NSC = kTab ? (NMAX * (NMAX - 1) / 2 + TPB - 1) / TPB : 1;   // strictly-lower elements per thread in the convergence scan
    const int nlow = n * (n - 1) / 2;
    unsigned scan_a[NSC], scan_b[NSC];      // byte offsets: element | A(c, c) << 16, A(r, r)
    auto scan_addresses = [&]() {
#pragma unroll
        for (int u = 0; u < NSC; ++u) {
            const int e = tid + TPB * u < nlow ? tid + TPB * u : 0;
            int r = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);   // row r >= 1 holds the elements r (r - 1) / 2 .. r (r + 1) / 2 - 1
            while (r * (r - 1) / 2 > e) --r;
            while ((r + 1) * r / 2 <= e) ++r;
            const int c = e - r * (r - 1) / 2;
            scan_a[u] = (unsigned)(8 * (r * (r + 1) / 2 + c)) | ((unsigned)(8 * (c * (c + 1) / 2 + c)) << 16);
            scan_b[u] = (unsigned)(8 * (r * (r + 1) / 2 + r));
        }
    };
    if constexpr (kTab) {
        // wavefront 0 carries the longest dependent chain of a round (its items, then the next parameters): let it issue ahead
        if (tid < 64) __builtin_amdgcn_s_setprio(SLAM_UKF_PRIO);
    }
    // pass table: the thread's block lane, fixed for the launch (w7 of any pass): parameter slots 2 I + i | (2 J + j) << 8 | I << 16 | J << 24, -1 = none
    int qrole = -1;
    if constexpr (kTab) {
        const unsigned w7 = qw7;
        if ((w7 & 3u) == 1u) {
            const int I = (w7 >> 8) & 0xff, J = (w7 >> 16) & 0xff;
            qrole = (2 * I + (int)((w7 >> 24) & 1u)) | ((2 * J + (int)((w7 >> 25) & 1u)) << 8) | (I << 16) | (J << 24);
        }
    }
    int vitem[2] = {-1, -1};   // quadruple | pair of columns << 8, -1 = none
    if constexpr (kTab) {
        if (tid >= 64) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int it = tid - 64 + (kUkfRotThreads - 64) * u, q = it / mj, kp = it - q * mj;   // (the table numbers nj / 2 pairs of columns per quadruple)
                if (q < (nj >> 2) && kp < m) vitem[u] = q | (kp << 8);                                 // the columns >= n of V^T are not formed
            }
        }
    }
    // ---- pieces of the quadruple ("pass") paths (launch_ukf_quad_table; jacobi_schedule.h) ----
    constexpr int MQ = (MMAX + 1) / 2;
    // convergence scan of the passes without the table: threads 64 .. walk the strictly-lower triangle, scan_tpr of them per row
    const int scan_tpr = (TPB - 64) / n > 0 ? (TPB - 64) / n : 1, scan_dr = (TPB - 64) / scan_tpr > 0 ? (TPB - 64) / scan_tpr : 1;
    const int scan_r0 = (tid - 64) / scan_tpr, scan_c0 = (tid - 64) - scan_r0 * scan_tpr;
    // passes without the table: the thread's block lanes (block (I, J) of quadruple positions, I > J: I << 8 | J) and V items (quadruple | pair of
    // columns << 8).  As in the table path the "critical" blocks - (1, 0), (k + 1, k - 1), (mq - 1, mq - 2): the ones that hold the next pass's pivots -
    // belong to the first lanes of the workgroup (wavefront 0, and wavefront 1 when there are more than sixteen), the others start at thread QOT0.
    constexpr bool kQuadGen = ITB <= 2 && !kTab;
    constexpr int QOT0 = 4 * MQ > 64 ? 128 : 64;
    constexpr int QNB = kQuadGen ? 1 + (4 * (MQ * (MQ - 1) / 2) + (TPB - QOT0) - 1) / (TPB - QOT0) : 1, QNV = kQuadGen ? (MQ * MMAX + (TPB - VT0) - 1) / (TPB - VT0) : 1;
    int qb_desc[QNB], qv_desc[QNV];
    if constexpr (kQuadGen) {
        const int mq_ = nj >> 2;
#pragma unroll
        for (int ub = 0; ub < QNB; ++ub) qb_desc[ub] = -1;
        if (mq_ >= 2 && tid < 4 * mq_) {   // slot 0 of the first lanes: critical block tid / 4
            const int c = tid >> 2;
            qb_desc[0] = c == 0 ? (1 << 8) : (c <= mq_ - 2 ? (((c + 1) << 8) | (c - 1)) : (mq_ >= 3 ? (((mq_ - 1) << 8) | (mq_ - 2)) : -1));
        }
        if (tid >= QOT0) {
#pragma unroll
            for (int ub = 1; ub < QNB; ++ub) {
                // rank r among the other blocks, lexicographic.  Row I = 2 .. mq - 2 holds I - 1 of them (every J < I but I - 2), row mq - 1 the
                // mq - 3 with J < mq - 3: the rows before the last are a triangle, decoded in closed form
                const int r = ((tid - QOT0) + (TPB - QOT0) * (ub - 1)) >> 2;
                const int ntri = mq_ >= 3 ? (mq_ - 2) * (mq_ - 3) / 2 : 0;
                int d = -1;
                if (r < ntri) {
                    int Ip = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)r)) * 0.5f);
                    while (Ip * (Ip - 1) / 2 > r) --Ip;
                    while ((Ip + 1) * Ip / 2 <= r) ++Ip;
                    const int Jp = r - Ip * (Ip - 1) / 2, I = Ip + 1;
                    d = (I << 8) | (Jp < I - 2 ? Jp : Jp + 1);
                } else if (mq_ >= 4 && r - ntri <= mq_ - 4) {
                    d = ((mq_ - 1) << 8) | (r - ntri);
                }
                qb_desc[ub] = d;
            }
        }
#pragma unroll
        for (int u = 0; u < QNV; ++u) {
            const int it = tid - VT0 + (TPB - VT0) * u, Q = it / m;
            qv_desc[u] = (tid >= VT0 && Q < mq_) ? (Q | ((it - Q * m) << 8)) : -1;
        }
    }
    char* const sAb = reinterpret_cast<char*>(sA);
    char* const sVb = reinterpret_cast<char*>(sVt);
    auto ldA = [&](unsigned off) -> double { return *reinterpret_cast<const double*>(sAb + off); };
    auto stA = [&](unsigned off, double v) { *reinterpret_cast<double*>(sAb + off) = v; };
    // B' = R_i^T B R_j on a pair-block (rows: the pair with the higher index)
    auto rot_block = [&](double& b00, double& b01, double& b10, double& b11, const double2 ri, const double2 rj) {
        // (no test for identity rotations here: the oracle has none either, and a pass in which a whole quadruple rests is skipped as one)
        const double ci = ri.x, si = ri.y, cj = rj.x, sj = rj.y;
        const double t00 = fma(ci, b00, -(si * b10)), t01 = fma(ci, b01, -(si * b11));
        const double t10 = fma(si, b00, ci * b10), t11 = fma(si, b01, ci * b11);
        b00 = fma(t00, cj, -(t01 * sj)); b01 = fma(t00, sj, t01 * cj);
        b10 = fma(t10, cj, -(t11 * sj)); b11 = fma(t10, sj, t11 * cj);
    };
    // Parameter lane 2 q + u of wavefront 0, pair u of quadruple q = blocks (X, Y) of the pass that comes next: the pass's rounds
    // one after the other on the quadruple's diagonal 4 x 4 block (a, b | c, d) = (2X, 2X+1 | 2Y, 2Y+1) IN LDS - parameters from the
    // pair's three elements, the pair's diagonal block, then lane 1 the cross block (rows: pair 1) with both rotations.
    //   round 0 (first pass of a sweep): pairs (a,b) (c,d), cross (c,d) x (a,b);  1: (a,c) (b,d), cross (b,d) x (a,c);  2: (a,d) (b,c), cross (b,c) x (a,d)
    auto param_phase = [&](const unsigned xy, const bool first, const int parw, const int sweep) -> int {
        const int u = tid & 1, q = tid >> 1;
        const unsigned a = 2u * (xy & 0xffu), c = 2u * ((xy >> 8) & 0xffu);
        const unsigned ta = 4u * a * (a + 1u), tb = ta + 8u * (a + 1u), tc = 4u * c * (c + 1u), td = tc + 8u * (c + 1u);   // 8 * row (row + 1) / 2
        const unsigned aa = ta + 8u * a, ba = tb + 8u * a, bb = ba + 8u, ca = tc + 8u * a, cb = ca + 8u, cc = tc + 8u * c;
        const unsigned da = td + 8u * a, db = da + 8u, dc = td + 8u * c, dd = dc + 8u;
        bool any = false;
        auto round_of_pass = [&](const unsigned pp, const unsigned qq, const unsigned pq,
                                 const unsigned x00, const unsigned x01, const unsigned x10, const unsigned x11, const int s) {
            const double app = ldA(pp), aqq = ldA(qq), apq = ldA(pq);
            double cr, sr, tr;
            jacobi_param(app, aqq, apq, sweep, cr, sr, tr);
            s_csn[(parw * 3 + s) * MMAX + tid] = make_double2(cr, sr);
            stA(pp, fma(-tr, apq, app)); stA(qq, fma(tr, apq, aqq));
            if (apq != 0.0) stA(pq, 0.0);
            any = any || sr != 0.0;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // lane 1 reads lane 0's rotation
            if (u == 1) {
                const double2 r0 = s_csn[(parw * 3 + s) * MMAX + tid - 1];
                double b00 = ldA(x00), b01 = ldA(x01), b10 = ldA(x10), b11 = ldA(x11);
                rot_block(b00, b01, b10, b11, make_double2(cr, sr), r0);
                stA(x00, b00); stA(x01, b01); stA(x10, b10); stA(x11, b11);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the next round reads what this one wrote
        };
        if (first) round_of_pass(u ? cc : aa, u ? dd : bb, u ? dc : ba, ca, cb, da, db, 0);
        round_of_pass(u ? bb : aa, u ? dd : cc, u ? db : ca, ba, cb, da, dc, 1);
        round_of_pass(u ? bb : aa, u ? cc : dd, u ? cb : da, ba, db, ca, dc, 2);
        // any rotation of the quadruple in this pass that is not the identity?  (both lanes of the quadruple: lane ^ 1)
        const int mine = any ? 1 : 0;
        const int both = mine | __builtin_amdgcn_mov_dpp(mine, 0xB1, 0xf, 0xf, true);
        if (u == 0) s_qflag[parw * MQ + q] = both;
        return both;
    };
    // V <- V J for a pass: rows a, b, c, d of V^T (quadruple Q = blocks xy), the 16-byte pair kp of columns, through every rotation of the pass
    auto v_item = [&](const int Q, const int kp, const unsigned xy, const bool first, const double2* const cs) {
        char* const ra = sVb + 16 * kp + 16 * n * (int)(xy & 0xffu);
        char* const rc = sVb + 16 * kp + 16 * n * (int)(xy >> 8);
        double2 va = *reinterpret_cast<const double2*>(ra), vb = *reinterpret_cast<const double2*>(ra + 8 * n);
        double2 vc = *reinterpret_cast<const double2*>(rc), vd = *reinterpret_cast<const double2*>(rc + 8 * n);
        auto rot_v = [&](double2& xp, double2& xq, const double2 r) {
            const double c = r.x, sn = r.y;
            const double2 np = make_double2(fma(c, xp.x, -(sn * xq.x)), fma(c, xp.y, -(sn * xq.y)));
            xq = make_double2(fma(sn, xp.x, c * xq.x), fma(sn, xp.y, c * xq.y));
            xp = np;
        };
        if (first) { rot_v(va, vb, cs[2 * Q]); rot_v(vc, vd, cs[2 * Q + 1]); }
        rot_v(va, vc, cs[MMAX + 2 * Q]); rot_v(vb, vd, cs[MMAX + 2 * Q + 1]);
        rot_v(va, vd, cs[2 * MMAX + 2 * Q]); rot_v(vb, vc, cs[2 * MMAX + 2 * Q + 1]);
        *reinterpret_cast<double2*>(ra) = va; *reinterpret_cast<double2*>(ra + 8 * n) = vb;
        *reinterpret_cast<double2*>(rc) = vc; *reinterpret_cast<double2*>(rc + 8 * n) = vd;
    };
#pragma unroll 1
    for (int sweep = 0; sweep < 60; ++sweep) {
        // convergence: every off-diagonal element is exactly zero OR would only be zeroed by the small-element rule
        // below.  A sweep in which every pair is either zero or "tiny" performs no rotation: it leaves V and the diagonal
        // untouched and merely writes the zeros, so it can be skipped without changing sqtP by a single bit (the oracle
        // runs that last sweep and arrives at the same V and diagonal).
        int live = 0;
        if constexpr (kTab) {
            // the thread's (at most four) strictly-lower elements, addresses decoded once per launch: every read of the scan is issued at
            // once (the two-threads-per-row walk below left two thirds of the workgroup idle and each lane 22 dependent round trips)
            const char* const sAc = reinterpret_cast<const char*>(sA);
            double v[NSC], dp[NSC], dq[NSC];
            scan_addresses();   // (once per sweep: kept across the sweep they were eight registers of a kernel that has 80)
#pragma unroll
            for (int u = 0; u < NSC; ++u) {
                v[u] = *reinterpret_cast<const double*>(sAc + (scan_a[u] & 0xffffu));
                dp[u] = *reinterpret_cast<const double*>(sAc + (scan_a[u] >> 16));
                dq[u] = *reinterpret_cast<const double*>(sAc + scan_b[u]);
            }
#pragma unroll
            for (int u = 0; u < NSC; ++u) {
                if (tid + TPB * u < nlow && v[u] != 0.0 && v[u] == v[u]) {   // (a NaN is not "live": the oracle's max() passes over it, see below)
                    const double g = 100.0 * fabs(v[u]);
                    const double app = fabs(dp[u]), aqq = fabs(dq[u]);
                    if (!(sweep >= tiny_from && (app + g == app) && (aqq + g == aqq))) live = 1;
                }
            }
        } else if constexpr (kFast) {
            // passes without the table: wavefront 0 forms the parameters of pass 0 (three rounds on the diagonal 4 x 4 blocks: the longest chain of
            // a sweep, and nothing else of the workgroup can run beside it but this scan) while the other wavefronts scan.  The two touch the
            // same elements - the pivots the parameter lanes zero or rotate, the diagonal entries they update - and the verdict is still
            // exact: a rotation that is not the identity means an element was live at the start of the sweep and is reported by its lane
            // (param_phase returns it); if every rotation is the identity, nothing the scan reads has changed (a_pp' = fma(-0, a_pq, a_pp)).
            const int mq = nj >> 2;
            if (tid < 64) {
                if (tid < 2 * mq) {
                    int X, Y;
                    rr_pair(tid >> 1, 0, mj, X, Y);
                    const unsigned xy = (unsigned)X | ((unsigned)Y << 8);
                    if (!(tid & 1)) s_xy[(par ^ 1) * MQ + (tid >> 1)] = (int)xy;
                    live = param_phase(xy, true, par ^ 1, sweep);
                }
            } else {
                for (int r = scan_r0; r < n; r += scan_dr)   // scan_tpr threads per row of the strictly-lower part (divisions: once per launch)
                    for (int c = scan_c0; c < r; c += scan_tpr) {
                        const double v = sA[r * (r + 1) / 2 + c];
                        if (v != 0.0 && v == v) {   // (NaN: see below)
                            const double g = 100.0 * fabs(v);
                            const double app = fabs(sA[c * (c + 1) / 2 + c]), aqq = fabs(sA[r * (r + 1) / 2 + r]);
                            if (!(sweep >= tiny_from && (app + g == app) && (aqq + g == aqq))) live = 1;
                        }
                    }
            }
        } else
        for (int r = tid / 2; r < n; r += TPB / 2)          // two threads per row, strictly-lower part
            for (int c = (tid & 1); c < r; c += 2) {
                const double v = sA[r * (r + 1) / 2 + c];
                // (NaN: the oracle's convergence test is max |a_ij| == 0 with std::max, which passes over NaNs - an instance whose P went
                // non-finite is flagged SLAM_INST_NONFINITE by the step kernel either way; it must not ALSO end as "no convergence" here only)
                if (v != 0.0 && v == v) {
                    const double g = 100.0 * fabs(v);
                    const double app = fabs(sA[c * (c + 1) / 2 + c]), aqq = fabs(sA[r * (r + 1) / 2 + r]);
                    if (!(sweep >= tiny_from && (app + g == app) && (aqq + g == aqq))) live = 1;
                }
            }
        const int any_live = __syncthreads_or(live);
        SQ_STAMP(1);   // convergence check
        if (!any_live) {
            converged = true;
            if (tid == 0 && p.khist) { atomicAdd(&p.khist[8], (unsigned long long)sweep); atomicAdd(&p.khist[9], 1ull); }
            break;
        }
        if constexpr (kTab) {
            {
                // ======== pass-table path: nj / 2 - 1 passes of two rounds, the in-block round inside pass 0 ========
                const int mq = nj >> 2;
#pragma unroll 1
                for (int T = -1; T < mj - 1; ++T) {   // T = -1: only the parameters of pass 0 (every later pass gets its own a pass ahead)
                    const uint4 ea = te_next;
                    const unsigned ez = qnz;
                    if (T >= 0) {
                        const int Tn = T + 1 < mj - 1 ? T + 1 : 0;
                        te_next = qtab[(size_t)Tn * (2 * kUkfRotThreads)]; qnz = qtab[(size_t)Tn * (2 * kUkfRotThreads) + 1].z;
                        const bool first = T == 0;
                        const double2* const cs = s_csn + par * 3 * MMAX;
                        const int* const qf = s_qflag + par * MQ;
                        if (qrole >= 0) {   // a lane of the 4 x 4 block between quadruples I > J
                            const int I = (qrole >> 16) & 0xff, J = qrole >> 24, si = qrole & 0xff, sj = (qrole >> 8) & 0xff;
                            if (qf[I] | qf[J]) {   // (the same for the four lanes of the block)
                                if (first) {
                                    const uint2 e0 = *reinterpret_cast<const uint2*>(qtab + 1);   // w4 w5 of pass 0: once per sweep, on demand
                                    double g00 = ldA(e0.x & 0xffffu), g01 = ldA(e0.x >> 16), g10 = ldA(e0.y & 0xffffu), g11 = ldA(e0.y >> 16);
                                    rot_block(g00, g01, g10, g11, cs[si], cs[sj]);
                                    stA(e0.x & 0xffffu, g00); stA(e0.x >> 16, g01); stA(e0.y & 0xffffu, g10); stA(e0.y >> 16, g11);
                                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the lanes of the block read each other's results below
                                }
                                double e00 = ldA(ea.x & 0xffffu), e01 = ldA(ea.x >> 16), e10 = ldA(ea.y & 0xffffu), e11 = ldA(ea.y >> 16);
                                rot_block(e00, e01, e10, e11, cs[MMAX + si], cs[MMAX + sj]);
                                // rows (i, i+2) x columns (j, j+2) -> rows (i, 3-i) x columns (j, 3-j): three of the four elements come from the other lanes
                                e01 = dpp_quad<0xB1>(e01); e10 = dpp_quad<0x4E>(e10); e11 = dpp_quad<0x1B>(e11);
                                rot_block(e00, e01, e10, e11, cs[2 * MMAX + si], cs[2 * MMAX + sj]);
                                stA(ea.z & 0xffffu, e00); stA(ea.z >> 16, e01); stA(ea.w & 0xffffu, e10); stA(ea.w >> 16, e11);
                            }
                        }
#pragma unroll 1
                        for (int u = 0; u < 2; ++u) {   // V <- V J: four rows of V^T, a 16-byte pair of columns (one item at a time: registers)
                            const int vi = u ? vitem[1] : vitem[0];
                            if (vi < 0) continue;
                            const int Q = vi & 0xff, kp = vi >> 8;
                            if (!qf[Q]) continue;
                            v_item(Q, kp, (ez >> (16 * u)) & 0xffffu, first, cs);
                        }
                    }
                    // wavefront 0: the next pass's parameters, from what it has just written (its lanes < n / 2 carry their quadruple's blocks in w6)
                    if (tid < 2 * mq && T + 1 < mj - 1) param_phase(T < 0 ? ez : qnz, T < 0, par ^ 1, sweep);
                    __syncthreads();
                    SQ_STAMP(3);   // one pass (one barrier)
                    if constexpr (PROF) { if (tid == 0 && T >= 0) sacc[5] += 2; }   // rounds
                    par ^= 1;
                }
                continue;
            }
        }
        if constexpr (kFast && !kTab) {
            {
                // ======== passes without the table (the other fast variants; L = 50 runs <104, 1024>, one workgroup per CU) ========
                // One barrier per pass, as in the table path: wavefront 0 (and 1) rotate the critical blocks first, then wavefront 0 goes on to the
                // parameters of the NEXT pass (two rounds of sqrt / div / sqrt chains on the diagonal 4 x 4 blocks in LDS: ~1 us, the longest thing in
                // a pass) while the other wavefronts do the rest of the blocks and the V items.  Critical blocks beyond wavefront 0's sixteen are
                // wavefront 1's: it raises s_pass_flag after writing them and wavefront 0 waits for that before it reads the pivots.
                const int mq = nj >> 2;
                const bool crit_in_w1 = 4 * mq > 64;
                par ^= 1;   // the parameters of pass 0 were formed beside the convergence scan
#pragma unroll 1
                for (int T = 0; T < mj - 1; ++T) {
                    {
                        const bool first = T == 0;
                        const double2* const cs = s_csn + par * 3 * MMAX;
                        const int* const qf = s_qflag + par * MQ;
#pragma unroll
                        for (int ub = 0; ub < QNB; ++ub) {
                            const int bd = qb_desc[ub];   // block (I, J), I > J: I << 8 | J
                            if (bd < 0) continue;
                            const int I = bd >> 8, J = bd & 0xff;
                            if (!(qf[I] | qf[J])) continue;   // (the same for the four lanes of the block)
                            const int li = (tid >> 1) & 1, lj = tid & 1, si = 2 * I + li, sj = 2 * J + lj;
                            const unsigned xi = (unsigned)s_xy[par * MQ + I], xj = (unsigned)s_xy[par * MQ + J];
                            // the blocks (two consecutive indices each) behind the lane's rows and columns; an element (r, c) of the packed lower triangle sits
                            // at tri(max) + min, and which of r, c is larger is a property of the two BLOCKS (four comparisons for all twelve elements)
                            const int bX = (int)(xi & 0xffu), bY = (int)(xi >> 8), dX = (int)(xj & 0xffu), dY = (int)(xj >> 8);
                            auto off = [&](const int r, const int c, const bool r_gt_c) -> unsigned {   // byte offset of A(r, c)
                                const int hi = r_gt_c ? r : c, lo = r_gt_c ? c : r;
                                return (unsigned)(4 * hi * (hi + 1) + 8 * lo);
                            };
                            const bool gXX = bX > dX, gXY = bX > dY, gYX = bY > dX, gYY = bY > dY;
                            if (first) {   // the in-block round: rows (2 i, 2 i + 1) x columns (2 j, 2 j + 1) of the block
                                const int r0 = 2 * (li ? bY : bX), c0 = 2 * (lj ? dY : dX);
                                const bool g = li ? (lj ? gYY : gYX) : (lj ? gXY : gXX);
                                const unsigned a00 = off(r0, c0, g), a01 = off(r0, c0 + 1, g), a10 = off(r0 + 1, c0, g), a11 = off(r0 + 1, c0 + 1, g);
                                double b00 = ldA(a00), b01 = ldA(a01), b10 = ldA(a10), b11 = ldA(a11);
                                rot_block(b00, b01, b10, b11, cs[si], cs[sj]);
                                stA(a00, b00); stA(a01, b01); stA(a10, b10); stA(a11, b11);
                                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the lanes of the block read each other's results below
                            }
                            const int rA = 2 * bX + li, rB = 2 * bY + li, cA = 2 * dX + lj, cB = 2 * dY + lj;
                            double e00 = ldA(off(rA, cA, gXX)), e01 = ldA(off(rA, cB, gXY)), e10 = ldA(off(rB, cA, gYX)), e11 = ldA(off(rB, cB, gYY));
                            rot_block(e00, e01, e10, e11, cs[MMAX + si], cs[MMAX + sj]);
                            e01 = dpp_quad<0xB1>(e01); e10 = dpp_quad<0x4E>(e10); e11 = dpp_quad<0x1B>(e11);   // rows (i, i+2) x columns (j, j+2) -> (i, 3-i) x (j, 3-j)
                            rot_block(e00, e01, e10, e11, cs[2 * MMAX + si], cs[2 * MMAX + sj]);
                            const int rC = 2 * bY + 1 - li, cC = 2 * dY + 1 - lj;
                            stA(off(rA, cA, gXX), e00); stA(off(rA, cC, gXY), e01); stA(off(rC, cA, gYX), e10); stA(off(rC, cC, gYY), e11);
                        }
                        if (crit_in_w1 && tid >= 64 && tid < 128) {   // wavefront 1: its critical blocks are written (a wavefront's LDS accesses execute in order)
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (tid == 64) *reinterpret_cast<volatile int*>(&s_pass_flag) = pass_seq;
                        }
                        if (tid >= VT0) {
#pragma unroll
                            for (int u = 0; u < QNV; ++u) {
                                const int vd = qv_desc[u];   // quadruple | pair of columns << 8
                                if (vd < 0) continue;
                                const int Q = vd & 0xff;
                                if (!qf[Q]) continue;
                                v_item(Q, vd >> 8, (unsigned)s_xy[par * MQ + Q], first, cs);
                            }
                        }
                    }
                    if (tid < 64 && T + 1 < mj - 1) {   // wavefront 0: the next pass's parameters, from what the critical lanes have just written
                        if (crit_in_w1) {
                            while (*reinterpret_cast<volatile int*>(&s_pass_flag) != pass_seq) __builtin_amdgcn_s_sleep(1);
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        }
                        if (tid < 2 * mq) {
                            int X, Y;
                            rr_pair(tid >> 1, T + 1, mj, X, Y);
                            const unsigned xy = (unsigned)X | ((unsigned)Y << 8);
                            if (!(tid & 1)) s_xy[(par ^ 1) * MQ + (tid >> 1)] = (int)xy;
                            param_phase(xy, false, par ^ 1, sweep);
                        }
                    }
                    __syncthreads();
                    SQ_STAMP(3);   // one pass (one barrier)
                    if constexpr (PROF) { if (tid == 0) sacc[5] += 2; }   // rounds
                    par ^= 1;
                    pass_seq += 1;
                }
                continue;
            }
        }
        // ======== the variants without passes (more than two pair-block items per thread): round by round, two barriers each ========
        if constexpr (!kFast) {
#pragma unroll 1
        for (int t = 0; t < nj - 1; ++t) {
            if (tid < mj) {  // rotation parameters of this round's pairs (jacobi_schedule.h)
                const int k = tid;
                int pidx, qidx;
                jacobi_pair(k, t, nj, pidx, qidx);
                double c, s, tt;
                jacobi_param(AT(pidx, pidx), AT(qidx, qidx), AT(qidx, pidx), sweep, c, s, tt);
                s_pp[k] = pidx; s_qq[k] = qidx; s_cs[k] = c; s_sn[k] = s; s_tn[k] = tt;
                s_csn[par * MMAX + k] = make_double2(c, s);
            }
            __syncthreads();
            SQ_STAMP(2);   // rotation parameters (lanes of wavefront 0) + barrier
            if constexpr (PROF) { if (tid == 0) sacc[5] += 1; }   // rounds
#pragma unroll
            for (int u = 0; u < IT; ++u) {
                const int d = desc[u];
                if (d < 0) continue;
                const int kind = d >> 16, i = (d >> 8) & 0xff, jk = d & 0xff;
                if (kind == 0) {
                    const int j = jk;
                    const double si = s_sn[i], sj = s_sn[j];
                    if (si == 0.0 && sj == 0.0) continue;   // both rotations are the identity (c = 1 exactly): B' = B bit for bit
                    const int pi = s_pp[i], qi = s_qq[i], pj = s_pp[j], qj = s_qq[j];
                    const double ci = s_cs[i], cj = s_cs[j];
                    double& e00 = AT(pi, pj); double& e01 = AT(pi, qj); double& e10 = AT(qi, pj); double& e11 = AT(qi, qj);
                    const double b00 = e00, b01 = e01, b10 = e10, b11 = e11;
                    const double t00 = fma(ci, b00, -(si * b10)), t01 = fma(ci, b01, -(si * b11));
                    const double t10 = fma(si, b00, ci * b10), t11 = fma(si, b01, ci * b11);
                    e00 = fma(t00, cj, -(t01 * sj)); e01 = fma(t00, sj, t01 * cj);
                    e10 = fma(t10, cj, -(t11 * sj)); e11 = fma(t10, sj, t11 * cj);
                } else if (kind == 1) {
                    const int pq = s_pp[i], qq = s_qq[i];
                    const double app = AT(pq, pq), aqq = AT(qq, qq), apq = AT(qq, pq);
                    AT(pq, pq) = fma(-s_tn[i], apq, app);
                    AT(qq, qq) = fma(s_tn[i], apq, aqq);
                    if (apq != 0.0) AT(qq, pq) = 0.0;
                } else {
                    const int k = jk;
                    const double s = s_sn[i];
                    if (s == 0.0) continue;                 // identity rotation: the V row pair is unchanged
                    const int pq = s_pp[i], qq = s_qq[i];
                    const double c = s_cs[i];
                    const double vp = sVt[pq * n + k], vq = sVt[qq * n + k];
                    sVt[pq * n + k] = fma(c, vp, -(s * vq));
                    sVt[qq * n + k] = fma(s, vp, c * vq);
                }
            }
            __syncthreads();
            SQ_STAMP(3);   // rotation phase + barrier
        }
        }
    }
    if (!converged) {
        // ukf.cpp:209-211 swallows the exception and reuses the stale sqtP.  A stale matrix of another size cannot be
        // used: zero it.  Flag the instance either way.
        if (p.n_sq[b] != n)
            for (int e = tid; e < n * n; e += TPB) Sq[e] = 0.0;
        if (tid == 0) { p.flags[b] = p.flags[b] | SLAM_INST_SQRT_FAILED; p.n_sq[b] = n; p.v_age[b] = -1; }
        return;
    }
    {   // keep V^T for the next timestep's warm start
        for (int e = tid; e < n * n; e += TPB) Vs[e] = sVt[e];
        if (tid == 0) p.v_age[b] = warm ? age + 1 : 0;
    }
    for (int k = tid; k < n; k += TPB) {
        const double d = sA[k * (k + 1) / 2 + k];
        s_sd[k] = sqrt(d > 0.00000001 ? d : 0.00000001);   // cwiseMax(1e-8), then the principal square root
    }
    __syncthreads();
    {   // sqtP = (V sqrt(D)) V^T on v_mfma_f64_16x16x4_f64 (round 4; as n^3 / 2 scalar multiply-adds out of LDS it was bound by LDS bandwidth: two
        // reads per product): one 16 x 16 tile of the lower triangle per wavefront at a time, k in steps of four; per element the chain is
        // acc = fma(V(r, k) sd_k, V(c, k), acc) in ascending k, which is what the oracle evaluates; rows, columns and k beyond n give zero operands
        const int nt = (n + 15) >> 4, nk = (n + 3) >> 2;
        const int wv = tid >> 6, ln = tid & 63, kq = ln >> 4, cl = ln & 15;
#pragma unroll 1
        for (int t = wv; t < nt * nt; t += TPB / 64) {
            const int tr = t / nt, tc = t - tr * nt;
            if (tc > tr) continue;                              // (wave-uniform)
            const int ar = 16 * tr + cl, bc = 16 * tc + cl;     // A-operand: row ar of V sqrt(D); B-operand: column bc of V^T
            const bool va = ar < n, vb = bc < n;
            const int arc = va ? ar : 0, bcc = vb ? bc : 0;
            dbl4_t acc = dbl4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
            for (int ks = 0; ks < nk; ++ks) {
                const int k = 4 * ks + kq;
                const bool vk = k < n;
                const int kc = vk ? k : 0;
                double a = sVt[kc * n + arc] * s_sd[kc];        // V(ar, k) sd_k   (row k of V^T = eigenvector k)
                double bv = sVt[kc * n + bcc];                  // V(bc, k)
                a = (va && vk) ? a : 0.0;
                bv = (vb && vk) ? bv : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {   // C/D layout: row = (lane >> 4) + 4 * reg, column = lane & 15
                const int r = 16 * tr + kq + 4 * r4, c = 16 * tc + cl;
                if (r < n && c <= r) { Sq[(size_t)r * n + c] = acc[r4]; Sq[(size_t)c * n + r] = acc[r4]; }
            }
        }
    }
    if (tid == 0) p.n_sq[b] = n;
    SQ_STAMP(4);   // V^T store, sqrt(D), sqtP = V sqrt(D) V^T
    if constexpr (PROF) {
        if (p.prof && tid == 0)
            for (int i = 0; i < 6; ++i) p.prof[(size_t)b * 16 + 10 + i] = sacc[i];
    }
#undef SQ_STAMP
}

// ------------------------------------------------------------------------------------------------------------------
// prediction + update
// ------------------------------------------------------------------------------------------------------------------
// PROF: phase timers compiled in (own instantiation, launched only with the debug buffer attached; as a run-time option
// the ten accumulators cost every variant 22 VGPRs of the production kernel).
#ifndef SLAM_UKF_STEP_WG
#define SLAM_UKF_STEP_WG 6   // workgroups of ukf_step_kernel<44, 128, 3> the compiler must leave room for on a CU (26 KB of LDS, 155 VGPRs: six)
#endif
template <int NMAX, int TPB, int KU, bool PROF = false>
__global__ __launch_bounds__(TPB, (NMAX == 44 && TPB == 128 && !PROF) ? SLAM_UKF_STEP_WG : 1) void ukf_step_kernel(const UkfStepParams p) {
    constexpr int LDN = NMAX + 2;
    constexpr int NS = 2 * NMAX + 1;
    constexpr int LMAX = (NMAX - 4) / 2;
    constexpr int KCAP = LMAX > 0 ? LMAX : 1;

    __shared__ double sS[NMAX * NMAX];        // sqtP (symmetric), row-major with the CURRENT n as leading dimension
    __shared__ double sX4[4 * NS];            // rows 0..3 of X_pred
    __shared__ double sZ0[NS], sD1[NS];       // range estimates; wrapped bearing differences
    __shared__ double s_xt[LDN], s_xp0[LDN], s_xp[LDN];
    __shared__ double sK[KU * NMAX * 4];      // per update: K[r][0..1], (K S)[r][0..1]
    __shared__ double s_sc[16];
    __shared__ float s_meas[3 * KCAP];
    __shared__ int s_ids[LMAX > 0 ? LMAX : 1];
    __shared__ int s_upd[KCAP], s_ins[KCAP];  // detections to update (landmark slot) / to insert (detection index)
    __shared__ int s_misc[8];                 // k, n_upd, n_ins, capacity, singular

    const int b = blockIdx.x + p.b_off, tid = threadIdx.x, lane = tid & 63;
    if (p.long_mode == 3) {   // (workgroup-uniform) a message this size class cannot hold: the streamed kernel's launch takes the instance (ukf_kernel.h)
        const int kk = p.meas_count_in[b];
        if ((kk < p.k_stride_in ? kk : p.k_stride_in) > p.long_cap) return;
    }
    unsigned long long tacc[PROF ? 10 : 1] = {0}, tprev = 0ull;
    if constexpr (PROF) tprev = wall_clock64();
#define UKF_STAMP(i) do { if constexpr (PROF) { if (tid == 0) { const unsigned long long now_ = wall_clock64(); tacc[i] += now_ - tprev; tprev = now_; } } } while (0)
    int flags = p.flags[b];
    const int M_old = p.M[b];
    const int n = 4 + 2 * M_old, ns = 2 * n + 1;
    double* __restrict__ Pout = p.P_out + (size_t)b * p.pstride;
    double* __restrict__ xb = p.x + (size_t)b * p.xstride;
    const double* __restrict__ Sq = p.sqtP + (size_t)b * p.pstride;

    // prologue loads
    for (int i = tid; i < LDN; i += TPB) {
        const double v = i < n ? xb[i] : 0.0;
        s_xt[i] = v;
        if (i < n && p.x_prev) p.x_prev[(size_t)b * p.xstride + i] = v;   // centre of this step's sigma points (ukf.cpp:214)
    }
    for (int i = tid; i < M_old; i += TPB) s_ids[i] = p.ids[(size_t)b * p.L_max + i];
    if (tid < 8) s_misc[tid] = 0;
    double tx = 0.0, ty = 0.0, tth = 0.0, lmx = 0.0, lmy = 0.0;
    if (p.sim && tid < 64) {
        tx = p.truth[3 * (size_t)b]; ty = p.truth[3 * (size_t)b + 1]; tth = p.truth[3 * (size_t)b + 2];
        if (tid < p.L) { lmx = p.map[2 * tid]; lmy = p.map[2 * tid + 1]; }
    }
    for (int e = tid; e < n * n; e += TPB) sS[e] = Sq[e];
    __syncthreads();
    UKF_STAMP(0);

    // ---- measurements ----
    if (p.sim) {
        if (tid < 64) {
            const int cnt = sim_wave<KCAP>(p, b, lane, p.fwd, p.ang, p.step, tx, ty, tth, lmx, lmy, s_meas);
            if (lane == 0) s_misc[0] = cnt;
        }
    } else {
        int kk = p.meas_count_in[b];
        kk = kk < p.k_stride_in ? kk : p.k_stride_in;
        kk = kk < 0 ? 0 : kk;
        const int kc = kk < KCAP ? kk : KCAP;
        for (int i = tid; i < 3 * kc; i += TPB) s_meas[i] = p.meas_in[(size_t)b * p.k_stride_in * 3 + i];
        if (tid == 0) s_misc[0] = kk;
    }
    __syncthreads();
    UKF_STAMP(1);
    if (s_misc[0] > KCAP) flags |= SLAM_INST_CAPACITY;
    const int k = s_misc[0] < KCAP ? s_misc[0] : KCAP;
    if (tid == 0 && p.khist) atomicAdd(&p.khist[k < 7 ? k : 7], 1ull);
    if (p.sim && p.meas_out != nullptr) {
        for (int i = tid; i < 3 * k && i < 3 * p.k_stride_out; i += TPB) p.meas_out[(size_t)b * p.k_stride_out * 3 + i] = s_meas[i];
        if (tid == 0) p.meas_count_out[b] = k;
    }

    // ---- association (ukf.cpp:256-277): known landmarks are updated first, unknown ids inserted afterwards ----
    if (tid < 64) {
        int found = -1;
        bool valid = lane < k;
        const int myid = valid ? (int)s_meas[3 * lane] : -1;
#pragma unroll 1
        for (int l = 0; l < k; ++l) {
            const int id = (int)s_meas[3 * l];
            int f = -1;
#pragma unroll 1
            for (int j0 = 0; j0 < M_old && f < 0; j0 += 64) {
                const int j = j0 + lane;
                const unsigned long long mm = __ballot(j < M_old && s_ids[j] == id);
                if (mm != 0ull) f = j0 + (__ffsll((long long)mm) - 1);
            }
            if (lane == l) found = p.loc ? ((id >= 0 && id < p.L) ? id : -2) : f;
        }
        (void)myid;
        if (__ballot(valid && found == -2) != 0ull && lane == 0) s_misc[5] = 1;   // LOC: id outside the known map
        const unsigned long long um = __ballot(valid && found >= 0);
        const unsigned long long im = __ballot(valid && found == -1);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (valid && found >= 0) s_upd[__popcll(um & below)] = (found << 8) | lane;   // slot, detection index
        if (valid && found == -1) s_ins[__popcll(im & below)] = lane;
        if (lane == 0) { s_misc[1] = __popcll(um); s_misc[2] = __popcll(im); }
    }

    // ---- sigma points through the motion model (ukf.cpp:214-226,125-135); only rows 0..3 change ----
    const float u_d = p.fwd, u_th = p.ang;
    const float dd = u_d + p.v_d;
    for (int i = tid; i < ns; i += TPB) {
        double v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (i == 0) v[r] = s_xt[r];
            else if (i <= n) v[r] = s_xt[r] + sS[(i - 1) * n + r];
            else v[r] = s_xt[r] - sS[(i - 1 - n) * n + r];
        }
        const float yaw = yaw_of(v[2], v[3]);
        double sy, cy;
        tsincos(yaw, p.float_trig, &sy, &cy);
        if (p.float_trig) {
            sX4[0 * ns + i] = v[0] + (double)(dd * (float)cy);   // float * float (ukf.cpp:129)
            sX4[1 * ns + i] = v[1] + (double)(dd * (float)sy);
        } else {
            sX4[0 * ns + i] = v[0] + (double)dd * cy;
            sX4[1 * ns + i] = v[1] + (double)dd * sy;
        }
        const float new_yaw = (float)remainder((double)(yaw + u_th + p.v_th), kTwoPi);   // float adds (ukf.cpp:131)
        double sn, cn;
        tsincos(new_yaw, p.float_trig, &sn, &cn);
        sX4[2 * ns + i] = cn;
        sX4[3 * ns + i] = sn;
    }
    __syncthreads();
    UKF_STAMP(2);
    const int n_upd = s_misc[1], n_insq = s_misc[2];

    const double w0 = (double)kW0;
    const double wi = (double)((1 - kW0) / (2 * n));   // float arithmetic (ukf.cpp:174-175)
    auto xpred_elem = [&](int r, int i) -> double {    // X_pred(r, i)
        if (r < 4) return sX4[r * ns + i];
        if (i == 0) return s_xt[r];
        if (i <= n) return s_xt[r] + sS[(i - 1) * n + r];
        return s_xt[r] - sS[(i - 1 - n) * n + r];
    };

    // ---- weighted mean (ukf.cpp:228-232), sequential in i ----
    // Same walk as the covariance pass below: the three uniform ranges of i in order (identical terms, identical order),
    // x_t[r] hoisted, both candidate operands read unconditionally and selected, unrolled so the LDS reads run ahead of
    // the serial chain of additions instead of one round trip per sigma point.
    for (int r = tid; r < n; r += TPB) {
        const bool pose = r < 4;
        const double xt = s_xt[r];
        const double* X4 = sX4 + (pose ? r : 0) * ns;
        double acc = 0.0;
        acc = acc + w0 * (pose ? X4[0] : xt);
#pragma unroll 4
        for (int i = 1; i <= n; ++i) {
            const double x4 = X4[i], sv = sS[(i - 1) * n + r];
            acc = acc + wi * (pose ? x4 : xt + sv);
        }
#pragma unroll 4
        for (int i = n + 1; i < ns; ++i) {
            const double x4 = X4[i], sv = sS[(i - 1 - n) * n + r];
            acc = acc + wi * (pose ? x4 : xt - sv);
        }
        s_xp0[r] = acc;
        s_xp[r] = acc;
    }
    if (tid == 0) {  // process noise diagonal (ukf.cpp:182-186) and the sensing-model yaw (ukf.cpp:139), both from x_t
        const float yaw = yaw_of(s_xt[2], s_xt[3]);
        double sy, cy;
        tsincos(yaw, p.float_trig, &sy, &cy);
        s_sc[0] = p.V00 * cy; s_sc[1] = p.V00 * sy; s_sc[2] = p.V11 * cy; s_sc[3] = p.V11 * sy;
        s_sc[4] = (double)yaw;
    }
    __syncthreads();
    UKF_STAMP(3);

    // ---- landmark updates (ukf.cpp:293-349); K and S never depend on P (sigma points are not redrawn) ----
    const int nfin_ins = (M_old + n_insq <= p.L_max && M_old + n_insq <= LMAX) ? n_insq : ((p.L_max < LMAX ? p.L_max : LMAX) - M_old);
    if (nfin_ins < n_insq) flags |= SLAM_INST_CAPACITY;
    const int n_fin = n + 2 * nfin_ins;
    int done = 0;        // updates already applied to Pout by earlier passes
    bool first_pass = true;
    while (first_pass || done < n_upd) {
        const int ug = (n_upd - done) < KU ? (n_upd - done) : KU;
#pragma unroll 1
        for (int u = 0; u < ug; ++u) {
            const int packed = s_upd[done + u];
            const int li = 2 * (packed >> 8) + 4, l = packed & 0xff;
            const float r_m = s_meas[3 * l + 1], b_m = s_meas[3 * l + 2];
            const double yaw_s = s_sc[4];
            const double mx = p.loc ? (double)p.mapf[3 * (packed >> 8) + 1] : 0.0, my = p.loc ? (double)p.mapf[3 * (packed >> 8) + 2] : 0.0;
            for (int i = tid; i < ns; i += TPB) {
                const double dx = (p.loc ? mx : xpred_elem(li, i)) - xpred_elem(0, i), dy = (p.loc ? my : xpred_elem(li + 1, i)) - xpred_elem(1, i);
                sZ0[i] = sqrt(dx * dx + dy * dy) + (double)p.w_r;
                // the sensing model's yaw comes from x_t (quirk D-9, ukf.cpp:139) unless switched to the sigma point's own rows 2, 3
                const double yaw_i = p.yaw_sigma ? (double)yaw_of(xpred_elem(2, i), xpred_elem(3, i)) : yaw_s;
                const double z1 = remainder((det_atan2(dy, dx) - yaw_i) + (double)p.w_b, kTwoPi);
                sD1[i] = p.acc_zest1 ? z1 : remainder(z1 - 0.0, kTwoPi);   // z_est(1) stays 0 (quirk D-8, ukf.cpp:310-314); switched off: the leader subtracts it below
            }
            __syncthreads();
        UKF_STAMP(4);
            if (tid == 0) {  // leader: z_est(0), S (sequential in i), S^-1, innovation
                // The sums are sequential in i (the reference's order), the LDS reads feeding them are not: unrolled by
                // eight, the reads of a group issue together and the additions follow in order.  One iteration at a time
                // this single lane paid 2 x 89 LDS round trips per detection (10.7 us against ~1 us of dependent additions).
                double z0 = 0.0;
                z0 = z0 + w0 * sZ0[0];
#pragma unroll 8
                for (int i = 1; i < ns; ++i) z0 = z0 + wi * sZ0[i];
                double zb = 0.0;
                if (p.acc_zest1) {   // quirk D-8 switched off (not the reference): z_est(1) = the weighted mean of the bearings; the deviations follow
                    zb = zb + w0 * sD1[0];
                    for (int i = 1; i < ns; ++i) zb = zb + wi * sD1[i];
                    for (int i = 0; i < ns; ++i) sD1[i] = remainder(sD1[i] - zb, kTwoPi);
                }
                double S[4] = {0.0, 0.0, 0.0, 0.0}, Si[4];
                {   // i = 0 carries w0
                    const double d0 = sZ0[0] - z0, d1 = sD1[0];
                    const double a0 = w0 * d0, a1 = w0 * d1;
                    S[0] = S[0] + a0 * d0; S[1] = S[1] + a0 * d1; S[2] = S[2] + a1 * d0; S[3] = S[3] + a1 * d1;
                }
#pragma unroll 8
                for (int i = 1; i < ns; ++i) {
                    const double d0 = sZ0[i] - z0, d1 = sD1[i];
                    const double a0 = wi * d0, a1 = wi * d1;
                    S[0] = S[0] + a0 * d0; S[1] = S[1] + a0 * d1; S[2] = S[2] + a1 * d0; S[3] = S[3] + a1 * d1;
                }
                S[0] = S[0] + p.W00; S[1] = S[1] + 0.0; S[2] = S[2] + 0.0; S[3] = S[3] + p.W11;
                if (!inv2x2_lu_ukf(S, Si)) s_misc[4] = 1;
                s_sc[5] = z0;
                s_sc[6] = S[0]; s_sc[7] = S[1]; s_sc[8] = S[2]; s_sc[9] = S[3];
                s_sc[10] = Si[0]; s_sc[11] = Si[1]; s_sc[12] = Si[2]; s_sc[13] = Si[3];
                s_sc[14] = (double)r_m - z0;
                s_sc[15] = remainder((double)b_m - zb, kTwoPi);
            }
            __syncthreads();
        UKF_STAMP(5);
            {
                const double z0 = s_sc[5];
                double* Ku = sK + (size_t)u * NMAX * 4;
                for (int r = tid; r < n; r += TPB) {
                    const double xr = s_xp[r];   // CURRENT x_pred (ukf.cpp:330)
                    double c0 = 0.0, c1 = 0.0;
                    // X_pred(r, i) as in the weighted mean: three uniform ranges of i in order, x_t[r] hoisted, both
                    // candidate operands read and selected, unrolled (same terms, same order)
                    const bool pose = r < 4;
                    const double xt = s_xt[r];
                    const double* X4 = sX4 + (pose ? r : 0) * ns;
                    {
                        const double wd = w0 * ((pose ? X4[0] : xt) - xr);
                        c0 = c0 + wd * (sZ0[0] - z0);
                        c1 = c1 + wd * sD1[0];
                    }
#pragma unroll 4
                    for (int i = 1; i <= n; ++i) {
                        const double x4 = X4[i], sv = sS[(i - 1) * n + r];
                        const double wd = wi * ((pose ? x4 : xt + sv) - xr);
                        c0 = c0 + wd * (sZ0[i] - z0);
                        c1 = c1 + wd * sD1[i];
                    }
#pragma unroll 4
                    for (int i = n + 1; i < ns; ++i) {
                        const double x4 = X4[i], sv = sS[(i - 1 - n) * n + r];
                        const double wd = wi * ((pose ? x4 : xt - sv) - xr);
                        c0 = c0 + wd * (sZ0[i] - z0);
                        c1 = c1 + wd * sD1[i];
                    }
                    const double k0 = c0 * s_sc[10] + c1 * s_sc[12];
                    const double k1 = c0 * s_sc[11] + c1 * s_sc[13];
                    Ku[4 * r + 0] = k0; Ku[4 * r + 1] = k1;
                    Ku[4 * r + 2] = k0 * s_sc[6] + k1 * s_sc[8];    // (K S)[r][0]
                    Ku[4 * r + 3] = k0 * s_sc[7] + k1 * s_sc[9];    // (K S)[r][1]
                    s_xp[r] = xr + (k0 * s_sc[14] + k1 * s_sc[15]);
                }
            }
            __syncthreads();
        UKF_STAMP(6);
        }

        // ---- P pass: P_pred = sum_i (w_i d_r) d_c + Q (ukf.cpp:235-240), minus the K S K^T terms of this group's updates.
        //      First pass: the n x (2n+1) x n contraction runs on the matrix pipe.  v_mfma_f64_16x16x4_f64 computes, for every
        //      output element, the chain acc = fma(a_k, b_k, acc) over its four k in ascending order (tools/ubench_mfma_f64.hip
        //      checks that bit for bit), so chaining the instruction over k-blocks 0, 1, ... IS the reference's sequential sum over
        //      the sigma points with each term fused - which is what the oracle evaluates (std::fma).  A operand = w_i d_r(i)
        //      (rounded product, Eigen's `Wts(i) * d` first), B operand = d_c(i); a wavefront owns a row of 16 x 16 tiles, builds
        //      the d operands of every 16-row block once per k-block and feeds all tiles of its row from them.  Padding rows /
        //      sigma points give zero operands (fma(0, 0, acc) = acc).
        if (first_pass) {
            constexpr int NT = (NMAX + 15) / 16;
            const int nt = (n + 15) >> 4;
            const int wv = tid >> 6, kq = lane >> 4, cl = lane & 15;
            const int nks = (ns + 3) >> 2;
            unsigned hiacc = 0u;
            // per 16-row block t: this lane's state row, its x_t and x_pred entries
            int rr[NT];
            bool vr[NT], pr[NT];
            double bt[NT], xm[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int row = 16 * t + cl;
                vr[t] = row < n; rr[t] = vr[t] ? row : 0; pr[t] = rr[t] < 4;
                bt[t] = s_xt[rr[t]]; xm[t] = vr[t] ? s_xp0[rr[t]] : 0.0;
            }
#pragma unroll 1
            for (int tr = wv; tr < nt; tr += TPB / 64) {
                dbl4_t acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = dbl4_t{0.0, 0.0, 0.0, 0.0};
                // operands of k-block ks: d[t] for every 16-row block t, a = w * d[tr]
                auto operands = [&](int ks, double (&d)[NT], double& a) {
                    const int k = 4 * ks + kq;                      // sigma point of this lane's operands
                    const bool vk = k < ns;
                    const int kc = vk ? k : 0;
                    const bool plus = kc <= n;                      // 1..n: x_t + column k-1 of sqtP; n+1..2n: x_t - column k-1-n
                    int kk = plus ? kc - 1 : kc - 1 - n;
                    kk = kk < 0 ? 0 : kk;
                    const double* Srow = sS + (size_t)kk * n;
                    const double w = kc == 0 ? w0 : wi;
                    // branch-free: both candidate operands are read unconditionally and selected (blocks beyond the state size
                    // read row 0 and yield zeros), so the LDS reads of a k-block issue together
                    double x4[NT], sv[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        x4[t] = sX4[(pr[t] ? rr[t] : 0) * ns + kc]; sv[t] = Srow[rr[t]];
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {   // (the asm keeps the compiler from sinking the loads into per-lane branches)
                        asm volatile("" : "+v"(x4[t]), "+v"(sv[t]));
                        const double lp = bt[t] + sv[t], lm = bt[t] - sv[t];
                        const double lv = kc == 0 ? bt[t] : (plus ? lp : lm);
                        const double dv = (pr[t] ? x4[t] : lv) - xm[t];
                        d[t] = (vr[t] && vk) ? dv : 0.0;
                    }
                    double av = d[0];
#pragma unroll
                    for (int t = 1; t < NT; ++t) av = (t == tr) ? d[t] : av;
                    a = w * av;
                };
                // software pipeline: the operands of k-block ks + 1 are formed while the matrix pipe works on k-block ks
                double dc[NT], ac;
                operands(0, dc, ac);
#pragma unroll 1
                for (int ks = 0; ks < nks; ++ks) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac, dc[t], acc[t], 0, 0, 0);
                    double dn[NT], an;
                    operands(ks + 1 < nks ? ks + 1 : ks, dn, an);
#pragma unroll
                    for (int t = 0; t < NT; ++t) dc[t] = dn[t];
                    ac = an;
                }
                UKF_STAMP(9);   // contraction of this tile row
                // C/D layout of the f64 MFMA: row = (lane >> 4) + 4 * reg, column = lane & 15
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (t >= nt) continue;
                    const int c = 16 * t + cl;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * tr + kq + 4 * i;
                        if (r >= n || c >= n) continue;
                        double v = acc[t][i];
                        if (r == c && r < 4) v = v + s_sc[r];   // + Q (signed diagonal)
#pragma unroll 1
                        for (int u = 0; u < ug; ++u) {
                            const double* Ku = sK + (size_t)u * NMAX * 4;
                            v = v - (Ku[4 * r + 2] * Ku[4 * c + 0] + Ku[4 * r + 3] * Ku[4 * c + 1]);
                        }
                        Pout[(size_t)r * n_fin + c] = v;
                        const unsigned h = (unsigned)(__double_as_longlong(v) >> 32) & 0x7fffffffu;
                        hiacc = hiacc > h ? hiacc : h;
                    }
                }
            }
            if (__syncthreads_or(hiacc >= 0x7ff00000u)) flags |= SLAM_INST_NONFINITE;
        } else {
            // later passes (more than KU updates in one step): P comes back from HBM, 4 x 4 register tiles
            constexpr int TR = 4;
            const int ntr = (n + TR - 1) / TR, ntc = (n + 3) / 4;
            unsigned hiacc = 0u;
            for (int tile = tid; tile < ntr * ntc; tile += TPB) {
                const int r0 = TR * (tile / ntc), c0 = 4 * (tile % ntc);
                double acc[TR][4];
#pragma unroll
                for (int a = 0; a < TR; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[a][c] = (r0 + a < n && c0 + c < n) ? Pout[(size_t)(r0 + a) * n_fin + c0 + c] : 0.0;
#pragma unroll 1
                for (int u = 0; u < ug; ++u) {
                    const double* Ku = sK + (size_t)u * NMAX * 4;
#pragma unroll
                    for (int a = 0; a < TR; ++a) {
                        if (r0 + a >= n) continue;
                        const double ks0 = Ku[4 * (r0 + a) + 2], ks1 = Ku[4 * (r0 + a) + 3];
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (c0 + c < n) acc[a][c] = acc[a][c] - (ks0 * Ku[4 * (c0 + c) + 0] + ks1 * Ku[4 * (c0 + c) + 1]);
                    }
                }
#pragma unroll
                for (int a = 0; a < TR; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (r0 + a < n && c0 + c < n) {
                            Pout[(size_t)(r0 + a) * n_fin + c0 + c] = acc[a][c];
                            const unsigned h = (unsigned)(__double_as_longlong(acc[a][c]) >> 32) & 0x7fffffffu;
                            hiacc = hiacc > h ? hiacc : h;
                        }
            }
            if (__syncthreads_or(hiacc >= 0x7ff00000u)) flags |= SLAM_INST_NONFINITE;
        }
        done += ug;
        first_pass = false;
    }

    // ---- landmark insertions (ukf.cpp:351-372): P = blkdiag(P_pred, W) per new landmark, x_pred grows ----
    if (tid == 0) {
        for (int q = 0; q < nfin_ins; ++q) {
            const int l = s_ins[q];
            const float r_m = s_meas[3 * l + 1], b_m = s_meas[3 * l + 2];
            const int nn = n + 2 * q;
            const float yaw = yaw_of(s_xp[2], s_xp[3]);
            const float ang = yaw + b_m;
            double sa, ca;
            tsincos(ang, p.float_trig, &sa, &ca);
            if (p.float_trig) {
                s_xp[nn] = s_xp[0] + (double)(r_m * (float)ca);
                s_xp[nn + 1] = s_xp[1] + (double)(r_m * (float)sa);
            } else {
                s_xp[nn] = s_xp[0] + (double)r_m * ca;
                s_xp[nn + 1] = s_xp[1] + (double)r_m * sa;
            }
            s_ids[M_old + q] = (int)s_meas[3 * l];
        }
    }
    for (int e = tid; e < n_fin * n_fin; e += TPB) {   // new rows / cols: zeros, W on the diagonal
        const int r = e / n_fin, c = e - r * n_fin;
        if (r >= n || c >= n) Pout[e] = (r == c) ? (((r - n) & 1) ? p.W11 : p.W00) : 0.0;
    }
    __syncthreads();
    UKF_STAMP(7);

    // ---- x_t = x_pred (ukf.cpp:289) and bookkeeping ----
    unsigned hx = 0u;
    for (int i = tid; i < n_fin; i += TPB) {
        const double v = s_xp[i];
        xb[i] = v;
        const unsigned h = (unsigned)(__double_as_longlong(v) >> 32) & 0x7fffffffu;
        hx = hx > h ? hx : h;
    }
    if (__syncthreads_or(hx >= 0x7ff00000u)) flags |= SLAM_INST_NONFINITE;
    if (s_misc[4]) flags |= SLAM_INST_S_SINGULAR;
    if (s_misc[5]) flags |= SLAM_INST_INDEX_OOR;
    const int M_new = M_old + nfin_ins;
    if (M_new != M_old)
        for (int i = tid; i < M_new; i += TPB) p.ids[(size_t)b * p.L_max + i] = s_ids[i];
    if (tid == 0) {
        p.M[b] = M_new;
        p.flags[b] = flags | (p.flags[b] & SLAM_INST_SQRT_FAILED);
        p.timestep[b] = p.timestep[b] + 1;
        if (p.sim) {
            const double ex = (double)(float)s_xp[0] - tx, ey = (double)(float)s_xp[1] - ty;
            p.err_sum[b] = p.err_sum[b] + sqrt(ex * ex + ey * ey);
        }
    }
    UKF_STAMP(8);
    if constexpr (PROF) {
        if (p.prof && tid == 0)
            for (int i = 0; i < 10; ++i) p.prof[(size_t)b * 16 + i] = tacc[i];
    }
#undef UKF_STAMP
}

// Tuning: threads per instance.  At BASELINE's batch of 4096 only 16 instances share a CU, so wide workgroups win
// (measured: n<=44 256 threads, n<=104 1024 threads); env SLAM_UKF_TPB = <sqrt threads>*10000 + <step threads> overrides (tools/gpu_ukf_time.py).
static int env_tpb(int which, int dflt) {
    const char* e = getenv("SLAM_UKF_TPB");
    if (!e) return dflt;
    const int v = atoi(e);
    const int t = which == 0 ? v / 10000 : v % 10000;
    return t > 0 ? t : dflt;
}

hipError_t launch_ukf_sqrt(const UkfStepParams& p, hipStream_t stream) {
    const int nmax = 4 + 2 * p.L_max;
    if (nmax <= 44) {
        switch (env_tpb(0, 256)) {
            case 128: hipLaunchKernelGGL((ukf_sqrt_kernel<44, 128>), dim3(p.b_cnt), dim3(128), 0, stream, p); break;
            case 64: hipLaunchKernelGGL((ukf_sqrt_kernel<44, 64>), dim3(p.b_cnt), dim3(64), 0, stream, p); break;
            default:
                if (p.quad_tab == nullptr) return hipErrorInvalidValue;   // <44, 256> takes its operand addresses from the pass table
                if (p.prof) hipLaunchKernelGGL((ukf_sqrt_kernel<44, 256, true>), dim3(p.b_cnt), dim3(256), 0, stream, p);
                else hipLaunchKernelGGL((ukf_sqrt_kernel<44, 256>), dim3(p.b_cnt), dim3(256), 0, stream, p);
                break;
        }
    } else if (nmax <= 104) {
        switch (env_tpb(0, 1024)) {
            case 512: hipLaunchKernelGGL((ukf_sqrt_kernel<104, 512>), dim3(p.b_cnt), dim3(512), 0, stream, p); break;
            case 256: hipLaunchKernelGGL((ukf_sqrt_kernel<104, 256>), dim3(p.b_cnt), dim3(256), 0, stream, p); break;
            default:
                if (p.prof) hipLaunchKernelGGL((ukf_sqrt_kernel<104, 1024, true>), dim3(p.b_cnt), dim3(1024), 0, stream, p);
                else hipLaunchKernelGGL((ukf_sqrt_kernel<104, 1024>), dim3(p.b_cnt), dim3(1024), 0, stream, p);
                break;
        }
    } else {
        return launch_ukf_big_sqrt(p, stream);   // every n x n object in HBM / L2 (ukf_big_kernel.hip)
    }
    return hipGetLastError();
}

hipError_t launch_ukf_step(const UkfStepParams& p, hipStream_t stream) {
    // The size class also sets how many detections of one message the kernel holds (20 or 50).  UKF_LOC has a 4-state filter but
    // sees a map of any size: with more than 20 landmarks it takes the large class (found by tools/gpu_soak_ekf.py: SLAM_INST_CAPACITY
    // and dropped detections on a 35-landmark map where the reference - ukf.cpp:146-154 - uses every one).
    const int nmax = (p.loc && p.L > 20) ? 104 : 4 + 2 * p.L_max;
    if (p.long_mode == 1 && nmax <= 104) {   // a message may exceed what the class holds (ukf_kernel.h)
        if (p.sim) return launch_ukf_big_step(p, stream);
        UkfStepParams q = p;
        q.long_mode = 3;                     // the LDS kernel below: every instance whose message fits ...
        if (const hipError_t e = launch_ukf_step(q, stream); e != hipSuccess) return e;
        q.long_mode = 2;                     // ... and the streamed kernel: the others
        return launch_ukf_big_step(q, stream);
    }
    if (nmax <= 44) {
        switch (env_tpb(1, 128)) {
            case 64: hipLaunchKernelGGL((ukf_step_kernel<44, 64, 8>), dim3(p.b_cnt), dim3(64), 0, stream, p); break;
            case 256: hipLaunchKernelGGL((ukf_step_kernel<44, 256, 8>), dim3(p.b_cnt), dim3(256), 0, stream, p); break;
            case 192: hipLaunchKernelGGL((ukf_step_kernel<44, 192, 8>), dim3(p.b_cnt), dim3(192), 0, stream, p); break;   // one wavefront per tile row of the covariance
            default:   // measured best
                if (p.prof) hipLaunchKernelGGL((ukf_step_kernel<44, 128, 8, true>), dim3(p.b_cnt), dim3(128), 0, stream, p);
                // KU = 3 updates kept for one pass over P (round 4; it was 8: 33 KB of LDS and 188 VGPRs = four workgroups per CU; 3: 26 KB, 155 =
                // six; a message with more than three detections of mapped landmarks takes further passes, P back from HBM): 4.69 -> 4.92 M
                // steps/s at L = 20 (KU = 4: 4.74, KU = 2: 4.74 - the mean is 1.3 - 2 detections per message)
#ifndef SLAM_UKF_STEP_KU
#define SLAM_UKF_STEP_KU 3
#endif
                else hipLaunchKernelGGL((ukf_step_kernel<44, 128, SLAM_UKF_STEP_KU>), dim3(p.b_cnt), dim3(128), 0, stream, p);
                break;
        }
    } else if (nmax <= 104) {
        // 512 threads (220 VGPRs) since round 4: with 1024 the compiler has 128 VGPRs and spills 82 of them; 345 -> 350 k steps/s at L = 50
        // once the sqrt kernel's passes left the step kernel a quarter of the step (1024 had measured best against the round-3 sqrt kernel)
        switch (p.prof ? 1024 : env_tpb(1, 512)) {
            case 256: hipLaunchKernelGGL((ukf_step_kernel<104, 256, 8>), dim3(p.b_cnt), dim3(256), 0, stream, p); break;
            case 1024:
                if (p.prof) hipLaunchKernelGGL((ukf_step_kernel<104, 1024, 8, true>), dim3(p.b_cnt), dim3(1024), 0, stream, p);
                else hipLaunchKernelGGL((ukf_step_kernel<104, 1024, 8>), dim3(p.b_cnt), dim3(1024), 0, stream, p);
                break;
            default: hipLaunchKernelGGL((ukf_step_kernel<104, 512, 8>), dim3(p.b_cnt), dim3(512), 0, stream, p); break;
        }
    } else {
        return launch_ukf_big_step(p, stream);   // every n x n object in HBM / L2 (ukf_big_kernel.hip)
    }
    return hipGetLastError();
}

__global__ void ukf_init_kernel(const UkfInitParams p) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    double* P = p.P + (size_t)b * p.pstride;
    double* x = p.x + (size_t)b * p.xstride;
    for (int i = 0; i < 16; ++i) P[i] = 0.0;
    P[0] = 0.01 * 0.01; P[5] = 0.01 * 0.01; P[10] = 0.005 * 0.005; P[15] = 0.005 * 0.005;   // ukf.cpp:9-13
    x[0] = p.x0; x[1] = p.y0; x[2] = p.c0; x[3] = p.s0;                                       // ukf.cpp:33
    p.M[b] = 0; p.flags[b] = 0; p.timestep[b] = 0; p.n_sq[b] = 0; p.v_age[b] = -1;
    p.truth[3 * (size_t)b] = p.tx; p.truth[3 * (size_t)b + 1] = p.ty; p.truth[3 * (size_t)b + 2] = p.tyaw;
    p.err_sum[b] = 0.0;
}
hipError_t launch_ukf_init(const UkfInitParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(ukf_init_kernel, dim3((p.B + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace slam
