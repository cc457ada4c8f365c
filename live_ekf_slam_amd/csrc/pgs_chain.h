// pgs_chain.h — the sequential elimination of the pose chain (rounds 1-4; the path of graphs the segmented elimination cannot hold).
// Part of pgs_kernel.hip (round 6: split by phase, pure moves); included there inside namespace slam { namespace {.  DESIGN.md 4.4.
#pragma once

// Block-tridiagonal Cholesky of H_pp + lambda I fused with the forward recurrence over the landmark columns.
// Wavefront 0 is the PRODUCER: per chunk of 64 poses its lanes stage A_i, C_{i-1}, gp_i in LDS, lane 0 runs the
// sequential 3x3 chain (G_i = C_{i-1} L_{i-1}^-T, L_i = chol(A_i + lambda I - G_i G_i^T), L_i^-1) and leaves
// (L_i^-1, G_i, gp_i) in an LDS ring; it works one chunk ahead of the CONSUMER wavefronts, whose threads own one column
// of Y each (c < 2M: landmark column, c == 2M: gradient column z) and apply  Y_i = L_i^-1 (E_i - G_i Y_{i-1}).
// A column's non-zero E entries come from its landmark's chronological factor list (evt_*, Elm), prefetched one
// event ahead, so the recurrence never searches the measurement slots.  One barrier per chunk.
constexpr int CHAIN_CH = 64;
// 1 / sqrt(x) for x > 0 to ~1 ulp: hardware estimate refined by two Newton steps y <- y + y * (1 - x y^2) / 2
__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = __builtin_fma(-(x * y), y, 1.0);
        y = __builtin_fma(y * 0.5, e, y);
    }
    return y;
}
__global__ __launch_bounds__(1024) void pgs_chain_kernel(const PgsParams p) {
    __shared__ double s_in[CHAIN_CH][18];          // A (6 unique), C (9), gp (3)
    __shared__ double s_ring[2][CHAIN_CH][18];     // Linv (6), G (9), gp (3)
    __shared__ int s_fail;
    const int b = pgs_slot(p, blockIdx.x), tid = threadIdx.x;
    if (p.state[b]) return;
    const int N = pgs_N(p, b), LD = p.LD, m2 = 2 * p.M[b];
    const double lambda = p.lambda[b];
    const double* Ab = p.A + (size_t)b * p.N_max * 9;
    const double* Cb = p.C + (size_t)b * p.N_max * 9;
    const double* gpb = p.gp + (size_t)b * p.N_max * 3;
    double* Lb = p.Linv + (size_t)b * p.N_max * 6;
    double* Gb = p.G + (size_t)b * p.N_max * 9;
    double* Yb = p.Y + (size_t)b * p.y_stride;
    const bool producer = tid < 64;
    const int c = tid - 64;                        // consumer column
    const int nch = (N + CHAIN_CH - 1) / CHAIN_CH;
    if (tid == 0) s_fail = 0;
    // consumer state
    double y0 = 0.0, y1 = 0.0, y2 = 0.0, e0 = 0.0, e1 = 0.0, e2 = 0.0;
    int cur = 0, end = 0, next_i = 0x7fffffff;
    const double* Elmb = p.Elm + (size_t)b * p.N_max * p.KP * 6;
    const int32_t* evt_pose = p.evt_pose + (size_t)b * p.N_max * p.KP;
    const int myd = c & 1;
    if (!producer && c < m2) {
        const int32_t* evt_start = p.evt_start + (size_t)b * (p.L_max + 1);
        cur = evt_start[c >> 1]; end = evt_start[(c >> 1) + 1];
        if (cur < end) {
            next_i = evt_pose[cur];
            e0 = Elmb[6 * (size_t)cur + myd]; e1 = Elmb[6 * (size_t)cur + 2 + myd]; e2 = Elmb[6 * (size_t)cur + 4 + myd];
        }
    }
    // producer state (lane 0): Linv of the previous pose
    double I0 = 0, I1 = 0, I2 = 0, I3 = 0, I4 = 0, I5 = 0;
    __syncthreads();
#pragma unroll 1
    for (int it = 0; it <= nch; ++it) {
        if (producer) {
            if (it < nch) {
                const int base = it * CHAIN_CH;
                const int n = (N - base) < CHAIN_CH ? (N - base) : CHAIN_CH;
                const int i = base + tid;
                if (tid < n) {
                    const double* A = Ab + 9 * i;
                    s_in[tid][0] = A[0]; s_in[tid][1] = A[3]; s_in[tid][2] = A[4]; s_in[tid][3] = A[6]; s_in[tid][4] = A[7]; s_in[tid][5] = A[8];
                    if (i > 0) {
                        const double* C = Cb + 9 * (i - 1);
#pragma unroll
                        for (int k = 0; k < 9; ++k) s_in[tid][6 + k] = C[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 9; ++k) s_in[tid][6 + k] = 0.0;
                    }
                    s_in[tid][15] = gpb[3 * i]; s_in[tid][16] = gpb[3 * i + 1]; s_in[tid][17] = gpb[3 * i + 2];
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (tid == 0) {
                    double (*out)[18] = s_ring[it & 1];
                    bool ok = s_fail == 0;
                    double in[18], nx[18];
#pragma unroll
                    for (int k = 0; k < 18; ++k) in[k] = s_in[0][k];
#pragma unroll 1
                    for (int l = 0; l < n && ok; ++l) {
                        const int ln = l + 1 < n ? l + 1 : l;       // next pose's inputs are fetched under this pose's chain
#pragma unroll
                        for (int k = 0; k < 18; ++k) nx[k] = s_in[ln][k];
                        double G[9];
#pragma unroll
                        for (int r = 0; r < 3; ++r) {   // G = C Linv_prev^T (zero for the first pose: C = 0)
                            G[3 * r + 0] = in[6 + 3 * r] * I0;
                            G[3 * r + 1] = in[6 + 3 * r] * I1 + in[6 + 3 * r + 1] * I2;
                            G[3 * r + 2] = (in[6 + 3 * r] * I3 + in[6 + 3 * r + 1] * I4) + in[6 + 3 * r + 2] * I5;
                        }
                        const double T0 = (in[0] + lambda) - ((G[0] * G[0] + G[1] * G[1]) + G[2] * G[2]);
                        const double T3 = in[1] - ((G[3] * G[0] + G[4] * G[1]) + G[5] * G[2]);
                        const double T4 = (in[2] + lambda) - ((G[3] * G[3] + G[4] * G[4]) + G[5] * G[5]);
                        const double T6 = in[3] - ((G[6] * G[0] + G[7] * G[1]) + G[8] * G[2]);
                        const double T7 = in[4] - ((G[6] * G[3] + G[7] * G[4]) + G[8] * G[5]);
                        const double T8 = (in[5] + lambda) - ((G[6] * G[6] + G[7] * G[7]) + G[8] * G[8]);
                        // 3x3 Cholesky through reciprocal square roots (v_rsq_f64 + two Newton steps, ~1 ulp): the three
                        // pivots are the only long-latency operations on the sequential critical path of the solve
                        if (!(T0 > 0.0)) { ok = false; break; }
                        I0 = rsqrt_nr(T0);
                        const double l10 = T3 * I0, l20 = T6 * I0;
                        const double t11 = T4 - l10 * l10;
                        if (!(t11 > 0.0)) { ok = false; break; }
                        I2 = rsqrt_nr(t11);
                        const double l21 = (T7 - l20 * l10) * I2;
                        const double t22 = (T8 - l20 * l20) - l21 * l21;
                        if (!(t22 > 0.0)) { ok = false; break; }
                        I5 = rsqrt_nr(t22);
                        I1 = -(l10 * I0) * I2;
                        I4 = -(l21 * I2) * I5;
                        I3 = -(l20 * I0 + l21 * I1) * I5;
                        double* o = out[l];
                        o[0] = I0; o[1] = I1; o[2] = I2; o[3] = I3; o[4] = I4; o[5] = I5;
#pragma unroll
                        for (int k = 0; k < 9; ++k) o[6 + k] = G[k];
                        o[15] = in[15]; o[16] = in[16]; o[17] = in[17];
#pragma unroll
                        for (int k = 0; k < 18; ++k) in[k] = nx[k];
                    }
                    if (!ok) s_fail = 1;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (tid < n && s_fail == 0) {   // factor to HBM for the pose back-substitution
                    const double* o = s_ring[it & 1][tid];
                    double* L = Lb + 6 * i;
#pragma unroll
                    for (int k = 0; k < 6; ++k) L[k] = o[k];
                    double* Go = Gb + 9 * i;
#pragma unroll
                    for (int k = 0; k < 9; ++k) Go[k] = o[6 + k];
                }
            }
        } else if (it > 0 && c <= m2) {
            const int base = (it - 1) * CHAIN_CH;
            const int n = (N - base) < CHAIN_CH ? (N - base) : CHAIN_CH;
            const double (*rg)[18] = s_ring[(it - 1) & 1];
            double* Yi = Yb + (size_t)3 * base * LD + c;
#pragma unroll 2
            for (int l = 0; l < n; ++l) {
                const int i = base + l;
                const double* o = rg[l];
                double u0 = 0.0, u1 = 0.0, u2 = 0.0;
                if (c == m2) { u0 = o[15]; u1 = o[16]; u2 = o[17]; }
                u0 -= (o[6] * y0 + o[7] * y1) + o[8] * y2;      // G is zero for pose 0
                u1 -= (o[9] * y0 + o[10] * y1) + o[11] * y2;
                u2 -= (o[12] * y0 + o[13] * y1) + o[14] * y2;
                if (i == next_i) {
                    u0 += e0; u1 += e1; u2 += e2;
                    cur += 1;
                    if (cur < end) {
                        next_i = evt_pose[cur];
                        e0 = Elmb[6 * (size_t)cur + myd]; e1 = Elmb[6 * (size_t)cur + 2 + myd]; e2 = Elmb[6 * (size_t)cur + 4 + myd];
                    } else {
                        next_i = 0x7fffffff;
                    }
                }
                y0 = o[0] * u0;
                y1 = o[1] * u0 + o[2] * u1;
                y2 = (o[3] * u0 + o[4] * u1) + o[5] * u2;
                Yi[0] = y0; Yi[LD] = y1; Yi[2 * LD] = y2;
                Yi += 3 * LD;
            }
        }
        __syncthreads();
        if (s_fail) break;
    }
    if (tid == 0) p.solve_ok[b] = s_fail ? 0 : 1;
}
