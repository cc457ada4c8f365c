// ukf_big_kernel.hip — UKF::update (reference ekf_ws/src/localization_pkg/src/ukf.cpp:161-372) for states that do NOT fit the LDS size
// classes of ukf_kernel.hip (n = 4 + 2 L > 104, up to kUkfMaxLandmarks landmarks).  The reference grows its state by two per inserted
// landmark without a limit (ukf.cpp:357,371); the fast kernels keep the packed matrix, the eigenvectors and the square root of one
// instance in LDS and therefore end at 50 landmarks.  This is the size class beyond (round 4): the same two launches per timestep, one
// workgroup per instance, with every n x n object (the scaled symmetrised matrix, the eigenvectors, the square root, P_pred) in HBM / L2
// and only vectors in LDS.  Every element is formed by the expression the fast kernels and the oracle use, in the same order - the
// parallel-order Jacobi with its warm start, fused products, tau-free parameters; the weighted covariance as a chain of fused
// multiply-adds in ascending sigma index; updates before insertions, detection by detection - so the results are bit-identical to the
// oracle (and a message may hold any number of detections: it is walked where it lies).  Slow by design: thousands of barrier-separated
// rounds over global memory per eigen-decomposition.  The STEP kernel also serves the LDS classes (UKF_SLAM up to 50 landmarks, UKF_LOC) for
// the instances whose message is longer than their step kernel holds (UkfStepParams::long_mode; the sqrt kernel stays the class's own).
#include "ukf_kernel.h"

#include "../../include/slam_batch.h"
#include "jacobi_schedule.h"
#include "lds_attr.h"
#include "sim_device.h"
#include "slam_math.h"
#include "slam_rng.h"

namespace slam {

namespace {

constexpr int kTpb = 1024;
constexpr float kW0b = 0.2f;   // filter.h:207
constexpr int kWarmMaxAge = 100;

__device__ __forceinline__ void tsc(float a, int float_trig, double* s, double* c) {   // unqualified cos / sin on a float argument
    double ss, cc;
    det_sincos((double)a, &ss, &cc);
    *s = float_trig ? (double)(float)ss : ss;
    *c = float_trig ? (double)(float)cc : cc;
}
__device__ __forceinline__ float yawf(double c, double s) { return (float)remainder(det_atan2(s, c), kTwoPi); }
__device__ __forceinline__ bool inv2(const double S[4], double Si[4]) {   // MatrixXd::inverse() of a 2 x 2 (ukf.cpp:339): PartialPivLU
    const bool sw = fabs(S[2]) > fabs(S[0]);
    const double a00 = sw ? S[2] : S[0], a01 = sw ? S[3] : S[1];
    const double a10 = sw ? S[0] : S[2], a11 = sw ? S[1] : S[3];
    const double l = a10 / a00;
    const double u11 = a11 - l * a01;
    const bool ok = (a00 != 0.0) && (u11 != 0.0);
    { const double r0 = sw ? 0.0 : 1.0, r1 = sw ? 1.0 : 0.0; const double y1 = r1 - l * r0; const double x1 = y1 / u11; Si[0] = (r0 - a01 * x1) / a00; Si[2] = x1; }
    { const double r0 = sw ? 1.0 : 0.0, r1 = sw ? 0.0 : 1.0; const double y1 = r1 - l * r0; const double x1 = y1 / u11; Si[1] = (r0 - a01 * x1) / a00; Si[3] = x1; }
    return ok;
}

// ------------------------------------------------------------------------------------------------------------------
// nearestSPD + principal square root (ukf.cpp:106-123,208): parallel-order Jacobi on matrices in global memory
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTpb) void ukf_big_sqrt_kernel(const UkfStepParams p) {
    extern __shared__ double sm[];
    const int b = blockIdx.x + p.b_off, tid = threadIdx.x;
    const int M = p.M[b], n = 4 + 2 * M, m = n / 2;
    const int nmax = 4 + 2 * p.L_max, mmax = nmax / 2;
    double* const s_cs = sm;                    // [mmax]
    double* const s_sn = s_cs + mmax;
    double* const s_tn = s_sn + mmax;
    double* const s_sd = s_tn + mmax;           // [nmax]
    int* const s_pp = reinterpret_cast<int*>(s_sd + nmax);   // [mmax]
    int* const s_qq = s_pp + mmax;
    const double* __restrict__ Pb = p.P + (size_t)b * p.pstride;
    double* Sq = p.sqtP + (size_t)b * p.pstride;
    double* Vt = p.Vt_store + (size_t)b * p.pstride;         // V^T: row q = eigenvector q (n x n once this kernel has laid it out)
    double* A = p.big_ws + (size_t)b * 2 * p.pstride;        // the scaled symmetrised matrix, full n x n, kept exactly symmetric
    double* T = A + p.pstride;                               // scratch of the warm start
    const float scale_f = (float)(2 * M + 4) / (1 - kW0b);   // ukf.cpp:114, evaluated in float
    const double scale = (double)scale_f;
    const int age = p.v_age[b], n_v = p.n_sq[b];
    const bool warm = age >= 0 && age < kWarmMaxAge && n_v > 0 && n_v <= n;

    for (int e = tid; e < n * n; e += kTpb) {
        const int r = e / n, c = e - r * n;
        A[e] = (0.5 * (Pb[(size_t)r * n + c] + Pb[(size_t)c * n + r])) * scale;
    }
    if (warm) {
        // V0 extended by the identity for the landmarks inserted since (through T: the leading dimension changes)
        if (n_v != n) {
            for (int e = tid; e < n * n; e += kTpb) {
                const int r = e / n, c = e - r * n;
                T[e] = (r < n_v && c < n_v) ? Vt[(size_t)r * n_v + c] : (r == c ? 1.0 : 0.0);
            }
            __syncthreads();
            for (int e = tid; e < n * n; e += kTpb) Vt[e] = T[e];
        }
        __syncthreads();
        // T = A V0, then B = V0^T T (lower triangle, mirrored): each term fused, ascending k (V0(k, c) = Vt[c][k])
        for (int e = tid; e < n * n; e += kTpb) {
            const int r = e / n, c = e - r * n;
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc = fma(A[(size_t)r * n + k], Vt[(size_t)c * n + k], acc);
            T[e] = acc;
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += kTpb) {
            const int r = e / n, c = e - r * n;
            if (c > r) continue;
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc = fma(Vt[(size_t)r * n + k], T[(size_t)k * n + c], acc);
            A[(size_t)r * n + c] = acc; A[(size_t)c * n + r] = acc;
        }
    } else {
        for (int e = tid; e < n * n; e += kTpb) { const int r = e / n, c = e - r * n; Vt[e] = (r == c) ? 1.0 : 0.0; }
    }
    __syncthreads();
    const int tiny_from = warm ? 0 : 3;
    const int nb = m * (m - 1) / 2;
    bool converged = false;
    int sweeps_done = 0;
#pragma unroll 1
    for (int sweep = 0; sweep < 60; ++sweep) {
        int live = 0;   // convergence: every off-diagonal element is exactly zero
        for (int e = tid; e < n * n; e += kTpb) {
            const int r = e / n, c = e - r * n;
            if (c < r && A[e] != 0.0 && A[e] == A[e]) live = 1;   // (NaNs are passed over, as by the oracle's std::max)
        }
        if (!__syncthreads_or(live)) { converged = true; sweeps_done = sweep; break; }
#pragma unroll 1
        for (int t = 0; t < n - 1; ++t) {
            for (int k = tid; k < m; k += kTpb) {   // rotation parameters of this round's pairs (jacobi_schedule.h)
                int pi, qi;
                jacobi_pair(k, t, n, pi, qi);
                const double app = A[(size_t)pi * n + pi], aqq = A[(size_t)qi * n + qi], apq = A[(size_t)qi * n + pi];
                double c = 1.0, s = 0.0, tt = 0.0;
                const double g = 100.0 * fabs(apq);
                const bool tiny = sweep >= tiny_from && (fabs(app) + g == fabs(app)) && (fabs(aqq) + g == fabs(aqq));
                if (apq != 0.0 && !tiny) {
                    const double d = aqq - app, b2 = 2.0 * apq;
                    const double h = sqrt(fma(d, d, b2 * b2));
                    if (h > 0.0) {
                        const double w = fabs(d) + h;
                        const bool pos = (d == 0.0) || ((d > 0.0) == (b2 > 0.0));
                        tt = (pos ? fabs(b2) : -fabs(b2)) / w;
                        c = sqrt(w / (2.0 * h));
                        s = tt * c;
                    }
                }
                s_pp[k] = pi; s_qq[k] = qi; s_cs[k] = c; s_sn[k] = s; s_tn[k] = tt;
            }
            __syncthreads();
            // pair-blocks (i, j), i > j:  B' = R_i^T B R_j, written to both triangles
            for (int it = tid; it < nb; it += kTpb) {
                int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)it)) * 0.5f);
                while (i * (i - 1) / 2 > it) --i;
                while ((i + 1) * i / 2 <= it) ++i;
                const int j = it - i * (i - 1) / 2;
                const double si = s_sn[i], sj = s_sn[j];
                if (si == 0.0 && sj == 0.0) continue;   // both rotations the identity (c = 1 exactly): B' = B bit for bit
                const int pi = s_pp[i], qi = s_qq[i], pj = s_pp[j], qj = s_qq[j];
                const double ci = s_cs[i], cj = s_cs[j];
                const double b00 = A[(size_t)pi * n + pj], b01 = A[(size_t)pi * n + qj], b10 = A[(size_t)qi * n + pj], b11 = A[(size_t)qi * n + qj];
                const double t00 = fma(ci, b00, -(si * b10)), t01 = fma(ci, b01, -(si * b11));
                const double t10 = fma(si, b00, ci * b10), t11 = fma(si, b01, ci * b11);
                const double r00 = fma(t00, cj, -(t01 * sj)), r01 = fma(t00, sj, t01 * cj);
                const double r10 = fma(t10, cj, -(t11 * sj)), r11 = fma(t10, sj, t11 * cj);
                A[(size_t)pi * n + pj] = r00; A[(size_t)pj * n + pi] = r00;
                A[(size_t)pi * n + qj] = r01; A[(size_t)qj * n + pi] = r01;
                A[(size_t)qi * n + pj] = r10; A[(size_t)pj * n + qi] = r10;
                A[(size_t)qi * n + qj] = r11; A[(size_t)qj * n + qi] = r11;
            }
            for (int i = tid; i < m; i += kTpb) {   // diagonal blocks
                const int pq = s_pp[i], qq = s_qq[i];
                const double app = A[(size_t)pq * n + pq], aqq = A[(size_t)qq * n + qq], apq = A[(size_t)qq * n + pq];
                A[(size_t)pq * n + pq] = fma(-s_tn[i], apq, app);
                A[(size_t)qq * n + qq] = fma(s_tn[i], apq, aqq);
                if (apq != 0.0) { A[(size_t)qq * n + pq] = 0.0; A[(size_t)pq * n + qq] = 0.0; }
            }
            for (int it = tid; it < m * n; it += kTpb) {   // V <- V J: rows p, q of V^T
                const int i = it / n, k = it - i * n;
                const double s = s_sn[i];
                if (s == 0.0) continue;
                const int pq = s_pp[i], qq = s_qq[i];
                const double c = s_cs[i];
                const double vp = Vt[(size_t)pq * n + k], vq = Vt[(size_t)qq * n + k];
                Vt[(size_t)pq * n + k] = fma(c, vp, -(s * vq));
                Vt[(size_t)qq * n + k] = fma(s, vp, c * vq);
            }
            __syncthreads();
        }
    }
    if (tid == 0 && p.khist && converged) { atomicAdd(&p.khist[8], (unsigned long long)sweeps_done); atomicAdd(&p.khist[9], 1ull); }
    if (!converged) {   // ukf.cpp:209-211 swallows the exception and reuses the stale sqtP; a stale matrix of another size cannot be used
        if (p.n_sq[b] != n)
            for (int e = tid; e < n * n; e += kTpb) Sq[e] = 0.0;
        __syncthreads();
        if (tid == 0) { p.flags[b] = p.flags[b] | SLAM_INST_SQRT_FAILED; p.n_sq[b] = n; p.v_age[b] = -1; }
        return;
    }
    for (int k = tid; k < n; k += kTpb) {
        const double d = A[(size_t)k * n + k];
        s_sd[k] = sqrt(d > 0.00000001 ? d : 0.00000001);   // cwiseMax(1e-8), then the principal square root
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += kTpb) {
        const int r = e / n, c = e - r * n;
        if (c > r) continue;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc = fma(Vt[(size_t)k * n + r] * s_sd[k], Vt[(size_t)k * n + c], acc);   // (fused, ascending k: as the oracle and the MFMA form of the LDS classes)
        Sq[(size_t)r * n + c] = acc;
        Sq[(size_t)c * n + r] = acc;
    }
    if (tid == 0) { p.v_age[b] = warm ? age + 1 : 0; p.n_sq[b] = n; }
}

// ------------------------------------------------------------------------------------------------------------------
// predictionStage + updateStage (ukf.cpp:197-372)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTpb) void ukf_big_step_kernel(const UkfStepParams p) {
    extern __shared__ double sm[];
    const int b = blockIdx.x + p.b_off, tid = threadIdx.x, lane = tid & 63;
    const int nmax = 4 + 2 * p.L_max, nsmax = 2 * nmax + 1;
    double* const s_xt = sm;                      // [nmax]
    double* const s_xp = s_xt + nmax;             // [nmax]
    double* const s_X4 = s_xp + nmax;             // [4][nsmax]
    double* const s_Z0 = s_X4 + 4 * nsmax;        // [nsmax]
    double* const s_Z1 = s_Z0 + nsmax;            // [nsmax]
    double* const s_K = s_Z1 + nsmax;             // [nmax][2]   C, then K
    double* const s_KS = s_K + 2 * nmax;          // [nmax][2]
    double* const s_sc = s_KS + 2 * nmax;         // [32]
    int* const s_i = reinterpret_cast<int*>(s_sc + 32);   // [16]: 0 association, 1 detections, 2 flags raised
    int* const s_ids = s_i + 16;                          // [L_max]
    float* const s_meas = reinterpret_cast<float*>(s_ids + ((p.L_max + 1) & ~1));   // SIM mode: [3 L]

    if (p.long_mode == 2) {   // paired with the LDS kernel's launch: only the instances whose message that kernel cannot hold (ukf_kernel.h)
        const int kk = p.meas_count_in[b];
        if ((kk < p.k_stride_in ? kk : p.k_stride_in) <= p.long_cap) return;
    }
    int flags = p.flags[b];
    const int M_old = p.M[b];
    const int n = 4 + 2 * M_old, ns = 2 * n + 1;
    double* __restrict__ Pout = p.P_out + (size_t)b * p.pstride;
    double* __restrict__ xb = p.x + (size_t)b * p.xstride;
    const double* __restrict__ Sq = p.sqtP + (size_t)b * p.pstride;
    double* __restrict__ Pw = p.big_ws + (size_t)b * 2 * p.pstride;   // P_pred, n x n (the sqrt kernel's scratch is free now)

    for (int i = tid; i < nmax; i += kTpb) {
        const double v = i < n ? xb[i] : 0.0;
        s_xt[i] = v;
        if (i < n && p.x_prev) p.x_prev[(size_t)b * p.xstride + i] = v;   // centre of this step's sigma points (ukf.cpp:214)
    }
    for (int i = tid; i < M_old; i += kTpb) s_ids[i] = p.ids[(size_t)b * p.L_max + i];
    if (tid < 16) s_i[tid] = 0;
    __syncthreads();
    double tx = 0.0, ty = 0.0;
    const float* meas = s_meas;
    if (p.sim) {
        if (tid < 64) {
            double tth = p.truth[3 * (size_t)b + 2];
            tx = p.truth[3 * (size_t)b]; ty = p.truth[3 * (size_t)b + 1];
            const double lmx = lane < p.L ? p.map[2 * lane] : 0.0, lmy = lane < p.L ? p.map[2 * lane + 1] : 0.0;
            const int cnt = sim_wave<(1 << 30)>(p, b, lane, p.fwd, p.ang, p.step, tx, ty, tth, lmx, lmy, s_meas);   // stores the new true pose
            if (lane == 0) { s_i[1] = cnt; s_sc[24] = tx; s_sc[25] = ty; }
        }
    } else {
        int kk = p.meas_count_in[b];
        kk = kk < p.k_stride_in ? kk : p.k_stride_in;
        kk = kk < 0 ? 0 : kk;
        if (tid == 0) s_i[1] = kk;
        meas = p.meas_in + (size_t)b * p.k_stride_in * 3;   // walked where it lies
    }
    __syncthreads();
    const int k = s_i[1];
    if (tid == 0 && p.khist) atomicAdd(&p.khist[k < 7 ? k : 7], 1ull);
    if (p.sim && p.meas_out != nullptr) {
        for (int i = tid; i < 3 * k && i < 3 * p.k_stride_out; i += kTpb) p.meas_out[(size_t)b * p.k_stride_out * 3 + i] = s_meas[i];
        if (tid == 0) p.meas_count_out[b] = k < p.k_stride_out ? k : p.k_stride_out;
    }

    // ---- sigma points through the motion model (ukf.cpp:214-226,125-135); only rows 0..3 change ----
    const float u_d = p.fwd, u_th = p.ang;
    const float dd = u_d + p.v_d;
    for (int i = tid; i < ns; i += kTpb) {
        double v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (i == 0) v[r] = s_xt[r];
            else if (i <= n) v[r] = s_xt[r] + Sq[(size_t)r * n + (i - 1)];
            else v[r] = s_xt[r] - Sq[(size_t)r * n + (i - 1 - n)];
        }
        const float yaw = yawf(v[2], v[3]);
        double sy, cy;
        tsc(yaw, p.float_trig, &sy, &cy);
        if (p.float_trig) {
            s_X4[0 * ns + i] = v[0] + (double)(dd * (float)cy);   // float * float (ukf.cpp:129)
            s_X4[1 * ns + i] = v[1] + (double)(dd * (float)sy);
        } else {
            s_X4[0 * ns + i] = v[0] + (double)dd * cy;
            s_X4[1 * ns + i] = v[1] + (double)dd * sy;
        }
        const float new_yaw = (float)remainder((double)(yaw + u_th + p.v_th), kTwoPi);   // float adds (ukf.cpp:131)
        double sn, cn;
        tsc(new_yaw, p.float_trig, &sn, &cn);
        s_X4[2 * ns + i] = cn;
        s_X4[3 * ns + i] = sn;
    }
    __syncthreads();
    const double w0 = (double)kW0b;
    const double wi = (double)((1 - kW0b) / (2 * n));   // float arithmetic (ukf.cpp:174-175)
    auto xel = [&](int r, int i) -> double {   // X_pred(r, i): rows 0..3 from the motion model, rows >= 4 = the sigma point itself
        if (r < 4) return s_X4[r * ns + i];
        if (i == 0) return s_xt[r];
        if (i <= n) return s_xt[r] + Sq[(size_t)r * n + (i - 1)];
        return s_xt[r] - Sq[(size_t)r * n + (i - 1 - n)];
    };
    // ---- weighted mean (ukf.cpp:228-232), sequential in i ----
    for (int r = tid; r < n; r += kTpb) {
        double acc = 0.0;
        for (int i = 0; i < ns; ++i) acc = acc + (i == 0 ? w0 : wi) * xel(r, i);
        s_xp[r] = acc;
    }
    __syncthreads();
    // ---- weighted covariance (ukf.cpp:235-240): acc = fma(w_i d_r(i), d_c(i), acc) in ascending i, + signed Q ----
    for (int e = tid; e < n * n; e += kTpb) {
        const int r = e / n, c = e - r * n;
        const double xr = s_xp[r], xc = s_xp[c];
        double acc = 0.0;
        for (int i = 0; i < ns; ++i) acc = fma((i == 0 ? w0 : wi) * (xel(r, i) - xr), xel(c, i) - xc, acc);
        Pw[e] = acc;
    }
    __syncthreads();
    const float yaw_t = yawf(s_xt[2], s_xt[3]);   // yaw of x_t: process noise diagonal (ukf.cpp:182-186) and the sensing model (ukf.cpp:139)
    if (tid == 0) {
        double sy, cy;
        tsc(yaw_t, p.float_trig, &sy, &cy);
        Pw[0] = Pw[0] + p.V00 * cy;
        Pw[(size_t)1 * n + 1] = Pw[(size_t)1 * n + 1] + p.V00 * sy;
        Pw[(size_t)2 * n + 2] = Pw[(size_t)2 * n + 2] + p.V11 * cy;
        Pw[(size_t)3 * n + 3] = Pw[(size_t)3 * n + 3] + p.V11 * sy;
    }
    __syncthreads();

    // association of detection l against the landmarks known BEFORE this step (ukf.cpp:256-277): lowest match, or -1
    auto associate = [&](int id) -> int {
        if (tid == 0) s_i[0] = 0x7fffffff;
        __syncthreads();
        int hit = 0x7fffffff;
        for (int j = tid; j < M_old; j += kTpb)
            if (s_ids[j] == id) { hit = j; break; }
        if (hit != 0x7fffffff) atomicMin(&s_i[0], hit);
        __syncthreads();
        const int f = s_i[0];
        __syncthreads();
        return f == 0x7fffffff ? -1 : f;
    };
    // ---- pass 1: landmark updates in message order (ukf.cpp:293-349) ----
#pragma unroll 1
    for (int l = 0; l < k; ++l) {
        const int id_l = (int)meas[3 * l];
        // UKF_LOC (ukf.cpp:146-154,272-276): every detection is an update against the known map, there is no landmark in the state
        const int j = p.loc ? ((id_l >= 0 && id_l < p.L) ? id_l : -2) : associate(id_l);
        if (j == -2 && tid == 0) s_i[2] |= SLAM_INST_INDEX_OOR;   // an id outside the map
        if (j < 0) continue;
        const float r_m = meas[3 * l + 1], b_m = meas[3 * l + 2];
        const int li = p.loc ? 0 : 2 * j + 4;
        const double mx = p.loc ? (double)p.mapf[3 * j + 1] : 0.0, my = p.loc ? (double)p.mapf[3 * j + 2] : 0.0;
        for (int i = tid; i < ns; i += kTpb) {   // sensing model of every sigma point (yaw from x_t)
            const double dx = (p.loc ? mx : xel(li, i)) - xel(0, i), dy = (p.loc ? my : xel(li + 1, i)) - xel(1, i);
            s_Z0[i] = sqrt(dx * dx + dy * dy) + (double)p.w_r;
            const double yaw_i = p.yaw_sigma ? (double)yawf(xel(2, i), xel(3, i)) : (double)yaw_t;   // quirk D-9: from x_t
            s_Z1[i] = remainder((det_atan2(dy, dx) - yaw_i) + (double)p.w_b, kTwoPi);
        }
        __syncthreads();
        if (tid == 0) {   // z_est (the bearing component is never accumulated, ukf.cpp:310-314 - quirk D-8 - unless switched) and S
            double z0 = 0.0, zb = 0.0;
            for (int i = 0; i < ns; ++i) z0 = z0 + (i == 0 ? w0 : wi) * s_Z0[i];
            if (p.acc_zest1)
                for (int i = 0; i < ns; ++i) zb = zb + (i == 0 ? w0 : wi) * s_Z1[i];
            s_sc[1] = zb;
            double S[4] = {0.0, 0.0, 0.0, 0.0};
            for (int i = 0; i < ns; ++i) {
                const double d0 = s_Z0[i] - z0, d1 = remainder(s_Z1[i] - zb, kTwoPi);
                const double ww = (i == 0 ? w0 : wi);
                const double a0 = ww * d0, a1 = ww * d1;
                S[0] = S[0] + a0 * d0; S[1] = S[1] + a0 * d1; S[2] = S[2] + a1 * d0; S[3] = S[3] + a1 * d1;
            }
            S[0] = S[0] + p.W00; S[1] = S[1] + 0.0; S[2] = S[2] + 0.0; S[3] = S[3] + p.W11;
            double Si[4];
            if (!inv2(S, Si)) s_i[2] |= SLAM_INST_S_SINGULAR;
            s_sc[0] = z0;
            for (int q = 0; q < 4; ++q) { s_sc[4 + q] = S[q]; s_sc[8 + q] = Si[q]; }
            s_sc[12] = (double)r_m - z0;
            s_sc[13] = remainder((double)b_m - zb, kTwoPi);
        }
        __syncthreads();
        {
            const double z0 = s_sc[0], zb = s_sc[1];
            for (int r = tid; r < n; r += kTpb) {   // cross covariance C (uses the CURRENT x_pred, ukf.cpp:330), K = C S^-1
                const double xr = s_xp[r];
                double c0 = 0.0, c1 = 0.0;
                for (int i = 0; i < ns; ++i) {
                    const double wd = (i == 0 ? w0 : wi) * (xel(r, i) - xr);
                    const double d0 = s_Z0[i] - z0, d1 = remainder(s_Z1[i] - zb, kTwoPi);
                    c0 = c0 + wd * d0; c1 = c1 + wd * d1;
                }
                const double k0 = c0 * s_sc[8] + c1 * s_sc[10], k1 = c0 * s_sc[9] + c1 * s_sc[11];
                s_K[2 * r] = k0; s_K[2 * r + 1] = k1;
                s_KS[2 * r] = k0 * s_sc[4] + k1 * s_sc[6];
                s_KS[2 * r + 1] = k0 * s_sc[5] + k1 * s_sc[7];
            }
        }
        __syncthreads();
        for (int r = tid; r < n; r += kTpb) s_xp[r] = s_xp[r] + (s_K[2 * r] * s_sc[12] + s_K[2 * r + 1] * s_sc[13]);
        for (int e = tid; e < n * n; e += kTpb) {   // P_pred -= K S K^T
            const int r = e / n, c = e - r * n;
            Pw[e] = Pw[e] - (s_KS[2 * r] * s_K[2 * c] + s_KS[2 * r + 1] * s_K[2 * c + 1]);
        }
        __syncthreads();
    }
    // ---- pass 2: insertions in message order (ukf.cpp:351-372): x_pred grows, P = blkdiag(P_pred, W) ----
    int M = M_old;
#pragma unroll 1
    for (int l = 0; l < (p.loc ? 0 : k); ++l) {
        const int id = (int)meas[3 * l];
        const int j = associate(id);
        if (j >= 0) continue;
        if (M >= p.L_max) { if (tid == 0) s_i[2] |= SLAM_INST_CAPACITY; continue; }
        if (tid == 0) {
            const float r_m = meas[3 * l + 1], b_m = meas[3 * l + 2];
            const int nn = 4 + 2 * M;
            const float yaw = yawf(s_xp[2], s_xp[3]);
            const float ang = yaw + b_m;
            double sa, ca;
            tsc(ang, p.float_trig, &sa, &ca);
            if (p.float_trig) {
                s_xp[nn] = s_xp[0] + (double)(r_m * (float)ca);
                s_xp[nn + 1] = s_xp[1] + (double)(r_m * (float)sa);
            } else {
                s_xp[nn] = s_xp[0] + (double)r_m * ca;
                s_xp[nn + 1] = s_xp[1] + (double)r_m * sa;
            }
            s_ids[M] = id;
        }
        M += 1;
        __syncthreads();
    }
    __syncthreads();
    // ---- x_t = x_pred, P_t = P_pred (ukf.cpp:289-290) in the layout of the new state size ----
    const int n_fin = 4 + 2 * M;
    int bad = 0;
    for (int e = tid; e < n_fin * n_fin; e += kTpb) {
        const int r = e / n_fin, c = e - r * n_fin;
        const double v = (r < n && c < n) ? Pw[(size_t)r * n + c] : ((r == c) ? (((r - n) & 1) ? p.W11 : p.W00) : 0.0);
        bad |= !isfinite(v);
        Pout[e] = v;
    }
    for (int i = tid; i < n_fin; i += kTpb) {
        const double v = s_xp[i];
        bad |= !isfinite(v);
        xb[i] = v;
    }
    if (__syncthreads_or(bad)) flags |= SLAM_INST_NONFINITE;
    flags |= s_i[2];
    if (M != M_old)
        for (int i = tid; i < M; i += kTpb) p.ids[(size_t)b * p.L_max + i] = s_ids[i];
    if (tid == 0) {
        p.M[b] = M;
        p.flags[b] = flags | (p.flags[b] & SLAM_INST_SQRT_FAILED);
        p.timestep[b] = p.timestep[b] + 1;
        if (p.sim) {
            const double ex = (double)(float)s_xp[0] - s_sc[24], ey = (double)(float)s_xp[1] - s_sc[25];
            p.err_sum[b] = p.err_sum[b] + sqrt(ex * ex + ey * ey);
        }
    }
}

size_t big_sqrt_lds(int L_max) {
    const int nmax = 4 + 2 * L_max, mmax = nmax / 2;
    return sizeof(double) * (size_t)(3 * mmax + nmax) + sizeof(int) * (size_t)(2 * mmax) + 16;
}
size_t big_step_lds(int L_max, int L_map) {
    const int nmax = 4 + 2 * L_max, nsmax = 2 * nmax + 1;
    return sizeof(double) * (size_t)(2 * nmax + 6 * nsmax + 4 * nmax + 32) + sizeof(int) * (size_t)(16 + ((L_max + 1) & ~1)) + sizeof(float) * 3 * (size_t)(L_map > 1 ? L_map : 1) + 16;
}

}  // namespace

hipError_t launch_ukf_big_sqrt(const UkfStepParams& p, hipStream_t stream) {
    if (p.big_ws == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ukf_big_sqrt_kernel, dim3(p.b_cnt), dim3(kTpb), big_sqrt_lds(p.L_max), stream, p);
    return hipGetLastError();
}
hipError_t launch_ukf_big_step(const UkfStepParams& p, hipStream_t stream) {
    if (p.big_ws == nullptr) return hipErrorInvalidValue;
    const size_t lds = big_step_lds(p.L_max, p.sim ? p.L : 1);
    if (lds > 159 * 1024) return hipErrorInvalidConfiguration;
    if (lds > 64 * 1024) {   // once per device, to the kernel's maximum (lds_attr.h)
        const hipError_t e = slam_allow_full_lds(reinterpret_cast<const void*>(&ukf_big_step_kernel));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(ukf_big_step_kernel, dim3(p.b_cnt), dim3(kTpb), lds, stream, p);
    return hipGetLastError();
}

}  // namespace slam
