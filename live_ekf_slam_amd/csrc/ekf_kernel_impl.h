// ekf_kernel_impl.h — fused EKF-SLAM step for gfx950 (MI355X): predict + per-detection update / insertion of
// EKF::update (reference ekf_ws/src/localization_pkg/src/ekf.cpp:37-179), optionally preceded by the
// range-bearing measurement generator get_cmd (ekf_ws/src/base_pkg/src/sim_node.py:209-250).
// Included by the per-variant instantiation units ekf_inst_*.hip so the variants compile in parallel.
//
// Mapping: one workgroup of W wavefronts per filter instance.
//  * BULK: the whole covariance P (n x n fp64, n = 3+2M <= NMAX) lives in VGPRs for the step.  Thread `tid` owns
//    element pairs q = tid + 64W*j (row-major linear index e = 2q, 2q+1), so the one HBM read and the one HBM
//    write of P per step are perfectly coalesced 16-byte-per-lane streams (1 KiB per wave instruction) and move
//    exactly the algorithmic minimum 2*n^2*8 bytes.  The register file (512 KiB/CU) holds four n=103 filters
//    per CU where LDS (160 KiB) could hold one.
//  * THIN: everything EKF::update does besides the rank-2 downdate touches only the rows/columns
//    T = {0,1,2} U {landmarks detected this step} of P.  Those (<= 11 rows + 11 columns) are gathered once into
//    LDS, the predict / H P / S / K / x / insertion algebra runs on the LDS copies (O(k n) work), each
//    detection leaves its K (n x 2) and H P (2 x n) in LDS, and the bulk is then corrected in ONE branch-free
//    register pass  p -= K_l[r] . HP_l[c]  (l in detection order), after which the thin rows/cols are
//    scattered back.  More than KG detections in one step are processed in groups.
//
// Arithmetic: plain IEEE fp64 mul/add/div (-ffp-contract=off), operation order identical to the CPU oracle's
// MODE_FAST (oracle/slam_oracle.cpp) so results are bit-identical; float truncations of the reference
// (ekf.cpp:43-44,57,75-76,115,129-131) are reproduced with real fp32 operations.
#pragma once
#include "ekf_kernel.h"

#include "../../include/slam_batch.h"
#include "slam_math.h"
#include "slam_rng.h"

namespace slam {

template <int NMAX, int W>
struct EkfGeom {
    static constexpr int TPB = 64 * W;
    static constexpr int NEL = NMAX * NMAX;
    static constexpr int NPAIR = (NEL + 1) / 2;
    static constexpr int NP = (NPAIR + TPB - 1) / TPB;   // register pairs per thread
    static constexpr int LDP = (NMAX + 2) & ~1;          // LDS row length (> NMAX, even)
    static constexpr int LMAX = (NMAX - 3) / 2;
    static constexpr int KCAP = LMAX > 0 ? LMAX : 1;     // detections held per step
    static constexpr int KG = 4;                         // detections per group
    static constexpr int TS = 3 + 2 * KG;                // thin rows / cols held in LDS
};

// Occupancy request (waves per SIMD).  P itself needs 4*NP VGPRs per lane; the thin pipeline peaks at ~170
// more (fp64 Jacobian entries, 2x2 inverse, sincos/atan2 polynomials), so today: n<=43 -> 2, n<=103 with four
// waves per filter -> 2 (two filters per CU), with two waves per filter -> 1.  Lowering the thin pipeline's
// register peak is the next occupancy lever.
constexpr int ekf_waves_per_simd(int nmax, int w) {
    const int np = ((nmax * nmax + 1) / 2 + 64 * w - 1) / (64 * w);
    return (4 * np + 170 <= 256) ? 2 : 1;
}

// Visit every owned register pair: f(pair&, r, c, r1, c1, ok0, ok1) with (r,c) the coordinates of .x and
// (r1,c1) of .y in the nf-leading-dimension layout.  (r, c) are re-derived incrementally inside every pass from
// laundered copies of tid / nf: without the empty asm the compiler CSEs the index sequence across passes and
// keeps it live for the whole kernel, which spills.
template <int NP, int TPB, int SCHED_GROUP = 2, class F>
__device__ __forceinline__ void for_each_pair(double2 (&p)[NP], int nf, int nn2, int tid, F&& f) {
    asm volatile("" : "+v"(tid));
    asm volatile("" : "+s"(nf), "+s"(nn2));
    const int rs = (2 * TPB) / nf, cs = (2 * TPB) - rs * nf;
    int r = (2 * tid) / nf;
    int c = 2 * tid - r * nf;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (2 * TPB * j < nn2) {  // wave-uniform: skip register pairs beyond the live matrix
            const int e = 2 * (tid + TPB * j);
            int c1 = c + 1, r1 = r;
            if (c1 == nf) { c1 = 0; r1 = r + 1; }
            f(p[j], r, c, r1, c1, e < nn2, e + 1 < nn2);
        }
        if ((j % SCHED_GROUP) == SCHED_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        c += cs;
        r += rs;
        if (c >= nf) { c -= nf; r += 1; }
    }
}

// PartialPivLU inverse of a 2x2 (MatrixXd::inverse(), ekf.cpp:135); same sequence as the oracle's inv2x2_lu.
__device__ __forceinline__ bool inv2x2_lu(const double S[4], double Si[4]) {
    const bool sw = fabs(S[2]) > fabs(S[0]);
    const double a00 = sw ? S[2] : S[0], a01 = sw ? S[3] : S[1];
    const double a10 = sw ? S[0] : S[2], a11 = sw ? S[1] : S[3];
    const double l = a10 / a00;
    const double u11 = a11 - l * a01;
    const bool ok = (a00 != 0.0) && (u11 != 0.0);
    {   // column 0 of the inverse: rhs = P e_0
        const double r0 = sw ? 0.0 : 1.0, r1 = sw ? 1.0 : 0.0;
        const double y1 = r1 - l * r0;
        const double x1 = y1 / u11;
        Si[0] = (r0 - a01 * x1) / a00;
        Si[2] = x1;
    }
    {   // column 1
        const double r0 = sw ? 1.0 : 0.0, r1 = sw ? 0.0 : 1.0;
        const double y1 = r1 - l * r0;
        const double x1 = y1 / u11;
        Si[1] = (r0 - a01 * x1) / a00;
        Si[3] = x1;
    }
    return ok;
}

template <int NMAX, int W>
__global__ __launch_bounds__(64 * W, (ekf_waves_per_simd(NMAX, W))) void ekf_step_kernel(const EkfStepParams p) {
    using G = EkfGeom<NMAX, W>;
    constexpr int TPB = G::TPB, NP = G::NP, LDP = G::LDP, KCAP = G::KCAP, LMAX = G::LMAX, KG = G::KG, TS = G::TS;

    __shared__ double s_xt[LDP];          // x_t  (posterior of the previous step; landmark positions for H)
    __shared__ double s_xp[LDP];          // x_pred
    __shared__ double s_R[TS * LDP];      // thin rows   R[s][c] = P[T_s][c]
    __shared__ double s_C[TS * LDP];      // thin cols   C[s][r] = P[r][T_s]
    __shared__ double2 s_K[KG * LDP];     // per update of the group: K[r][0..1]
    __shared__ double2 s_HP[KG * LDP];    // per update of the group: (H P)[0..1][c]
    __shared__ double s_r2[LDP], s_c2[LDP];  // row 2 / col 2 of P_t (predict operands)
    __shared__ float s_meas[3 * KCAP];
    __shared__ int s_ids[LMAX > 0 ? LMAX : 1];
    __shared__ int s_didx[KCAP];          // per detection: landmark number (>= M_old: inserted this step), -1 dropped
    __shared__ int s_T[TS];               // thin index set of the current group
    __shared__ signed char s_slot[LDP];   // state index -> thin slot or -1
    __shared__ int s_misc[8];             // k, n_insert, freeze flag, capacity flag

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    int flags = p.flags[b];
    if (flags & SLAM_INST_INDEX_OOR) return;  // frozen instance: the reference node died here (filter.h:5)

    const int M_old = p.M[b];
    const int n_old = 3 + 2 * M_old;
    double* __restrict__ Pb = p.P + (size_t)b * p.pstride;
    double* __restrict__ xb = p.x + (size_t)b * p.xstride;

    // ---- issue the bulk read of P_t first (fast path: no re-layout); everything below overlaps its latency ----
    double2 pr[NP];
    {
        const double2* __restrict__ Pb2 = reinterpret_cast<const double2*>(Pb);
        const int nn2o = n_old * n_old;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int q = tid + TPB * j;
            double2 v = make_double2(0.0, 0.0);
            // (an odd n*n leaves one don't-care element in the last pair; the bulk pass zeroes it)
            if (2 * TPB * j < nn2o && 2 * q < nn2o) v = Pb2[q];
            pr[j] = v;
        }
    }

    #pragma unroll 1
    for (int i = tid; i < LDP; i += TPB) {
        const double v = i < n_old ? xb[i] : 0.0;
        s_xt[i] = v;
        s_xp[i] = v;
    }
    #pragma unroll 1
    for (int i = tid; i < M_old; i += TPB) s_ids[i] = p.ids[(size_t)b * p.L_max + i];
    if (tid < 8) s_misc[tid] = 0;

    // ------------------------------------------------------------------------------------------------------
    // measurements: generate (sim_node.py:209-250) or fetch
    // ------------------------------------------------------------------------------------------------------
    double tx = 0.0, ty = 0.0;
    __syncthreads();
    if (p.sim) {
        const uint64_t inst = (uint64_t)(p.inst0 + b);
        tx = p.truth[3 * (size_t)b];
        ty = p.truth[3 * (size_t)b + 1];
        double tth = p.truth[3 * (size_t)b + 2];
        double u0, u1;
        noise_pair(p.seed, inst, p.step, 0u, &u0, &u1);
        double d = ((double)p.fwd + (2 * p.sV00) * u0) - p.sV00;        // sim_node.py:216
        double hdg = ((double)p.ang + (2 * p.sV11) * u1) - p.sV11;      // :217
        d = (p.d_max < d) ? p.d_max : d;                                 // min(d, d_max)        :219
        d = (0.0 < d) ? d : 0.0;                                         // max(0, .)
        hdg = (p.th_max < hdg) ? p.th_max : hdg;                         // :220
        hdg = (-p.th_max < hdg) ? hdg : -p.th_max;
        double s, c;
        det_sincos(tth, &s, &c);
        tx = tx + d * c;                                                 // :222 (yaw not wrapped)
        ty = ty + d * s;
        tth = tth + hdg;
        if (tid < 64) {  // one wavefront scans the map in ascending id (sim_node.py:231-243)
            int count = 0;
            #pragma unroll 1
            for (int base = 0; base < p.L; base += 64) {
                const int id = base + lane;
                bool vis = false;
                double r = 0.0, beta = 0.0;
                if (id < p.L) {
                    const double dx = p.map[2 * id] - tx, dy = p.map[2 * id + 1] - ty;
                    r = sqrt(dx * dx + dy * dy);
                    const double gb = det_atan2(dy, dx);
                    beta = remainder(gb - tth, kTwoPi);
                    vis = !(r > p.range_max) && (beta > p.fov_min && beta < p.fov_max);
                }
                const unsigned long long mask = __ballot(vis);
                const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
                if (vis && pos < KCAP) {  // noise in visible-id order (sim_node.py:245-249), float32 wire format
                    double v0, v1;
                    noise_pair(p.seed, inst, p.step, (uint32_t)(1 + pos), &v0, &v1);
                    const double rn = (r + (2 * p.sW00) * v0) - p.sW00;
                    const double bn = (beta + (2 * p.sW11) * v1) - p.sW11;
                    s_meas[3 * pos] = (float)id;
                    s_meas[3 * pos + 1] = (float)rn;
                    s_meas[3 * pos + 2] = (float)bn;
                }
                count += __popcll(mask);
            }
            if (lane == 0) s_misc[0] = count < KCAP ? count : KCAP;
        }
        if (tid == 0) {
            p.truth[3 * (size_t)b] = tx;
            p.truth[3 * (size_t)b + 1] = ty;
            p.truth[3 * (size_t)b + 2] = tth;
        }
    } else {
        int kk = p.meas_count_in[b];
        kk = kk < p.k_stride_in ? kk : p.k_stride_in;
        kk = kk < KCAP ? kk : KCAP;
        kk = kk < 0 ? 0 : kk;
        #pragma unroll 1
        for (int i = tid; i < 3 * kk; i += TPB) s_meas[i] = p.meas_in[(size_t)b * p.k_stride_in * 3 + i];
        if (tid == 0) s_misc[0] = kk;
    }
    __syncthreads();
    const int k = s_misc[0];
    if (p.sim && p.meas_out != nullptr) {
        #pragma unroll 1
        for (int i = tid; i < 3 * k && i < 3 * p.k_stride_out; i += TPB)
            p.meas_out[(size_t)b * p.k_stride_out * 3 + i] = s_meas[i];
        if (tid == 0) p.meas_count_out[b] = k;
    }

    // ------------------------------------------------------------------------------------------------------
    // known-id association for the whole message up front (ekf.cpp:99-108): lane l <-> detection l
    // ------------------------------------------------------------------------------------------------------
    int n_ins = 0;  // insertions this step (exact for known ids, upper bound k otherwise)
    if (p.id_known) {
        if (tid < 64) {
            int idx = -1;
            bool isnew = false, dup = false;
            if (lane < k) {
                const int id = (int)s_meas[3 * lane];
                #pragma unroll 1
                for (int j = 0; j < M_old; ++j)
                    if (idx < 0 && s_ids[j] == id) idx = j;
                if (idx < 0) {
                    isnew = true;
                    #pragma unroll 1
                    for (int l2 = 0; l2 < lane; ++l2) dup = dup || ((int)s_meas[3 * l2] == id);
                }
            }
            // a repeated NEW id would be found among the ids pushed this step and index x_t out of range
            // (ekf.cpp:115 -> eigen_assert -> exception, filter.h:5): the reference dies, we freeze.
            const unsigned long long dmask = __ballot(dup);
            const unsigned long long nmask = __ballot(isnew);
            const int rank = __popcll(nmask & ((1ull << lane) - 1ull));
            if (isnew) idx = (M_old + rank < p.L_max && M_old + rank < LMAX) ? M_old + rank : -1;
            if (lane < k) s_didx[lane] = idx;
            if (lane == 0) {
                int cnt = __popcll(nmask);
                const int room = (p.L_max < LMAX ? p.L_max : LMAX) - M_old;
                s_misc[3] = cnt > room ? 1 : 0;           // capacity overflow
                s_misc[1] = cnt > room ? room : cnt;      // insertions
                s_misc[2] = dmask != 0ull ? 1 : 0;        // freeze
            }
        }
        __syncthreads();
        if (s_misc[2]) {
            if (tid == 0) p.flags[b] = flags | SLAM_INST_INDEX_OOR;
            return;
        }
        if (s_misc[3]) flags |= SLAM_INST_CAPACITY;
        n_ins = s_misc[1];
    } else {
        n_ins = k;
    }
    int nf = n_old + 2 * n_ins;
    nf = nf < NMAX ? nf : NMAX;
    const int nn2 = nf * nf;

    if (nf != n_old) {  // the state grows this step: re-lay-out from leading dimension n_old to nf (rare)
#pragma unroll
        for (int j = 0; j < NP; ++j) pr[j] = make_double2(0.0, 0.0);
        for_each_pair<NP, TPB>(pr, nf, nn2, tid, [&](double2& v, int r, int c, int r1, int c1, bool ok0, bool ok1) {
            if (ok0 && r < n_old && c < n_old) v.x = Pb[r * n_old + c];
            if (ok1 && r1 < n_old && c1 < n_old) v.y = Pb[r1 * n_old + c1];
        });
    }

    // x_pred of the vehicle (ekf.cpp:56-59); needed before the first group because unknown-id association
    // (ekf.cpp:82-98) projects detections with the PREDICTED pose.  The covariance part of the prediction runs
    // on the thin rows/cols of the first group.
    if (tid == 0) {
        const float d_d = p.fwd, d_th = p.ang;
        const double th = s_xt[2];
        double s, c;
        det_sincos(th, &s, &c);
        const float dd = d_d + p.v_d;
        s_xp[0] = s_xt[0] + (double)dd * c;
        s_xp[1] = s_xt[1] + (double)dd * s;
        s_xp[2] = remainder((th + (double)d_th) + (double)p.v_th, kTwoPi);
    }

    // ------------------------------------------------------------------------------------------------------
    // groups of <= KG detections
    // ------------------------------------------------------------------------------------------------------
    int M = M_old;
    int na = n_old;      // active dimension
    int l0 = 0;
    bool first = true;
    while (first || l0 < k) {
        // ---- form the group: thread 0 decides, everybody reads ----
        __syncthreads();
        #pragma unroll 1
        for (int i = tid; i < LDP; i += TPB) s_slot[i] = (signed char)-1;
        #pragma unroll 1
        for (int i = tid; i < TS * LDP; i += TPB) { s_R[i] = 0.0; s_C[i] = 0.0; }
        __syncthreads();
        if (tid == 0) {
            int nT = 3, l1 = l0, na_g = na, M_g = M;
            s_T[0] = 0; s_T[1] = 1; s_T[2] = 2;
            s_slot[0] = 0; s_slot[1] = 1; s_slot[2] = 2;
            int frz = 0;
            while (l1 < k && l1 - l0 < KG) {
                int idx;
                if (p.id_known) {
                    idx = s_didx[l1];
                } else if (l1 == l0) {
                    // unknown ids (ekf.cpp:82-98): associate against the CURRENT x_pred, one detection per group
                    const float r_m = s_meas[3 * l1 + 1], b_m = s_meas[3 * l1 + 2];
                    double s, c;
                    det_sincos(s_xp[2] + (double)b_m, &s, &c);
                    const float x_det = (float)(s_xp[0] + (double)r_m * c);
                    const float y_det = (float)(s_xp[1] + (double)r_m * s);
                    idx = -2;
                    #pragma unroll 1
                    for (int j = 0; j < M_g; ++j) {
                        const float xd = (float)fabs((double)x_det - s_xp[3 + 2 * j]);
                        const float yd = (float)fabs((double)y_det - s_xp[3 + 2 * j + 1]);
                        if (xd < p.min_sep && yd < p.min_sep) { idx = j; break; }
                    }
                    if (idx == -2) idx = (M_g < p.L_max && M_g < LMAX && na_g + 2 <= nf) ? M_g : -1;
                    if (idx == -1) s_misc[3] = 1;
                    if (idx >= 0 && idx < M_g && 2 * idx + 4 >= n_old) frz = 1;  // matched a landmark inserted this step
                    s_didx[l1] = idx;
                } else {
                    break;
                }
                if (idx >= 0) {
                    const int ii = 3 + 2 * idx;
                    if (s_slot[ii] < 0) {
                        if (nT + 2 > TS) break;
                        s_T[nT] = ii; s_slot[ii] = (signed char)nT;
                        s_T[nT + 1] = ii + 1; s_slot[ii + 1] = (signed char)(nT + 1);
                        nT += 2;
                    }
                    if (idx >= M_g) { M_g += 1; na_g += 2; }
                }
                l1 += 1;
            }
            s_misc[4] = l1;
            s_misc[5] = nT;
            s_misc[2] = frz;
        }
        __syncthreads();
        const int l1 = s_misc[4], nT = s_misc[5];
        if (s_misc[2]) {
            if (tid == 0) p.flags[b] = flags | SLAM_INST_INDEX_OOR;
            return;
        }
        if (s_misc[3]) flags |= SLAM_INST_CAPACITY;

        // ---- gather: registers -> thin rows / cols ----
        for_each_pair<NP, TPB>(pr, nf, nn2, tid, [&](double2& v, int r, int c, int r1, int c1, bool ok0, bool ok1) {
            const int sr = s_slot[r], sr1 = s_slot[r1], sc = s_slot[c], sc1 = s_slot[c1];
            if (ok0 && sr >= 0) s_R[sr * LDP + c] = v.x;
            if (ok0 && sc >= 0) s_C[sc * LDP + r] = v.x;
            if (ok1 && sr1 >= 0) s_R[sr1 * LDP + c1] = v.y;
            if (ok1 && sc1 >= 0) s_C[sc1 * LDP + r1] = v.y;
        });
        __syncthreads();

        // ---- prediction stage on the thin copies (first group only), ekf.cpp:41-61 ----
        if (first) {
            const float d_d = p.fwd;
            const double th = s_xt[2];
            double s, c;
            det_sincos(th, &s, &c);
            const double fa = (double)(-1 * d_d) * s;  // F_x(0,2)
            const double fb = (double)d_d * c;         // F_x(1,2)
            const float dd = d_d + p.v_d;
            #pragma unroll 1
            for (int i = tid; i < LDP; i += TPB) { s_r2[i] = s_R[2 * LDP + i]; s_c2[i] = s_C[2 * LDP + i]; }
            __syncthreads();
            const double p22 = s_r2[2];
            const double cv = c * p.V00, sv = s * p.V00;
            auto predicted = [&](double t, int r, int cc) -> double {
                const double f_r = r == 0 ? fa : fb;
                if (r < 2) t = t + f_r * s_r2[cc];                 // rows 0,1 of F_x * P
                if (cc < 2) {                                      // cols 0,1 of (F_x P) F_x^T
                    double a2 = s_c2[r];
                    if (r < 2) a2 = a2 + f_r * p22;
                    t = t + a2 * (cc == 0 ? fa : fb);
                }
                if (r == 0 && cc == 0) t = t + cv * c;             // + F_v V F_v^T
                if (r == 0 && cc == 1) t = t + cv * s;
                if (r == 1 && cc == 0) t = t + sv * c;
                if (r == 1 && cc == 1) t = t + sv * s;
                if (r == 2 && cc == 2) t = t + p.V11;
                return t;
            };
            #pragma unroll 1
            for (int i = tid; i < nT * LDP; i += TPB) {
                const int sl = i / LDP, j = i - sl * LDP;
                const int t_s = s_T[sl];
                if (j < na && t_s < na) {
                    if (t_s < 2 || j < 2 || (t_s == 2 && j == 2)) {
                        s_R[i] = predicted(s_R[i], t_s, j);     // R[sl][j] = P[t_s][j]
                        s_C[i] = predicted(s_C[i], j, t_s);     // C[sl][j] = P[j][t_s]
                    }
                }
            }
            __syncthreads();
        }

        // ---- detections of the group in message order ----
        int nu = 0;  // updates recorded for the bulk pass
        #pragma unroll 1
        for (int l = l0; l < l1; ++l) {
            const int idx = s_didx[l];
            if (idx < 0) continue;  // dropped (capacity)
            const float r_m = s_meas[3 * l + 1], b_m = s_meas[3 * l + 2];
            const int ii = 3 + 2 * idx;
            if (idx < M) {
                // ---------------- landmark update, ekf.cpp:110-140 ----------------
                const int si = s_slot[ii];
                const double dx = s_xt[ii] - s_xp[0], dy = s_xt[ii + 1] - s_xp[1];
                const float dist = (float)sqrt(dx * dx + dy * dy);
                const double dd = (double)dist, d2 = (double)(dist * dist);
                const double h00 = -dx / dd, h01 = -dy / dd, h03 = dx / dd, h04 = dy / dd;
                const double h10 = dy / d2, h11 = -dx / d2, h12 = -1.0, h13 = -dy / d2, h14 = dx / d2;
                const float angf = (float)remainder(det_atan2(dy, dx) - s_xp[2], kTwoPi);
                const float nu0f = r_m - dist - p.w_r;
                const float nu1f = b_m - angf - p.w_b;
                const double nu0 = (double)nu0f, nu1 = (double)nu1f;
                double2* __restrict__ HPu = s_HP + nu * LDP;
                double2* __restrict__ Ku = s_K + nu * LDP;
                const double* Ri = s_R + si * LDP;
                const double* Rj = s_R + (si + 1) * LDP;
                const double* Ci = s_C + si * LDP;
                const double* Cj = s_C + (si + 1) * LDP;
                double2 pht[(LDP + TPB - 1) / TPB];
#pragma unroll
                for (int u = 0; u < (LDP + TPB - 1) / TPB; ++u) {
                    const int c = tid + TPB * u;
                    double2 hp = make_double2(0.0, 0.0), ph = make_double2(0.0, 0.0);
                    if (c < na) {
                        const double p0 = s_R[c], p1 = s_R[LDP + c], p2 = s_R[2 * LDP + c], pi = Ri[c], pj = Rj[c];
                        hp.x = ((h00 * p0 + h01 * p1) + h03 * pi) + h04 * pj;
                        hp.y = (((h10 * p0 + h11 * p1) + h12 * p2) + h13 * pi) + h14 * pj;
                        const double q0 = s_C[c], q1 = s_C[LDP + c], q2 = s_C[2 * LDP + c], qi = Ci[c], qj = Cj[c];
                        ph.x = ((q0 * h00 + q1 * h01) + qi * h03) + qj * h04;
                        ph.y = (((q0 * h10 + q1 * h11) + q2 * h12) + qi * h13) + qj * h14;
                    }
                    if (c < LDP) HPu[c] = hp;
                    pht[u] = ph;
                }
                __syncthreads();
                double S[4], Si[4];
                {
                    const double2 g0 = HPu[0], g1 = HPu[1], g2 = HPu[2], gi = HPu[ii], gj = HPu[ii + 1];
                    S[0] = ((g0.x * h00 + g1.x * h01) + gi.x * h03) + gj.x * h04;
                    S[1] = (((g0.x * h10 + g1.x * h11) + g2.x * h12) + gi.x * h13) + gj.x * h14;
                    S[2] = ((g0.y * h00 + g1.y * h01) + gi.y * h03) + gj.y * h04;
                    S[3] = (((g0.y * h10 + g1.y * h11) + g2.y * h12) + gi.y * h13) + gj.y * h14;
                    S[0] = S[0] + p.W00;
                    S[3] = S[3] + p.W11;
                }
                if (!inv2x2_lu(S, Si)) flags |= SLAM_INST_S_SINGULAR;
#pragma unroll
                for (int u = 0; u < (LDP + TPB - 1) / TPB; ++u) {
                    const int r = tid + TPB * u;
                    double2 kk = make_double2(0.0, 0.0);
                    if (r < na) {
                        kk.x = pht[u].x * Si[0] + pht[u].y * Si[2];
                        kk.y = pht[u].x * Si[1] + pht[u].y * Si[3];
                        double xv = s_xp[r] + (kk.x * nu0 + kk.y * nu1);
                        if (r == 2) xv = remainder(xv, kTwoPi);
                        s_xp[r] = xv;
                    }
                    if (r < LDP) Ku[r] = kk;
                }
                __syncthreads();
                // thin copies follow the same downdate  P -= K (H P)
                #pragma unroll 1
                for (int i = tid; i < nT * LDP; i += TPB) {
                    const int sl = i / LDP, j = i - sl * LDP;
                    const int t_s = s_T[sl];
                    if (j < na && t_s < na) {
                        const double2 kt = Ku[t_s], hj = HPu[j], kj = Ku[j], ht = HPu[t_s];
                        s_R[i] = s_R[i] - (kt.x * hj.x + kt.y * hj.y);   // P[t_s][j]
                        s_C[i] = s_C[i] - (kj.x * ht.x + kj.y * ht.y);   // P[j][t_s]
                    }
                }
                nu += 1;
                __syncthreads();
            } else {
                // ---------------- landmark insertion, ekf.cpp:141-173 ----------------
                const int sa = s_slot[ii], sb = sa + 1;
                const int no = na;
                const double phi = s_xp[2] + (double)b_m;
                double s, c;
                det_sincos(phi, &s, &c);
                const double rd = (double)r_m;
                const double g02 = -rd * s, g12 = rd * c;
                const double lx = s_xp[0] + rd * c, ly = s_xp[1] + rd * s;
                // new rows G_x P[0:3,:] and new cols P[:,0:3] G_x^T
                #pragma unroll 1
                for (int j = tid; j < no; j += TPB) {
                    s_R[sa * LDP + j] = s_R[j] + g02 * s_R[2 * LDP + j];
                    s_R[sb * LDP + j] = s_R[LDP + j] + g12 * s_R[2 * LDP + j];
                    s_C[sa * LDP + j] = s_C[j] + s_C[2 * LDP + j] * g02;
                    s_C[sb * LDP + j] = s_C[LDP + j] + s_C[2 * LDP + j] * g12;
                }
                __syncthreads();
                if (tid == 0) {  // corner: (G_x P_vv) G_x^T + (G_z W) G_z^T
                    const double gw00 = c * p.W00, gw01 = g02 * p.W11;   // (G_z W) row 0
                    const double gw10 = s * p.W00, gw11 = g12 * p.W11;   // (G_z W) row 1
                    const double* Ra = s_R + sa * LDP;
                    const double* Rb = s_R + sb * LDP;
                    const double v00 = ((Ra[0] + Ra[2] * g02) + gw00 * c) + gw01 * g02;
                    const double v01 = ((Ra[1] + Ra[2] * g12) + gw00 * s) + gw01 * g12;
                    const double v10 = ((Rb[0] + Rb[2] * g02) + gw10 * c) + gw11 * g02;
                    const double v11 = ((Rb[1] + Rb[2] * g12) + gw10 * s) + gw11 * g12;
                    s_R[sa * LDP + no] = v00; s_R[sa * LDP + no + 1] = v01;
                    s_R[sb * LDP + no] = v10; s_R[sb * LDP + no + 1] = v11;
                    s_C[sa * LDP + no] = v00; s_C[sa * LDP + no + 1] = v10;
                    s_C[sb * LDP + no] = v01; s_C[sb * LDP + no + 1] = v11;
                    s_xp[no] = lx;
                    s_xp[no + 1] = ly;
                    s_ids[M] = p.id_known ? (int)s_meas[3 * l] : M;
                }
                if (tid >= 64 - TS && tid < 64) {  // cross entries of the other thin rows / cols
                    const int sl = tid - (64 - TS);
                    if (sl < nT && sl != sa && sl != sb) {
                        const int t_s = s_T[sl];
                        if (t_s < no) {
                            s_R[sl * LDP + no] = s_C[sa * LDP + t_s];       // P[t_s][no]
                            s_R[sl * LDP + no + 1] = s_C[sb * LDP + t_s];   // P[t_s][no+1]
                            s_C[sl * LDP + no] = s_R[sa * LDP + t_s];       // P[no][t_s]
                            s_C[sl * LDP + no + 1] = s_R[sb * LDP + t_s];   // P[no+1][t_s]
                        }
                    }
                }
                M += 1;
                na += 2;
                __syncthreads();
            }
        }

        // ---- bulk: apply the group's rank-2 downdates in order, then scatter the thin rows / cols back ----
        for_each_pair<NP, TPB>(pr, nf, nn2, tid, [&](double2& v, int r, int c, int r1, int c1, bool ok0, bool ok1) {
            double vx = v.x, vy = v.y;
            #pragma unroll
            for (int u = 0; u < KG; ++u) {
                if (u >= nu) break;  // wave-uniform
                const double2 k0 = s_K[u * LDP + r], k1 = s_K[u * LDP + r1];
                const double2 h0 = s_HP[u * LDP + c], h1 = s_HP[u * LDP + c1];
                vx = vx - (k0.x * h0.x + k0.y * h0.y);
                vy = vy - (k1.x * h1.x + k1.y * h1.y);
            }
            const int sr = s_slot[r], sr1 = s_slot[r1], sc = s_slot[c], sc1 = s_slot[c1];
            if (sc >= 0) vx = s_C[sc * LDP + r];
            if (sr >= 0) vx = s_R[sr * LDP + c];
            if (sc1 >= 0) vy = s_C[sc1 * LDP + r1];
            if (sr1 >= 0) vy = s_R[sr1 * LDP + c1];
            v.x = ok0 ? vx : 0.0;
            v.y = ok1 ? vy : 0.0;
        });
        l0 = l1;
        first = false;
    }
    __syncthreads();

    // ------------------------------------------------------------------------------------------------------
    // x_t = x_pred ; P_t = P_pred (ekf.cpp:176-177): the only HBM write of P this step
    // ------------------------------------------------------------------------------------------------------
    unsigned hiacc = 0u;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const unsigned h0 = (unsigned)(__double_as_longlong(pr[j].x) >> 32) & 0x7fffffffu;
        const unsigned h1 = (unsigned)(__double_as_longlong(pr[j].y) >> 32) & 0x7fffffffu;
        hiacc = hiacc > h0 ? hiacc : h0;
        hiacc = hiacc > h1 ? hiacc : h1;
    }
    #pragma unroll 1
    for (int i = tid; i < na; i += TPB) {
        const double v = s_xp[i];
        xb[i] = v;
        const unsigned h0 = (unsigned)(__double_as_longlong(v) >> 32) & 0x7fffffffu;
        hiacc = hiacc > h0 ? hiacc : h0;
    }
    const bool nonfinite = __syncthreads_or(hiacc >= 0x7ff00000u);
    if (nonfinite) flags |= SLAM_INST_NONFINITE;

    if (na == nf) {
        double2* __restrict__ Pb2 = reinterpret_cast<double2*>(Pb);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int q = tid + TPB * j;
            if (2 * TPB * j < nn2 && 2 * q < nn2) Pb2[q] = pr[j];
        }
    } else {  // fewer insertions than provisioned (unknown-id mode): store with the final leading dimension
        for_each_pair<NP, TPB>(pr, nf, nn2, tid, [&](double2& v, int r, int c, int r1, int c1, bool ok0, bool ok1) {
            if (ok0 && r < na && c < na) Pb[r * na + c] = v.x;
            if (ok1 && r1 < na && c1 < na) Pb[r1 * na + c1] = v.y;
        });
    }
    if (M != M_old) {
        #pragma unroll 1
        for (int i = tid; i < M; i += TPB) p.ids[(size_t)b * p.L_max + i] = s_ids[i];
    }
    if (tid == 0) {
        p.M[b] = M;
        p.flags[b] = flags;
        p.timestep[b] = p.timestep[b] + 1;
        if (p.sim) {  // plotting_node.py:209-212 with the float32 wire format of EKFState.x_v / y_v
            const double ex = (double)(float)s_xp[0] - tx, ey = (double)(float)s_xp[1] - ty;
            p.err_sum[b] = p.err_sum[b] + sqrt(ex * ex + ey * ey);
        }
    }
}

template <int NMAX, int W>
hipError_t launch_variant(const EkfStepParams& p, hipStream_t stream) {
    hipLaunchKernelGGL((ekf_step_kernel<NMAX, W>), dim3(p.B), dim3(64 * W), 0, stream, p);
    return hipGetLastError();
}

}  // namespace slam
