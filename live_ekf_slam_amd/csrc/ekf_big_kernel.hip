// ekf_big_kernel.hip — EKF::update (reference ekf_ws/src/localization_pkg/src/ekf.cpp:37-179) for states that do NOT fit the LDS size
// classes of ekf_kernel_impl.h (n = 3 + 2 L > 403, up to kEkfBigMaxLandmarks landmarks).  The reference grows its state by two per new
// landmark without limit (ekf.cpp:144-146); the fused kernel keeps an instance's thin rows / columns and its K / H P ring in LDS and
// therefore ends at 200 landmarks.  This kernel is the size class beyond: one workgroup per instance, ONE timestep per launch, the
// covariance streamed through HBM / L2 for every phase.  It follows the reference's loop detection by detection (association, update or
// insertion, in message order) - which also means a message may hold ANY number of detections here (read from the caller's buffer as it is
// walked; no per-message capacity) - and evaluates every element with the same expressions in the same order as the fused kernel and the
// oracle's MODE_FAST, so results are bit-identical to both.  It is slow by design (a pass over P per detection instead of one per group of
// deferred updates, a barrier per phase); SLAM_ERR_UNSUPPORTED for a legal reference configuration was the alternative.
//
// Working matrix: the step works on a copy of P in the handle's second buffer with the FIXED leading dimension ekf_ld(3 + 2 L_max) (no
// re-layout when landmarks are inserted); the result is compacted into the first buffer with the leading dimension of the new state size,
// the layout every getter expects.  An instance that freezes (filter.h:5: the reference dies on an out-of-range index) simply does not
// write anything back: its pre-step state is still in the first buffer.
#include "ekf_kernel.h"

#include <stdio.h>
#include <string.h>

#include "../../include/slam_batch.h"
#include "lds_attr.h"
#include "sim_device.h"
#include "slam_math.h"
#include "slam_rng.h"

namespace slam {

namespace {

constexpr int kBigTpb = 1024;

__device__ __forceinline__ bool big_inv2x2_lu(const double S[4], double Si[4]) {   // PartialPivLU inverse (MatrixXd::inverse(), ekf.cpp:135)
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

// LDS layout (dynamic): x_t [np], x_pred [np], K / P H^T [2 np] (interleaved per state index), H P [2 np] (row 0, row 1), scalars [32],
// ints [16], new ids [L_max], message [3 L] floats (SIM mode)
// ST = storage type of x_t and P_t.  double: the streamed size class (L_max > 200) and the long messages of the fp64 LDS classes.  float
// (round 5): the long messages of the fp32-storage classes - x_t / P_t are read as floats, the timestep runs in fp64 in the working matrix
// (the handle's fp64 slab `scratch`, [B][pstride] doubles: n x ld8(n) fits since ld4(n) >= ld8(n)) and the result is rounded to float where it
// is stored: the oracle's STORAGE_F32 (slam_oracle.cpp round_storage: once per timestep), which is what the fp32 LDS kernel computes too.
template <class ST>
__global__ __launch_bounds__(kBigTpb) void ekf_big_step_kernel(const EkfStepParams p, const int t_off, const int multi) {
    constexpr int ESZ = (int)sizeof(ST);
    extern __shared__ double sm[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int Lcap = p.L_max, nmax = 3 + 2 * Lcap, np = (nmax + 1) & ~1;
    double* const s_xt = sm;
    double* const s_xp = s_xt + np;
    double* const s_K = s_xp + np;          // [r][2]
    double* const s_HP = s_K + 2 * np;      // [2][np]
    double* const s_sc = s_HP + 2 * np;     // 32 scalars
    int* const s_i = reinterpret_cast<int*>(s_sc + 32);   // 16 ints: 0 association result, 1 freeze, 2 flags raised, 3 detections, 4 non-finite
    int* const s_newid = s_i + 16;                        // ids of the landmarks inserted by this message
    float* const s_meas = reinterpret_cast<float*>(s_newid + ((Lcap + 1) & ~1));

    if (p.long_mode == 2) {   // paired with the LDS kernel's launch: only the instances whose message that kernel cannot hold (ekf_kernel.h)
        const int kk = p.meas_count_in[b];
        if ((kk < p.k_stride_in ? kk : p.k_stride_in) <= p.long_cap) return;
    }
    int flags = p.flags[b];
    if (flags & SLAM_INST_INDEX_OOR) return;   // frozen instance: the state stays as it is
    const int M0 = p.M[b], n0 = 3 + 2 * M0;
    const int ldw = ekf_ld(nmax, 8);           // leading dimension of the working matrix
    const int ld0 = ekf_ld(n0, ESZ);
    const ST* PA = static_cast<const ST*>(p.P) + (size_t)b * p.pstride;
    double* __restrict__ PB = (ESZ == 8 ? static_cast<double*>(p.P_out) : p.scratch) + (size_t)b * p.pstride;
    ST* __restrict__ xb = static_cast<ST*>(p.x) + (size_t)b * p.xstride;
    const int* __restrict__ idsb = p.ids + (size_t)b * p.L_max;

    for (int i = tid; i < np; i += kBigTpb) {
        const double v = i < n0 ? (double)xb[i] : 0.0;
        s_xt[i] = v; s_xp[i] = v;
    }
    if (tid < 16) s_i[tid] = 0;
    const float fwd = multi ? p.cmds[2 * t_off] : p.fwd;
    const float ang = multi ? p.cmds[2 * t_off + 1] : p.ang;
    // ---- the message of this timestep ----
    double tx = 0.0, ty = 0.0, tth = 0.0;
    const float* meas = s_meas;
    if (p.sim) {
        if (tid < 64) {
            tx = p.truth[3 * (size_t)b]; ty = p.truth[3 * (size_t)b + 1]; tth = p.truth[3 * (size_t)b + 2];
            const double lmx0 = lane < p.L ? p.map[2 * lane] : 0.0, lmy0 = lane < p.L ? p.map[2 * lane + 1] : 0.0;
            const int kr = sim_wave<(1 << 30), false>(p, b, lane, fwd, ang, p.step + (uint32_t)t_off, tx, ty, tth, lmx0, lmy0, s_meas);
            if (lane == 0) { s_i[3] = kr; s_sc[24] = tx; s_sc[25] = ty; s_sc[26] = tth; }
        }
    } else {
        int kk = p.meas_count_in[(size_t)t_off * p.B + b];
        kk = kk < p.k_stride_in ? kk : p.k_stride_in;
        kk = kk < 0 ? 0 : kk;
        if (tid == 0) s_i[3] = kk;
        meas = p.meas_in + ((size_t)t_off * p.B + b) * p.k_stride_in * 3;   // walked where it lies: no per-message capacity
    }
    __syncthreads();
    const int k = s_i[3];
    if (tid == 0 && p.khist != nullptr) atomicAdd(&p.khist[k < 7 ? k : 7], 1ull);
    if (p.sim && p.meas_out != nullptr) {
        for (int i = tid; i < 3 * k && i < 3 * p.k_stride_out; i += kBigTpb) p.meas_out[(size_t)b * p.k_stride_out * 3 + i] = s_meas[i];
        if (tid == 0) p.meas_count_out[b] = k < p.k_stride_out ? k : p.k_stride_out;
    }

    // ---- prediction (ekf.cpp:41-61): x_pred of the vehicle, P_pred = F_x P F_x^T + F_v V F_v^T into the working matrix ----
    {
        const double th = s_xt[2];
        double sn, cs;
        det_sincos(th, &sn, &cs);
        const double fa = (double)(-1 * fwd) * sn;   // F_x(0,2)
        const double fb = (double)fwd * cs;          // F_x(1,2)
        const float dd = fwd + p.v_d;
        const double cv = cs * p.V00, sv = sn * p.V00;
        const double q00 = cv * cs, q01 = cv * sn, q10 = sv * cs, q11 = sv * sn;
        __syncthreads();
        if (tid == 0) {
            s_xp[0] = s_xt[0] + (double)dd * cs;
            s_xp[1] = s_xt[1] + (double)dd * sn;
            s_xp[2] = rem2pi((th + (double)ang) + (double)p.v_th);
        }
        const double p22 = (double)PA[(size_t)2 * ld0 + 2];
        for (int e = tid; e < n0 * n0; e += kBigTpb) {
            const int r = e / n0, c = e - r * n0;
            double t = (double)PA[(size_t)r * ld0 + c];
            const double f_r = r == 0 ? fa : fb;
            if (r < 2) t = t + f_r * (double)PA[(size_t)2 * ld0 + c];   // rows 0, 1 of F_x P
            if (c < 2) {                                               // cols 0, 1 of (F_x P) F_x^T
                double a2 = (double)PA[(size_t)r * ld0 + 2];
                if (r < 2) a2 = a2 + f_r * p22;
                t = t + a2 * (c == 0 ? fa : fb);
            }
            if (r < 2 && c < 2) t = t + (r == 0 ? (c == 0 ? q00 : q01) : (c == 0 ? q10 : q11));   // + F_v V F_v^T
            if (r == 2 && c == 2) t = t + p.V11;
            PB[(size_t)r * ldw + c] = t;
        }
    }
    __syncthreads();

    int M = M0, n = n0;
    // ---- detections in message order (ekf.cpp:73): association, then landmark update or insertion ----
#pragma unroll 1
    for (int l = 0; l < k; ++l) {
        const float idf = meas[3 * l], r_m = meas[3 * l + 1], b_m = meas[3 * l + 2];
        // association: first match wins (ekf.cpp:82-108); every thread scans a stride of the landmarks, the lowest hit is kept
        if (tid == 0) s_i[0] = 0x7fffffff;
        __syncthreads();
        int id = M;
        {
            int hit = 0x7fffffff;
            if (p.id_known) {
                id = (int)idf;
                for (int j = tid; j < M; j += kBigTpb) {
                    const int idj = j < M0 ? idsb[j] : s_newid[j - M0];
                    if (idj == id) { hit = j; break; }
                }
            } else {
                double s, c;
                det_sincos(s_xp[2] + (double)b_m, &s, &c);
                const float x_det = (float)(s_xp[0] + (double)r_m * c);
                const float y_det = (float)(s_xp[1] + (double)r_m * s);
                for (int j = tid; j < M; j += kBigTpb) {
                    const float xd = assoc_abs((double)x_det - s_xp[3 + 2 * j], p.abs_is_int);
                    const float yd = assoc_abs((double)y_det - s_xp[3 + 2 * j + 1], p.abs_is_int);
                    if (xd < p.min_sep && yd < p.min_sep) { hit = j; break; }
                }
            }
            if (hit != 0x7fffffff) atomicMin(&s_i[0], hit);
        }
        __syncthreads();
        const int i = s_i[0] == 0x7fffffff ? -1 : s_i[0];
        if (i >= 0) {
            // ---------------- landmark update, ekf.cpp:110-140 ----------------
            const int ii = 2 * i + 3;
            if (ii + 1 >= n0) {   // x_t(ii) out of range (a landmark this message inserted): the reference throws, the instance freezes
                if (tid == 0) s_i[1] = 1;
                __syncthreads();
                break;
            }
            if (tid == 0) {   // the scalar chain: Jacobian entries with the reference's float truncations, innovation
                const double* const xl = p.lm_from_pred ? s_xp : s_xt;   // quirk D-2
                const double dx = xl[ii] - s_xp[0], dy = xl[ii + 1] - s_xp[1];
                const float dist = (float)sqrt(dx * dx + dy * dy);
                const double dd = (double)dist, d2 = (double)(dist * dist);
                s_sc[0] = -dx / dd; s_sc[1] = -dy / dd; s_sc[2] = dx / dd; s_sc[3] = dy / dd;           // H0 at columns 0, 1, ii, ii+1
                s_sc[4] = dy / d2; s_sc[5] = -dx / d2; s_sc[6] = -dy / d2; s_sc[7] = dx / d2;           // H1 at columns 0, 1, ii, ii+1 (H1[2] = -1)
                const float angf = (float)rem2pi(det_atan2(dy, dx) - s_xp[2]);
                const float nu0f = r_m - dist - p.w_r;
                const float nu1f = b_m - angf - p.w_b;
                s_sc[8] = (double)nu0f; s_sc[9] = (double)nu1f;
            }
            __syncthreads();
            const double h00 = s_sc[0], h01 = s_sc[1], h03 = s_sc[2], h04 = s_sc[3];
            const double h10 = s_sc[4], h11 = s_sc[5], h12 = -1.0, h13 = s_sc[6], h14 = s_sc[7];
            for (int c = tid; c < n; c += kBigTpb) {   // H P (rows of P) and P H^T (columns of P)
                const double p0 = PB[c], p1 = PB[(size_t)ldw + c], p2 = PB[(size_t)2 * ldw + c];
                const double pi = PB[(size_t)ii * ldw + c], pj = PB[(size_t)(ii + 1) * ldw + c];
                s_HP[c] = ((h00 * p0 + h01 * p1) + h03 * pi) + h04 * pj;
                s_HP[np + c] = (((h10 * p0 + h11 * p1) + h12 * p2) + h13 * pi) + h14 * pj;
                const double* pr = PB + (size_t)c * ldw;
                const double q0 = pr[0], q1 = pr[1], q2 = pr[2], qi = pr[ii], qj = pr[ii + 1];
                s_K[2 * c] = ((q0 * h00 + q1 * h01) + qi * h03) + qj * h04;
                s_K[2 * c + 1] = (((q0 * h10 + q1 * h11) + q2 * h12) + qi * h13) + qj * h14;
            }
            __syncthreads();
            if (tid == 0) {   // S = (H P) H^T + W and its inverse (ekf.cpp:133-135)
                const double* g0 = s_HP;
                const double* g1 = s_HP + np;
                double S[4], Si[4];
                S[0] = ((g0[0] * h00 + g0[1] * h01) + g0[ii] * h03) + g0[ii + 1] * h04;
                S[1] = (((g0[0] * h10 + g0[1] * h11) + g0[2] * h12) + g0[ii] * h13) + g0[ii + 1] * h14;
                S[2] = ((g1[0] * h00 + g1[1] * h01) + g1[ii] * h03) + g1[ii + 1] * h04;
                S[3] = (((g1[0] * h10 + g1[1] * h11) + g1[2] * h12) + g1[ii] * h13) + g1[ii + 1] * h14;
                S[0] = S[0] + p.W00;
                S[3] = S[3] + p.W11;
                if (!big_inv2x2_lu(S, Si)) s_i[2] |= SLAM_INST_S_SINGULAR;
                s_sc[10] = Si[0]; s_sc[11] = Si[1]; s_sc[12] = Si[2]; s_sc[13] = Si[3];
            }
            __syncthreads();
            {
                const double si0 = s_sc[10], si1 = s_sc[11], si2 = s_sc[12], si3 = s_sc[13], nu0 = s_sc[8], nu1 = s_sc[9];
                for (int r = tid; r < n; r += kBigTpb) {   // K = (P H^T) S^-1, x_pred += K nu
                    const double a = s_K[2 * r], bb = s_K[2 * r + 1];
                    const double k0 = a * si0 + bb * si2, k1 = a * si1 + bb * si3;
                    s_K[2 * r] = k0; s_K[2 * r + 1] = k1;
                    double xv = s_xp[r] + (k0 * nu0 + k1 * nu1);
                    if (r == 2) xv = rem2pi(xv);
                    s_xp[r] = xv;
                }
            }
            __syncthreads();
            {   // P_pred -= K (H P): one pass over the working matrix, a lane owns a 16-byte column pair
                typedef double dbl2_t __attribute__((ext_vector_type(2)));
                const int nv = (n + 1) >> 1;
                for (int e = tid; e < n * nv; e += kBigTpb) {
                    const int r = e / nv, j = e - r * nv, c = 2 * j;
                    const double k0 = s_K[2 * r], k1 = s_K[2 * r + 1];
                    dbl2_t v = *reinterpret_cast<dbl2_t*>(PB + (size_t)r * ldw + c);
                    v.x = v.x - (k0 * s_HP[c] + k1 * s_HP[np + c]);
                    if (c + 1 < n) v.y = v.y - (k0 * s_HP[c + 1] + k1 * s_HP[np + c + 1]);
                    *reinterpret_cast<dbl2_t*>(PB + (size_t)r * ldw + c) = v;
                }
            }
            __syncthreads();
        } else {
            // ---------------- landmark insertion, ekf.cpp:141-173 ----------------
            if (M >= Lcap) {   // no room: skipped (the reference has no capacity; here SLAM_INST_CAPACITY)
                if (tid == 0) s_i[2] |= SLAM_INST_CAPACITY;
                __syncthreads();   // (everybody has read this detection's association result before the next one resets it)
                continue;
            }
            const int no = n;
            const double phi = s_xp[2] + (double)b_m;
            double s, c;
            det_sincos(phi, &s, &c);
            const double rd = (double)r_m;
            const double g02 = -rd * s, g12 = rd * c;
            __syncthreads();
            if (tid == 0) {
                s_xp[no] = s_xp[0] + rd * c;
                s_xp[no + 1] = s_xp[1] + rd * s;
                s_newid[M - M0] = p.id_known ? id : M;
            }
            for (int j = tid; j < no; j += kBigTpb) {   // new rows G_x P[0:3, :] and new columns P[:, 0:3] G_x^T
                const double r0 = PB[j], r1 = PB[(size_t)ldw + j], r2 = PB[(size_t)2 * ldw + j];
                PB[(size_t)no * ldw + j] = r0 + g02 * r2;
                PB[(size_t)(no + 1) * ldw + j] = r1 + g12 * r2;
                const double* pr = PB + (size_t)j * ldw;
                const double c0 = pr[0], c1 = pr[1], c2 = pr[2];
                PB[(size_t)j * ldw + no] = c0 + c2 * g02;
                PB[(size_t)j * ldw + no + 1] = c1 + c2 * g12;
            }
            __syncthreads();
            if (tid == 0) {   // corner: (G_x P_vv) G_x^T + (G_z W) G_z^T
                const double gw00 = c * p.W00, gw01 = g02 * p.W11;
                const double gw10 = s * p.W00, gw11 = g12 * p.W11;
                const double* Ra = PB + (size_t)no * ldw;
                const double* Rb = PB + (size_t)(no + 1) * ldw;
                const double v00 = ((Ra[0] + Ra[2] * g02) + gw00 * c) + gw01 * g02;
                const double v01 = ((Ra[1] + Ra[2] * g12) + gw00 * s) + gw01 * g12;
                const double v10 = ((Rb[0] + Rb[2] * g02) + gw10 * c) + gw11 * g02;
                const double v11 = ((Rb[1] + Rb[2] * g12) + gw10 * s) + gw11 * g12;
                PB[(size_t)no * ldw + no] = v00; PB[(size_t)no * ldw + no + 1] = v01;
                PB[(size_t)(no + 1) * ldw + no] = v10; PB[(size_t)(no + 1) * ldw + no + 1] = v11;
            }
            M += 1;
            n += 2;
            __syncthreads();
        }
    }
    __syncthreads();
    flags |= s_i[2];
    if (s_i[1]) {   // frozen in the pre-step state: nothing of this timestep is written (filter.h:5)
        if (tid == 0) p.flags[b] = flags | SLAM_INST_INDEX_OOR;
        return;
    }
    // ---- x_t = x_pred, P_t = P_pred (ekf.cpp:176-177): compact the working matrix into the layout of the new state size ----
    const int ldn = ekf_ld(n, ESZ);
    ST* PAw = const_cast<ST*>(PA);
    int bad = 0;
    for (int e = tid; e < n * ldn; e += kBigTpb) {
        const int r = e / ldn, c = e - r * ldn;
        const ST v = (ST)(c < n ? PB[(size_t)r * ldw + c] : 0.0);   // pad columns stay zero
        bad |= !isfinite(v);
        PAw[e] = v;
    }
    for (int i = tid; i < n; i += kBigTpb) {
        const ST v = (ST)s_xp[i];
        bad |= !isfinite(v);
        xb[i] = v;
    }
    if (bad) s_i[4] = 1;
    for (int q = tid; q < M - M0; q += kBigTpb) p.ids[(size_t)b * p.L_max + M0 + q] = s_newid[q];
    __syncthreads();
    if (tid == 0) {
        if (s_i[4]) flags |= SLAM_INST_NONFINITE;
        p.M[b] = M;
        p.flags[b] = flags;
        p.timestep[b] = p.timestep[b] + 1;
        if (p.sim) {   // plotting_node.py:209-212 with the float32 wire format of EKFState.x_v / y_v
            const double ex = (double)(float)s_xp[0] - s_sc[24], ey = (double)(float)s_xp[1] - s_sc[25];
            p.err_sum[b] = p.err_sum[b] + sqrt(ex * ex + ey * ey);
            p.truth[3 * (size_t)b] = s_sc[24]; p.truth[3 * (size_t)b + 1] = s_sc[25]; p.truth[3 * (size_t)b + 2] = s_sc[26];
        }
    }
}

size_t big_lds_bytes(int L_max, int L_map) {
    const int nmax = 3 + 2 * L_max, np = (nmax + 1) & ~1;
    return sizeof(double) * (size_t)(6 * np + 32) + sizeof(int) * (size_t)(16 + ((L_max + 1) & ~1)) + sizeof(float) * 3 * (size_t)(L_map > 1 ? L_map : 1) + 16;
}

}  // namespace

hipError_t launch_ekf_big_step(const EkfStepParams& p, hipStream_t stream, int f32_storage) {
    const size_t lds = big_lds_bytes(p.L_max, p.sim ? p.L : 1);
    if (lds > 160 * 1024) return hipErrorInvalidConfiguration;
    if (f32_storage && p.scratch == nullptr) return hipErrorInvalidValue;   // the fp64 working matrix of an fp32-storage handle
    const void* fn = f32_storage ? reinterpret_cast<const void*>(&ekf_big_step_kernel<float>) : reinterpret_cast<const void*>(&ekf_big_step_kernel<double>);
    if (lds > 64 * 1024) {   // once per device, to the kernel's maximum (lds_attr.h)
        const hipError_t e = slam_allow_full_lds(fn);
        if (e != hipSuccess) return e;
    }
    const int multi = (p.cmds != nullptr && p.T > 1) ? 1 : 0;
    const int T = multi ? p.T : 1;
    for (int t = 0; t < T; ++t) {   // one launch per timestep: the kernel keeps nothing on chip between steps
        if (f32_storage) hipLaunchKernelGGL(ekf_big_step_kernel<float>, dim3(p.B), dim3(kBigTpb), lds, stream, p, t, multi);
        else hipLaunchKernelGGL(ekf_big_step_kernel<double>, dim3(p.B), dim3(kBigTpb), lds, stream, p, t, multi);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t ekf_big_kernel_info(EkfKernelInfo* out) {
    hipFuncAttributes a;
    const hipError_t e = hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&ekf_big_step_kernel<double>));
    if (e != hipSuccess) return e;
    snprintf(out->name, sizeof(out->name), "ekf_big_step_kernel");
    out->lds_bytes = (int)a.sharedSizeBytes;
    out->vgprs = a.numRegs;
    out->sgprs = 0;
    out->threads = kBigTpb;
    out->wg_per_cu = 1;
    return hipSuccess;
}

}  // namespace slam
