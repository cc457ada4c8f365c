// pgs_kernel.hip — batched pose-graph SLAM for gfx950 (MI355X): graph building + Levenberg–Marquardt solve of B
// independent graphs (Monte-Carlo instances over one map / command sequence), one workgroup per instance.
//
// Reference path: PoseGraph::{init, updateNaiveVehPoseEstimate, update, onLandmarkMeasurement, solvePoseGraph}
// (ekf_ws/src/localization_pkg/src/pose_graph.cpp:68-300) with `implementation: gtsam` (params.yaml:61).  The solve
// is gtsam::LevenbergMarquardtOptimizer(graph, initial_estimate).optimize() with default parameters over
//   PriorFactor<Pose2> (pose 0), BetweenFactor<Pose2> (t, t+1), BearingRangeFactor<Pose2, Point2> (t, landmark).
// The factor definitions, the LM control flow and its constants are the ones documented in oracle/slam_oracle_pgs.cpp.
//
// MI355X design.  One LM trial (= one tryLambda of every active instance) is six launches:
//   linearize   per pose: 3x3 Hessian blocks A_i, C_i (block-tridiagonal H_pp), 3x2 pose-landmark blocks E_k,
//               gradient; per landmark: 2x2 block D_j, gradient (factors of one landmark are chained in a list)
//   chain       poses are eliminated FIRST: block-tridiagonal Cholesky of H_pp + lambda I fused with the forward
//               recurrence  Y_i = L_i^-1 (E_i - G_i Y_{i-1})  — one thread per landmark COLUMN, sequential over the
//               poses; Y (3N x (2M+1), last column = transformed gradient) goes to HBM once
//   syrk        Schur complement  S = D + lambda I - Y^T Y  on the landmarks with v_mfma_f64_16x16x4_f64 (the one
//               GEMM-shaped piece: 2 * 3N * (2M)^2 / 2 FLOP), 64x64 tiles, k range trimmed by first-detection pose
//   chol        dense blocked Cholesky of S (2M x 2M) + forward/backward substitution -> landmark step
//   backsolve   pose step from the chain factor (forward/backward over the block-bidiagonal factor)
//   evaluate    linearised and true cost of the candidate, retraction, GTSAM's accept / lambda / convergence logic
// All per-instance decisions live on the device; the host only polls the number of active instances.
#include "pgs_kernel.h"
#include "lds_attr.h"

#include <mutex>

#include "slam_math.h"
#include "slam_rng.h"
#include "sim_device.h"

namespace slam {
namespace {

typedef double dbl4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------------------
// factors (whitened residuals / Jacobians); same formulas as the oracle, see there for the GTSAM definitions
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void prior_factor(const PgsParams& p, const double* ps, double e[3]) {
    double s, c;
    det_sincos(ps[2], &s, &c);
    const double dx = p.prior[0] - ps[0], dy = p.prior[1] - ps[1];
    e[0] = -(c * dx + s * dy) * p.w_prior[0];
    e[1] = -(-s * dx + c * dy) * p.w_prior[1];
    e[2] = -remainder(p.prior[2] - ps[2], kTwoPi) * p.w_prior[2];
}

template <bool JAC>
__device__ __forceinline__ void between_factor(const PgsParams& p, const double* pa, const double* pb, float fwd, float ang,
                                               double e[3], double J1[9]) {
    double si, ci, sm, cm;
    det_sincos(pa[2], &si, &ci);
    det_sincos((double)ang, &sm, &cm);
    const double dx = pb[0] - pa[0], dy = pb[1] - pa[1];
    const double hx = ci * dx + si * dy, hy = -si * dx + ci * dy, hth = pb[2] - pa[2];
    const double ux = hx - (double)fwd, uy = hy;
    e[0] = (cm * ux + sm * uy) * p.w_btw[0];
    e[1] = (-sm * ux + cm * uy) * p.w_btw[1];
    e[2] = remainder(hth - (double)ang, kTwoPi) * p.w_btw[2];
    if (JAC) {   // -Ad(h^-1)
        double sh, ch;
        det_sincos(hth, &sh, &ch);
        const double xi = -(ch * hx + sh * hy), yi = sh * hx - ch * hy;
        J1[0] = -ch * p.w_btw[0]; J1[1] = -sh * p.w_btw[0]; J1[2] = -yi * p.w_btw[0];
        J1[3] = sh * p.w_btw[1];  J1[4] = -ch * p.w_btw[1]; J1[5] = xi * p.w_btw[1];
        J1[6] = 0.0;              J1[7] = 0.0;              J1[8] = -p.w_btw[2];
    }
}

template <bool JAC>
__device__ __forceinline__ void bearing_range_factor(const PgsParams& p, const double* ps, const double* l, double b, double r,
                                                     double e[2], double Jp[6], double Jl[4]) {
    double s, c, sb, cb;
    det_sincos(ps[2], &s, &c);
    det_sincos(b, &sb, &cb);
    const double dx = l[0] - ps[0], dy = l[1] - ps[1];
    const double qx = c * dx + s * dy, qy = -s * dx + c * dy;
    const double d2 = qx * qx + qy * qy, n = sqrt(d2);
    const double cp = qx / n, sp = qy / n;
    e[0] = det_atan2(cb * sp - sb * cp, cb * cp + sb * sp) * p.w_meas[0];
    e[1] = (n - r) * p.w_meas[1];
    if (JAC) {
        Jp[0] = (qy / d2) * p.w_meas[0]; Jp[1] = (-qx / d2) * p.w_meas[0]; Jp[2] = -p.w_meas[0];
        Jp[3] = (-qx / n) * p.w_meas[1]; Jp[4] = (-qy / n) * p.w_meas[1]; Jp[5] = 0.0;
        Jl[0] = ((-qy / d2) * c + (qx / d2) * (-s)) * p.w_meas[0];
        Jl[1] = ((-qy / d2) * s + (qx / d2) * c) * p.w_meas[0];
        Jl[2] = (dx / n) * p.w_meas[1];
        Jl[3] = (dy / n) * p.w_meas[1];
    }
}

// per-instance views
struct Inst {
    const int32_t* cnt; const int32_t* mlm; const double* mb; const double* mr;
};
__device__ __forceinline__ Inst inst_view(const PgsParams& p, int b) {
    Inst v;
    v.cnt = p.cnt + (size_t)b * p.N_max;
    v.mlm = p.mlm + (size_t)b * p.N_max * p.KP;
    v.mb = p.mb + (size_t)b * p.N_max * p.KP;
    v.mr = p.mr + (size_t)b * p.N_max * p.KP;
    return v;
}

// deterministic block sum (fixed tree), result valid in every thread
template <int TPB>
__device__ __forceinline__ double block_sum(double v, double* s_buf) {
    const int tid = threadIdx.x;
    __syncthreads();
    s_buf[tid] = v;
    __syncthreads();
#pragma unroll
    for (int off = TPB / 2; off > 0; off >>= 1) {
        if (tid < off) s_buf[tid] = s_buf[tid] + s_buf[tid + off];
        __syncthreads();
    }
    return s_buf[0];
}

// 0.5 * sum |whitened e|^2 of the factors owned by pose i: prior (i = 0), between (i, i+1), bearing-range at i
__device__ __forceinline__ double pose_cost(const PgsParams& p, const Inst& g, const double* pose, const double* lm, int i, int N) {
    double acc = 0.0, e[3];
    if (i == 0) {
        prior_factor(p, pose, e);
        acc = acc + 0.5 * ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
    }
    if (i + 1 < N) {
        between_factor<false>(p, pose + 3 * i, pose + 3 * (i + 1), p.cmds[2 * i], p.cmds[2 * i + 1], e, nullptr);
        acc = acc + 0.5 * ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
    }
    const int kc = g.cnt[i];
    for (int s = 0; s < kc; ++s) {
        const size_t k = (size_t)i * p.KP + s;
        const int j = g.mlm[k] & (kPgsFirstBit - 1);
        bearing_range_factor<false>(p, pose + 3 * i, lm + 2 * j, g.mb[k], g.mr[k], e, nullptr, nullptr);
        acc = acc + 0.5 * (e[0] * e[0] + e[1] * e[1]);
    }
    return acc;
}

template <int TPB>
__device__ __forceinline__ double block_cost(const PgsParams& p, int b, int N, const double* pose, const double* lm, double* s_buf) {
    const Inst g = inst_view(p, b);
    double acc = 0.0;
    for (int i = threadIdx.x; i < N; i += TPB) acc = acc + pose_cost(p, g, pose, lm, i, N);
    return block_sum<TPB>(acc, s_buf);
}

// ------------------------------------------------------------------------------------------------------------
// graph building
// ------------------------------------------------------------------------------------------------------------
// PoseGraph::init (pose_graph.cpp:68-95)
__global__ void pgs_init_kernel(const PgsParams p, double x0, double y0, double yaw0) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    double* ps = p.pose0 + (size_t)b * p.N_max * 3;
    ps[0] = x0; ps[1] = y0; ps[2] = yaw0;
    p.cur[3 * b] = x0; p.cur[3 * b + 1] = y0; p.cur[3 * b + 2] = yaw0;
    p.truth[3 * b] = x0; p.truth[3 * b + 1] = y0; p.truth[3 * b + 2] = yaw0;
    p.M[b] = 0; p.flags[b] = 0;
    p.cnt[(size_t)b * p.N_max] = 0;
    p.state[b] = 1; p.iters[b] = 0; p.trials[b] = 0;
    p.error[b] = 0.0; p.err_init[b] = 0.0; p.lambda[b] = 0.0;
}

// The graph-building half of PoseGraph::update for ONE instance (pose_graph.cpp:216-256): pose node t1 from the
// secondary filter's estimate, then one BearingRangeFactor per detection (getLandmarkIndexFromID :122-147,
// onLandmarkMeasurement :150-178).  Sequential by design (ids are matched in message order).  The factors of one
// landmark are chained oldest -> newest (lm_head / mnext) so the landmark's Hessian block can be summed without atomics.
__device__ void append_step(const PgsParams& p, int b, int t1, const float* meas, int k) {
    const double cx = p.cur[3 * b], cy = p.cur[3 * b + 1], cth = p.cur[3 * b + 2];
    double* ps = p.pose0 + (size_t)b * p.N_max * 3 + 3 * t1;
    ps[0] = cx; ps[1] = cy; ps[2] = cth;                       // initial_estimate.insert(key(t), cur) :248
    int32_t* ids = p.ids + (size_t)b * p.L_max;
    int32_t* mlm = p.mlm + (size_t)b * p.N_max * p.KP;
    int32_t* mnext = p.mnext + (size_t)b * p.N_max * p.KP;
    double* mb = p.mb + (size_t)b * p.N_max * p.KP;
    double* mr = p.mr + (size_t)b * p.N_max * p.KP;
    int32_t* lm_head = p.lm_head + (size_t)b * p.L_max;
    int32_t* lm_last = p.lm_last + (size_t)b * p.L_max;
    int32_t* lm_first = p.lm_first + (size_t)b * p.L_max;
    double* lm0 = p.lm0 + (size_t)b * p.L_max * 2;
    int M = p.M[b], flags = p.flags[b], used = 0;
    for (int l = 0; l < k; ++l) {
        const int id = (int)meas[3 * l];
        const float r = meas[3 * l + 1], bb = meas[3 * l + 2];
        int idx = -1;
        for (int j = 0; j < M; ++j)
            if (ids[j] == id) { idx = j; break; }
        const bool first = idx < 0;
        if (first) {
            if (M >= p.L_max) { flags |= PGS_FLAG_LM_CAP; continue; }
            idx = M; ids[M] = id; M += 1;
            double s, c;                                       // :162  x_t(0) + range*cos(x_t(2)+bearing)
            det_sincos(cth + (double)bb, &s, &c);
            lm0[2 * idx] = cx + (double)r * c;
            lm0[2 * idx + 1] = cy + (double)r * s;
            lm_head[idx] = -1; lm_last[idx] = -1;
            lm_first[idx] = t1;
        }
        if (used >= p.KP) { flags |= PGS_FLAG_MEAS_CAP; continue; }
        const int slot = t1 * p.KP + used;                     // BearingRangeFactor(key(t), lmkey, Rot2(b), r) :174
        mlm[slot] = idx | (first ? kPgsFirstBit : 0);
        mb[slot] = (double)bb; mr[slot] = (double)r;
        mnext[slot] = -1;
        if (lm_last[idx] >= 0) mnext[lm_last[idx]] = slot; else lm_head[idx] = slot;
        lm_last[idx] = slot;
        used += 1;
    }
    p.cnt[(size_t)b * p.N_max + t1] = used;
    p.M[b] = M; p.flags[b] = flags;
}

// updateNaiveVehPoseEstimate + update for host/device supplied measurements: one thread per instance
__global__ void pgs_append_kernel(const PgsParams p, const float* meas, const int32_t* count, int k_stride, const double* sec) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    if (p.N >= p.N_max) { p.flags[b] |= PGS_FLAG_POSE_CAP; return; }
    if (sec) { p.cur[3 * b] = sec[3 * b]; p.cur[3 * b + 1] = sec[3 * b + 1]; p.cur[3 * b + 2] = sec[3 * b + 2]; }
    int k = count ? count[b] : 0;
    k = k < k_stride ? k : k_stride;
    k = k < 0 ? 0 : k;
    append_step(p, b, p.N, meas + (size_t)b * k_stride * 3, k);
}

// T x { get_cmd (sim_node.py:209-250), NaiveFilter::update (filter.h:342-348), updateNaiveVehPoseEstimate, update }
// for one instance per wavefront.  The secondary filter's state IS `cur` (the naive filter keeps nothing else).
__global__ __launch_bounds__(64) void pgs_run_sim_kernel(const PgsParams p, int T, uint32_t step0) {
    constexpr int KCAP = 64;
    __shared__ float s_meas[3 * KCAP];
    const int b = blockIdx.x, lane = threadIdx.x;
    double tx = p.truth[3 * b], ty = p.truth[3 * b + 1], tth = p.truth[3 * b + 2];
    double lmx = 0.0, lmy = 0.0;
    if (lane < p.L) { lmx = p.map[2 * lane]; lmy = p.map[2 * lane + 1]; }
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
        const int i = p.N - 1 + t, t1 = i + 1;
        if (t1 >= p.N_max) { if (lane == 0) p.flags[b] |= PGS_FLAG_POSE_CAP; break; }
        const float fwd = p.cmds[2 * i], ang = p.cmds[2 * i + 1];
        int k = sim_wave<KCAP>(p, b, lane, fwd, ang, step0 + (uint32_t)t, tx, ty, tth, lmx, lmy, s_meas);
        if (k > KCAP) { k = KCAP; if (lane == 0) p.flags[b] |= PGS_FLAG_MEAS_CAP; }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (lane == 0) {
            double s, c;
            const double th = p.cur[3 * b + 2];
            det_sincos(th, &s, &c);
            p.cur[3 * b] = p.cur[3 * b] + (double)fwd * c;
            p.cur[3 * b + 1] = p.cur[3 * b + 1] + (double)fwd * s;
            p.cur[3 * b + 2] = remainder(th + (double)ang, kTwoPi);
            double* th_hist = p.truth_hist + ((size_t)b * p.N_max + (t1 - 1)) * 2;
            th_hist[0] = tx; th_hist[1] = ty;
            append_step(p, b, t1, s_meas, k);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
}

// ------------------------------------------------------------------------------------------------------------
// LM: begin / linearize / chain / syrk / chol / backsolve / evaluate / end
// ------------------------------------------------------------------------------------------------------------
constexpr int TPB = 512;   // threads of the per-pose kernels (two poses per thread at 1000 poses)

// logical block `bl` of a trial-kernel launch -> slot: lane = bl / b_cnt, instance = b_off + bl % b_cnt (PgsParams::lanes)
__device__ __forceinline__ int pgs_slot(const PgsParams& p, int bl) {
    if (p.use_list) {
        if (p.n_list_dev && bl >= *p.n_list_dev) return p.dead_slot;   // enqueued-ahead launch: the list is shorter than the grid
        return p.alist[bl];
    }
    const int lane = bl / p.b_cnt;
    return lane * p.B + p.b_off + (bl - lane * p.b_cnt);
}
// poses of the graph behind slot b: the handle's (lockstep) or the graph's own (asynchronous ticks: lanes are off, slot == instance)
__device__ __forceinline__ int pgs_N(const PgsParams& p, int b) { return p.Nv ? p.Nv[b] : p.N; }
// slots a trial-kernel launch covers
__host__ __device__ __forceinline__ int pgs_nslot(const PgsParams& p) { return p.use_list ? p.n_list : p.b_cnt * (p.lanes > 0 ? p.lanes : 1); }

__global__ __launch_bounds__(TPB) void pgs_lm_begin_kernel(const PgsParams p) {
    __shared__ double s_buf[TPB];
    const int b = blockIdx.x + p.b_off, tid = threadIdx.x;
    if (p.async_ticks && p.state[b] != 6) return;   // asynchronous ticks: only the graphs whose next tick was just appended
    const int N = pgs_N(p, b), M = p.M[b];
    double* pw = p.pw + (size_t)b * p.N_max * 3;
    double* lw = p.lw + (size_t)b * p.L_max * 2;
    const double* p0 = p.pose0 + (size_t)b * p.N_max * 3;
    const double* l0 = p.lm0 + (size_t)b * p.L_max * 2;
    for (int i = tid; i < 3 * N; i += TPB) pw[i] = p0[i];
    for (int i = tid; i < 2 * M; i += TPB) lw[i] = l0[i];
    {   // factors regrouped by landmark in chronological order: event e of landmark j sits at evt_start[j] + e
        __shared__ int s_cnt[TPB];   // L_max <= 255 < TPB
        const int32_t* head = p.lm_head + (size_t)b * p.L_max;
        const int32_t* mnext = p.mnext + (size_t)b * p.N_max * p.KP;
        int32_t* evt_start = p.evt_start + (size_t)b * (p.L_max + 1);
        int32_t* evt_pose = p.evt_pose + (size_t)b * p.N_max * p.KP;
        int32_t* slot_pos = p.slot_pos + (size_t)b * p.N_max * p.KP;
        int c = 0;
        if (tid < M)
            for (int k = head[tid]; k >= 0; k = mnext[k]) ++c;
        s_cnt[tid] = c;
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int j = 0; j < M; ++j) { const int v = s_cnt[j]; s_cnt[j] = run; run += v; }
            s_cnt[M < 255 ? M : 255] = run;
            evt_start[M] = run;
        }
        __syncthreads();
        if (tid < M) {
            int pos = s_cnt[tid];
            evt_start[tid] = pos;
            int32_t* evt_slot = p.evt_slot + (size_t)b * p.N_max * p.KP;
            for (int k = head[tid]; k >= 0; k = mnext[k]) { evt_pose[pos] = k / p.KP; evt_slot[pos] = k; slot_pos[k] = pos; ++pos; }
        }
    }
    __syncthreads();
    if (p.seg_on) {   // segmented elimination: where each column of a segment starts in its landmark's event list
        const int SL = p.seg_len, NS = seg_ns(N, SL), nseg = NS + 1;
        const int32_t* evt_start = p.evt_start + (size_t)b * (p.L_max + 1);
        const int32_t* evt_pose = p.evt_pose + (size_t)b * p.N_max * p.KP;
        const int32_t* ncol = p.seg_ncol + (size_t)b * p.nseg_max;
        const int32_t* slm = p.seg_lm + (size_t)b * p.nseg_max * p.L_max;
        int32_t* sevt = p.seg_evt + (size_t)b * p.nseg_max * p.L_max;
        for (int idx = tid; idx < nseg * p.L_max; idx += TPB) {
            const int ps = idx / p.L_max, lc = idx - ps * p.L_max;
            if (lc >= ncol[ps]) continue;
            const int j = slm[idx], lo = seg_lo(ps, SL);
            int e0 = evt_start[j], e1 = evt_start[j + 1];   // first event with pose >= lo (the list is chronological)
            while (e0 < e1) {
                const int mid = (e0 + e1) >> 1;
                if (evt_pose[mid] < lo) e0 = mid + 1; else e1 = mid;
            }
            sevt[idx] = e0;
        }
        int32_t* spe = p.sep_evt + (size_t)b * p.nseg_max * p.L_max;   // the landmark's event AT a separator's pose
        for (int idx = tid; idx < NS * p.L_max; idx += TPB) {
            const int k = idx / p.L_max, j = idx - k * p.L_max, sp = (k + 1) * SL;
            int found = -1;
            if (j < M) {
                int e0 = evt_start[j], e1 = evt_start[j + 1];
                const int eend = e1;
                while (e0 < e1) {
                    const int mid = (e0 + e1) >> 1;
                    if (evt_pose[mid] < sp) e0 = mid + 1; else e1 = mid;
                }
                if (e0 < eend && evt_pose[e0] == sp) found = e0;
            }
            spe[idx] = found;
        }
    }
    {   // algorithmic FLOP of one Schur-complement SYRK of this instance: 2 per stored lower-triangle element of S_ext and per row of Y
        // that can be non-zero in it.  Sequential elimination: a column is dense from its landmark's first detection on (the
        // right-hand-side row is dense in k).  Segmented: the 3 NS separator rows from the landmark's first separator on, plus per
        // segment the Gram matrix of its own columns.
        double f = 0.0, extra = 0.0;
        const int m2 = 2 * M;
        if (p.seg_on) {
            const int SL = p.seg_len, NS = seg_ns(N, SL), nseg = NS + 1;
            const int32_t* first = p.sep_first + (size_t)b * p.L_max;
            const int32_t* ncol = p.seg_ncol + (size_t)b * p.nseg_max;
            for (int r = tid; r < m2; r += TPB) f += 2.0 * (r + 1) * (double)(3 * NS - 3 * first[r >> 1]);
            for (int ps = tid; ps < nseg; ps += TPB) {
                const double nc = 2.0 * ncol[ps] + 1.0;
                f += 3.0 * (seg_hi(ps, SL, NS, N) - seg_lo(ps, SL)) * nc * (nc + 1.0);
            }
            extra = 2.0 * m2 * (double)(3 * NS);
        } else {
            const int32_t* first = p.lm_first + (size_t)b * p.L_max;
            const int K3 = 3 * N;
            for (int r = tid; r < m2; r += TPB) f += 2.0 * (r + 1) * (double)(K3 - 3 * first[r >> 1]);
            extra = 2.0 * m2 * (double)K3;
        }
        f = block_sum<TPB>(f, s_buf);
        if (tid == 0) p.inst_flop[b] = f + extra;
    }
    const double err = block_cost<TPB>(p, b, N, pw, lw, s_buf);
    if (tid == 0) {
        p.error[b] = err; p.err_init[b] = err; p.cur_error[b] = err;
        p.lambda[b] = 1e-5;                    // LevenbergMarquardtParams::lambdaInitial
        p.iters[b] = 0; p.trials[b] = 0; p.solve_ok[b] = 1; p.nl[b] = 1;
        // first trial: every instance of the group, one lane - or, streaming, the first slots_cap of them; the others wait
        const bool runs = p.slots_cap <= 0 || (int)blockIdx.x < p.slots_cap;
        if (p.async_ticks) p.state[b] = 4;         // the next decide kernel lists it
        else {
            p.state[b] = runs ? 0 : 2;
            if (runs) p.alist[blockIdx.x] = b;
        }
        for (int j = 0; j < p.lanes_max; ++j) p.lin_ok[(size_t)j * p.B + b] = 0;   // (the clones copy nothing of this: plain per-slot state)
        p.flags[b] &= ~(PGS_FLAG_NOT_CONVERGED | PGS_FLAG_NONFINITE);
    }
}

// A += J^T J for a rows x 3 J (same order of operations as the oracle's add_JtJ)
template <int ROWS>
__device__ __forceinline__ void add_JtJ(double A[9], const double* J) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) v += J[3 * r + a] * J[3 * r + c];
            A[3 * a + c] += v;
        }
}

// Linearisation, part 1: one thread per FACTOR (event e of the (landmark, time) list: pose evt_pose[e], slot evt_slot[e]).  The
// bearing-range factors are where the time goes (two sincos, an atan2, a square root and eight divisions each, seven of them per pose
// at BASELINE configs[4]); a thread per pose walked its factors one after the other, each behind two dependent loads.  Every factor
// leaves its blocks E (slot order and event order), the landmark terms Wl, and its SHARE of the pose block in PF; part 2 adds the shares
// in slot order, so every sum has the terms and the order it always had (bit-identical to the one-kernel version).
constexpr int LF_TPB = 256;
__global__ __launch_bounds__(LF_TPB) void pgs_lin_factor_kernel(const PgsParams p) {
    const int nfb = (p.nfact_max + LF_TPB - 1) / LF_TPB;
    const int bl = blockIdx.x / nfb, fb = blockIdx.x - bl * nfb;
    const int b = pgs_slot(p, bl);
    if (p.state[b] || p.lin_ok[b]) return;
    const int e = fb * LF_TPB + threadIdx.x;
    const int M = p.M[b], KP = p.KP;
    if (e >= p.evt_start[(size_t)b * (p.L_max + 1) + M]) return;
    const Inst g = inst_view(p, b);
    const int i = p.evt_pose[(size_t)b * p.N_max * KP + e];
    const size_t k = (size_t)p.evt_slot[(size_t)b * p.N_max * KP + e];
    const double* pose = p.pw + (size_t)b * p.N_max * 3;
    const double* lm = p.lw + (size_t)b * p.L_max * 2;
    const int j = g.mlm[k] & (kPgsFirstBit - 1);
    double e2[2], Jp[6], Jl[4];
    bearing_range_factor<true>(p, pose + 3 * i, lm + 2 * j, g.mb[k], g.mr[k], e2, Jp, Jl);
    double* PF = p.PF + ((size_t)b * p.N_max * KP + k) * 12;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) {   // add_JtJ<2>'s term
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < 2; ++r) v += Jp[3 * r + a] * Jp[3 * r + c];
            PF[3 * a + c] = v;
        }
#pragma unroll
    for (int a = 0; a < 3; ++a) PF[9 + a] = -(Jp[a] * e2[0] + Jp[3 + a] * e2[1]);
    double* E = p.E + ((size_t)b * p.N_max * KP + k) * 6;
    double* El = p.Elm + ((size_t)b * p.N_max * KP + e) * 6;   // the same block in (landmark, time) order for the chain / segment kernels
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) { const double v = Jp[a] * Jl[c] + Jp[3 + a] * Jl[2 + c]; E[2 * a + c] = v; El[2 * a + c] = v; }
    double* W = p.Wl + ((size_t)b * p.N_max * KP + e) * 5;   // in (landmark, time) order: the landmark sum of part 2 reads contiguously
    W[0] = Jl[0] * Jl[0] + Jl[2] * Jl[2];
    W[1] = Jl[0] * Jl[1] + Jl[2] * Jl[3];
    W[2] = Jl[1] * Jl[1] + Jl[3] * Jl[3];
    W[3] = -(Jl[0] * e2[0] + Jl[2] * e2[1]);
    W[4] = -(Jl[1] * e2[0] + Jl[3] * e2[1]);
}

// Linearisation, part 2: per pose the prior / between factors and the sum of its factors' shares (slot order); per landmark the sum of
// its factors' terms (chronological order).
__global__ __launch_bounds__(TPB) void pgs_linearize_kernel(const PgsParams p) {
    const int b = pgs_slot(p, blockIdx.x), tid = threadIdx.x;
    if (p.state[b]) return;
    const int N = pgs_N(p, b), KP = p.KP, M = p.M[b];
    if (p.seg_on && tid == 0) p.solve_ok[b] = 1;   // segmented elimination: a failing segment / separator clears it (the sequential chain kernel sets it itself)
    if (p.lin_ok[b]) return;                       // the previous trial of this slot failed: same values, same linearisation
    const Inst g = inst_view(p, b);
    const double* pose = p.pw + (size_t)b * p.N_max * 3;
    double* Ab = p.A + (size_t)b * p.N_max * 9;
    double* Cb = p.C + (size_t)b * p.N_max * 9;
    double* gpb = p.gp + (size_t)b * p.N_max * 3;
    const double* PFb = p.PF + (size_t)b * p.N_max * KP * 12;
    double* Wlb = p.Wl + (size_t)b * p.N_max * KP * 5;
    for (int i = tid; i < N; i += TPB) {
        double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, gg[3] = {0, 0, 0}, e[3], J1[9];
        if (i == 0) {
            prior_factor(p, pose, e);
#pragma unroll
            for (int k = 0; k < 3; ++k) { A[4 * k] += p.w_prior[k] * p.w_prior[k]; gg[k] += -e[k] * p.w_prior[k]; }
        }
        if (i > 0) {   // between (i-1, i): J2 = diag(w); H[i][i-1] = J2^T J1
            between_factor<true>(p, pose + 3 * (i - 1), pose + 3 * i, p.cmds[2 * (i - 1)], p.cmds[2 * (i - 1) + 1], e, J1);
#pragma unroll
            for (int k = 0; k < 3; ++k) { A[4 * k] += p.w_btw[k] * p.w_btw[k]; gg[k] += -e[k] * p.w_btw[k]; }
            double* C = Cb + 9 * (i - 1);
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) C[3 * a + c] = p.w_btw[a] * J1[3 * a + c];
        }
        if (i + 1 < N) {   // between (i, i+1): J1
            between_factor<true>(p, pose + 3 * i, pose + 3 * (i + 1), p.cmds[2 * i], p.cmds[2 * i + 1], e, J1);
            add_JtJ<3>(A, J1);
#pragma unroll
            for (int a = 0; a < 3; ++a) gg[a] += -(J1[a] * e[0] + J1[3 + a] * e[1] + J1[6 + a] * e[2]);
        }
        const int kc = g.cnt[i];
        const double* PF = PFb + (size_t)i * KP * 12;
        constexpr int UB = 4;   // the shares are fetched four factors at a time, the additions stay in slot order
        int s = 0;
#pragma unroll 1
        for (; s + UB <= kc; s += UB) {
            double w[UB][12];
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int c = 0; c < 12; ++c) w[u][c] = PF[12 * (size_t)(s + u) + c];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
#pragma unroll
                for (int c = 0; c < 9; ++c) A[c] += w[u][c];
#pragma unroll
                for (int c = 0; c < 3; ++c) gg[c] += w[u][9 + c];
            }
        }
        for (; s < kc; ++s) {
#pragma unroll
            for (int c = 0; c < 9; ++c) A[c] += PF[12 * (size_t)s + c];
#pragma unroll
            for (int c = 0; c < 3; ++c) gg[c] += PF[12 * (size_t)s + 9 + c];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) Ab[9 * i + k] = A[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) gpb[3 * i + k] = gg[k];
    }
    const int32_t* evt_start = p.evt_start + (size_t)b * (p.L_max + 1);
    double* Db = p.D + (size_t)b * p.L_max * 3;
    double* glb = p.gl + (size_t)b * p.L_max * 2;
    for (int j = tid; j < M; j += TPB) {   // landmark j: chronological sum over its factors (contiguous event records)
        double d0 = 0, d1 = 0, d2 = 0, g0 = 0, g1 = 0;
        const int e1 = evt_start[j + 1];
        int e = evt_start[j];
        // The additions stay in chronological order (the oracle's order), the LOADS do not have to wait for them: the
        // records of a landmark are contiguous, so eight events are fetched at once.
        constexpr int UB = 8;
#pragma unroll 1
        for (; e + UB <= e1; e += UB) {
            double w[UB][5];
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int c = 0; c < 5; ++c) w[u][c] = Wlb[5 * (size_t)(e + u) + c];
#pragma unroll
            for (int u = 0; u < UB; ++u) { d0 += w[u][0]; d1 += w[u][1]; d2 += w[u][2]; g0 += w[u][3]; g1 += w[u][4]; }
        }
        for (; e < e1; ++e) {
            const double* W = Wlb + 5 * (size_t)e;
            d0 += W[0]; d1 += W[1]; d2 += W[2]; g0 += W[3]; g1 += W[4];
        }
        Db[3 * j] = d0; Db[3 * j + 1] = d1; Db[3 * j + 2] = d2;
        glb[2 * j] = g0; glb[2 * j + 1] = g1;
    }
    if (tid == 0) p.lin_ok[b] = 1;
}

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

// S_ext = [D + lambda I, .; gl^T, .] - Y^T Y on 128x128 tiles of the lower triangle; 4 wavefronts x (64x64) each = 4x4
// accumulators of v_mfma_f64_16x16x4_f64 per wavefront (8 operand loads feed 16 MFMAs: the kernel is bound by the
// L2 -> L1 operand stream, not by HBM, so the wave tile is as large as the register file allows).  Row 2M of S_ext is
// the right-hand side gl - Y^T z.
// WT = wavefront tile (64: bulk trials, most instances active; 32: straggler trials, where the few active instances need
// more wavefronts each).  Workgroup tile SY_T = 2 * WT.
template <int WT>
__global__ __launch_bounds__(256, 2) void pgs_syrk_kernel(const PgsParams p) {
    constexpr int SY_T = 2 * WT, NI = WT / 16;
    // XCD-aware placement: workgroup id w runs on XCD (w mod 8).  All tiles of one instance read the same Y, k chunk
    // by k chunk and roughly in step, so they are given ids that share one XCD (one L2): id = 8 * q + xcd with
    // q = (instance / 8) * ntiles + tile, instance = 8 * (q / ntiles) + xcd.
    const int ntr = (p.LD + SY_T - 1) / SY_T;
    const int ntl = ntr * (ntr + 1) / 2;
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int bl = (q / ntl) * 8 + xcd;     // instance within the launched group
    if (bl >= pgs_nslot(p)) return;
    const int b = pgs_slot(p, bl);
    if (p.state[b] || !p.solve_ok[b]) return;
    const int LD = p.LD, m2 = 2 * p.M[b];
    // decode the lower-triangular tile index
    int ti = 0, t = q % ntl;
    while (t >= ti + 1) { t -= ti + 1; ti += 1; }
    const int tj = t;
    if (ti * SY_T > m2) return;                     // tile row holds nothing (rows > 2M)
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wr = w >> 1, wc = w & 1;
    if (ti == tj && wr == 0 && wc == 1) return;     // strictly upper part of a diagonal tile
    const int rowbase = ti * SY_T + wr * WT, colbase = tj * SY_T + wc * WT;
    if (rowbase > m2 || colbase > m2) return;
    // (segmented elimination: the block of Y rows is the separators' - syrk_row0 / syrk_rows / syrk_first, pgs_kernel.h)
    const int K3 = p.syrk_rows >= 0 ? (p.Nv ? 3 * seg_ns(pgs_N(p, b), p.seg_len) : p.syrk_rows) : 3 * pgs_N(p, b);
    int k0 = 0;
    // Y[k][c] == 0 before the first detection of column c's landmark, and landmarks are numbered in order of first
    // detection: this wavefront's 64 rows are all zero before pose lm_first[rowbase / 2] (unless it holds the z row)
    const int32_t* firstrow = p.syrk_first ? p.syrk_first : p.lm_first;
    if (rowbase + WT - 1 < m2 && !p.syrk_notrim) k0 = (3 * firstrow[(size_t)b * p.L_max + (rowbase >> 1)]) & ~3;
    if (k0 > K3) k0 = K3 & ~3;
    const double* Yb = p.Y + (size_t)b * p.y_stride + (size_t)p.syrk_row0 * p.LD;
    dbl4_t acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (dbl4_t){0.0, 0.0, 0.0, 0.0};
    const int kq = lane >> 4, cl = lane & 15;
    // per-lane operand columns; columns >= LD do not exist (their products land in rows / cols that are never stored)
    int ca[NI], cb[NI];
#pragma unroll
    for (int h = 0; h < NI; ++h) {
        ca[h] = rowbase + 16 * h + cl; if (ca[h] >= LD) ca[h] = LD - 1;
        cb[h] = colbase + 16 * h + cl; if (cb[h] >= LD) cb[h] = LD - 1;
    }
    constexpr int KU = WT == 64 ? 2 : 4;   // k-steps (of 4 rows) in flight
    const int Kfull = k0 + ((K3 - k0) / (4 * KU)) * (4 * KU);
    const double* row = Yb + (size_t)(k0 + kq) * LD;
#pragma unroll 1
    for (int k = k0; k < Kfull; k += 4 * KU) {
        double a[KU][NI], bb[KU][NI];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
#pragma unroll
            for (int h = 0; h < NI; ++h) { a[u][h] = row[ca[h]]; bb[u][h] = row[cb[h]]; }
            row += (size_t)4 * LD;
        }
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][i], bb[u][j], acc[i][j], 0, 0, 0);
    }
    for (int k = Kfull; k < K3; k += 4) {   // remainder, row-guarded
        const int kk = k + kq;
        const bool in = kk < K3;
        double a[NI], bb[NI];
#pragma unroll
        for (int h = 0; h < NI; ++h) { a[h] = in ? row[ca[h]] : 0.0; bb[h] = in ? row[cb[h]] : 0.0; }
        row += (size_t)4 * LD;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
    const double lambda = p.lambda[b];
    const double* Db = p.D + (size_t)b * p.L_max * 3;
    const double* glb = p.gl + (size_t)b * p.L_max * 2;
    double* Sb = p.S + (size_t)b * LD * LD;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = rowbase + 16 * i + kq + 4 * r4;   // C/D layout of the f64 MFMA: row = (lane>>4) + 4*reg
                const int c = colbase + 16 * j + cl;
                double v = -acc[i][j][r4];
                if (r < m2) {
                    if (c == r) v += Db[3 * (r >> 1) + ((r & 1) ? 2 : 0)] + lambda;
                    else if ((c >> 1) == (r >> 1) && c < r) v += Db[3 * (r >> 1) + 1];
                } else if (r == m2 && c < m2) {
                    v += glb[c];
                }
                acc[i][j][r4] = v;
            }
    if constexpr (WT == 32) {
        if (p.seg_on) {
            // Segmented elimination: this launch covered the separators' rows of Y; the interior rows' products arrive as the segments'
            // Gram matrices T_p (pgs_seg_gram_kernel) and are subtracted here, segment after segment - a fixed order per element.  A
            // segment touches this wavefront's 32 x 32 tile only if it sees a landmark of the tile's row block AND one of its column block
            // (seg_blk: the local ranges of the 16-landmark blocks): a handful of the segments for a tile near the diagonal, none far from
            // it; the right-hand-side row (the gradient column of every segment) meets them all.
            const int nb1 = seg_nb1(p.L_max), nseg = seg_ns(pgs_N(p, b), p.seg_len) + 1;
            const int32_t* blk = p.seg_blk + (size_t)b * p.nseg_max * nb1;
            const int32_t* sinv = p.seg_inv + (size_t)b * p.nseg_max * p.L_max;
            const int32_t* ncolb = p.seg_ncol + (size_t)b * p.nseg_max;
            const double* Tb = p.segT + (size_t)b * p.nseg_max * (128 * 128);
            const int rb = rowbase >> 5, cb = colbase >> 5;
            const bool has_rhs = m2 >= rowbase && m2 < rowbase + WT;
            if (has_rhs) {   // wave-uniform
                // The right-hand-side row meets EVERY segment (its gradient column); one segment at a time that was 32 dependent
                // round trips for the tiles of the last row block.  Lane l takes column colbase + l of the row: the index loads of eight
                // segments go out together, then the eight T entries, then the subtractions in segment order.
                __shared__ double s_rhs[4][WT];
                double* rh = s_rhs[w];
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4)
                            if (rowbase + 16 * i + kq + 4 * r4 == m2) rh[16 * j + cl] = acc[i][j][r4];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                const int c = colbase + lane;
                if (lane < WT && c < m2) {
                    const int jl = c >> 1, d = c & 1;
                    double v = rh[lane];
                    constexpr int SB = 8;
#pragma unroll 1
                    for (int ps0 = 0; ps0 < nseg; ps0 += SB) {
                        int q[SB], nl[SB];
#pragma unroll
                        for (int u = 0; u < SB; ++u) {
                            const int ps = ps0 + u < nseg ? ps0 + u : nseg - 1;
                            q[u] = ps0 + u < nseg ? sinv[(size_t)ps * p.L_max + jl] : -1;
                            nl[u] = ncolb[ps];
                        }
                        double t[SB];
#pragma unroll
                        for (int u = 0; u < SB; ++u) {
                            const int ps = ps0 + u < nseg ? ps0 + u : nseg - 1;
                            t[u] = q[u] >= 0 ? Tb[(size_t)ps * (128 * 128) + (size_t)(2 * nl[u]) * 128 + 2 * q[u] + d] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < SB; ++u)
                            if (q[u] >= 0) v = v - t[u];
                    }
                    rh[lane] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4)
                            if (rowbase + 16 * i + kq + 4 * r4 == m2 && colbase + 16 * j + cl < m2) acc[i][j][r4] = rh[16 * j + cl];
            }
            // Which segments touch the tile: one LANE per segment tests its seg_blk row, a ballot gives the list - one round trip for all of
            // them (segment after segment with scalar loads it was one per segment, ~30 of them for the handful that are relevant).  The
            // relevant ones are then subtracted in ascending order, the index loads of the next one in flight beside the T entries of the
            // current one: about one dependent round trip per relevant segment instead of two.
            auto load_idx = [&](const int ps, int (&lr)[NI][4], int (&lc)[NI]) {
                const int32_t* iv = sinv + (size_t)ps * p.L_max;
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int r = rowbase + 16 * i + kq + 4 * r4;
                        int l = -1;
                        if (r < m2) { const int q = iv[r >> 1]; l = q >= 0 ? 2 * q + (r & 1) : -1; }
                        lr[i][r4] = l;
                    }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int c = colbase + 16 * j + cl;
                    int l = -1;
                    if (c < m2) { const int q = iv[c >> 1]; l = q >= 0 ? 2 * q + (c & 1) : -1; }
                    lc[j] = l;
                }
            };
#pragma unroll 1
            for (int ps0 = 0; ps0 < nseg; ps0 += 64) {
                bool rel = false;
                if (ps0 + lane < nseg) {
                    const int32_t* bk = blk + (size_t)(ps0 + lane) * nb1;
                    rel = bk[rb + 1] > bk[rb] && bk[cb + 1] > bk[cb];
                }
                unsigned long long mask = __ballot(rel);   // wave-uniform from here on
                int lr[NI][4], lc[NI];
                int ps = mask ? ps0 + (__ffsll((long long)mask) - 1) : -1;
                if (ps >= 0) load_idx(ps, lr, lc);
#pragma unroll 1
                while (ps >= 0) {
                    mask &= mask - 1ull;
                    const int psn = mask ? ps0 + (__ffsll((long long)mask) - 1) : -1;
                    int lrn[NI][4], lcn[NI];
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        lcn[i] = -1;
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) lrn[i][r4] = -1;
                    }
                    if (psn >= 0) load_idx(psn, lrn, lcn);
                    const double* Tp = Tb + (size_t)ps * (128 * 128);
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4)
                                if (lr[i][r4] >= 0 && lc[j] >= 0 && lc[j] <= lr[i][r4]) acc[i][j][r4] = acc[i][j][r4] - Tp[(size_t)lr[i][r4] * 128 + lc[j]];
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        lc[i] = lcn[i];
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) lr[i][r4] = lrn[i][r4];
                    }
                    ps = psn;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = rowbase + 16 * i + kq + 4 * r4;
                const int c = colbase + 16 * j + cl;
                if (r > m2 || c > r) continue;
                Sb[(size_t)r * LD + c] = acc[i][j][r4];
            }
}

// S_ext with INSTANCE-RESIDENT accumulators (the default from a few dozen active instances): SI_NB workgroups of 16
// wavefronts per instance hold the whole lower triangle of S_ext in registers (32x32 tiles dealt round-robin, in order
// of their first non-zero row, to the 16 * SI_NB wavefronts: at most SI_NS tiles = 64 accumulator VGPRs each) and
// stream Y through double-buffered LDS chunks of SI_ROWS rows, every row of Y read ONCE per workgroup with 16-byte
// loads that are issued a chunk ahead.  The tile kernel above re-reads Y per tile and leaves the sharing to L2, which
// it does not get (27 % hit rate, 62 % of the wavefront cycles waiting on misses, profiles/r01m_pgs_cache).
// LDS row stride = columns + 16 doubles: the four k rows of an MFMA operand (lanes 16 apart) then sit 128 bytes apart
// in bank space, so the 8-byte fragment reads are conflict-free.  The workgroups of one instance get ids on the same
// XCD and march through Y in step, so all but the first read L2.
constexpr int SI_ROWS = 16, SI_NB = 3, SI_NS = 2, SI_TPB = 1024;
__global__ __launch_bounds__(SI_TPB) void pgs_syrk_inst_kernel(const PgsParams p) {
    extern __shared__ double s_y[];   // [2][SI_ROWS][ldl]
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int bl = (q / SI_NB) * 8 + xcd, hb = q % SI_NB;
    if (bl >= pgs_nslot(p)) return;
    const int b = pgs_slot(p, bl);
    if (p.state[b] || !p.solve_ok[b]) return;
    const int LD = p.LD, m2 = 2 * p.M[b];
    int ncol = (m2 + 1 + 31) & ~31;               // columns that hold data (incl. the z column), in 32-wide tiles
    if (ncol > LD) ncol = LD;
    const int ldl = ncol + 16;
    const int nt = ncol / 32, ntile = nt * (nt + 1) / 2;
    if (hb >= ntile) return;                      // small graphs: this workgroup holds no tile
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, cl = lane & 15;
    const int gw = w * SI_NB + hb;                // wavefront number within the instance
    const int K3 = 3 * pgs_N(p, b);
    const int nchunk = (K3 + SI_ROWS - 1) / SI_ROWS;
    const double* Yb = p.Y + (size_t)b * p.y_stride;

    // Per 16-row half of a tile the first chunk that can hold a non-zero: rows of Y^T are zero before the first detection of
    // their landmark and landmarks are numbered by first detection, so half h of a tile starts at the chunk of
    // lm_first[(rowbase + 16 h) / 2]; a half without landmark rows (>= 2M) never runs.  The z row (2M: the right-hand side
    // gl - Y^T z, dense in k) is NOT given to the matrix pipe - it would keep the whole last tile row at the full k range,
    // 22 % of the MFMA work of an instance at 1000 x 171 - but accumulated on the VALU by the wavefront that holds the tile:
    // lane -> (column, half of the chunk's rows), eight FMAs per chunk.
    int rowbase[SI_NS], colbase[SI_NS], c0[SI_NS][2];
    bool have[SI_NS], dg[SI_NS];
    int zcol = -1;                                // column base of this wavefront's tile of the last tile row
    dbl4_t acc[SI_NS][2][2];
    const int32_t* lmf = p.lm_first + (size_t)b * p.L_max;
    const bool trim = !(p.syrk_notrim & 1);
#pragma unroll
    for (int s = 0; s < SI_NS; ++s) {
        const int t = gw + 16 * SI_NB * s;
        have[s] = t < ntile;
        int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (ti * (ti + 1) / 2 > t) --ti;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        const int tj = t - ti * (ti + 1) / 2;
        rowbase[s] = have[s] ? 32 * ti : 0; colbase[s] = have[s] ? 32 * tj : 0;
        dg[s] = ti == tj;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rb = rowbase[s] + 16 * h;
            c0[s][h] = 0x7fffffff;
            if (have[s] && rb < m2) c0[s][h] = trim ? (3 * lmf[rb >> 1]) / SI_ROWS : 0;
        }
        if (have[s] && rowbase[s] + 31 >= m2) zcol = colbase[s];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[s][i][j] = (dbl4_t){0.0, 0.0, 0.0, 0.0};
    }
    double zacc = 0.0;
    const int zoff = (lane >> 5) * (SI_ROWS / 2) * ldl;   // this lane's half of a chunk's rows

    // staging: a chunk is SI_ROWS x ncol doubles = SI_ROWS * ncol / 2 16-byte vectors
    const int vpr = ncol >> 1;                    // vectors per row
    const int nvec = SI_ROWS * vpr;
    constexpr int NV = (SI_ROWS * (448 / 2) + SI_TPB - 1) / SI_TPB;   // LD <= 448
    typedef double dbl2v __attribute__((ext_vector_type(2)));
    dbl2v stage[NV];
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int v = tid + SI_TPB * u;
            const int r = v / vpr, cv = v - r * vpr;
            const int k = c * SI_ROWS + r;
            stage[u] = (dbl2v){0.0, 0.0};
            if (v < nvec && k < K3) stage[u] = *reinterpret_cast<const dbl2v*>(Yb + (size_t)k * LD + 2 * cv);
        }
    };
    auto put = [&](int buf) {
        double* dst = s_y + (size_t)buf * SI_ROWS * ldl;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int v = tid + SI_TPB * u;
            const int r = v / vpr, cv = v - r * vpr;
            if (v < nvec) *reinterpret_cast<dbl2v*>(dst + r * ldl + 2 * cv) = stage[u];
        }
    };
    fetch(0);
    put(0);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk && !(p.syrk_notrim & 4)) fetch(c + 1);
        const double* cbuf = s_y + (size_t)(c & 1) * SI_ROWS * ldl;
        const double* src = cbuf + kq * ldl + cl;
#pragma unroll
        for (int s = 0; s < SI_NS; ++s) {
            if (c < c0[s][0] || (p.syrk_notrim & 2)) continue;            // wave-uniform
            const bool both = c >= c0[s][1];                              // rows 16..31 of the tile have begun
            const double* sa = src + rowbase[s];
            const double* sb = src + colbase[s];
#pragma unroll
            for (int ks = 0; ks < SI_ROWS / 4; ++ks) {
                const double a0 = sa[ks * 4 * ldl];
                const double b0 = sb[ks * 4 * ldl], b1 = sb[ks * 4 * ldl + 16];
                acc[s][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[s][0][0], 0, 0, 0);
                if (!dg[s]) acc[s][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[s][0][1], 0, 0, 0);   // strictly upper on a diagonal tile
                if (both) {
                    const double a1 = sa[ks * 4 * ldl + 16];
                    acc[s][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[s][1][0], 0, 0, 0);
                    acc[s][1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[s][1][1], 0, 0, 0);
                }
            }
        }
        if (zcol >= 0) {                                                  // wave-uniform: the z row of this wavefront's tile
            const double* zy = cbuf + zoff;
            const int cc = zcol + (lane & 31);
#pragma unroll
            for (int r = 0; r < SI_ROWS / 2; ++r) zacc = fma(zy[r * ldl + m2], zy[r * ldl + cc], zacc);
        }
        if (c + 1 < nchunk) put((c + 1) & 1);
        __syncthreads();
    }
    const double lambda = p.lambda[b];
    const double* Db = p.D + (size_t)b * p.L_max * 3;
    const double* glb = p.gl + (size_t)b * p.L_max * 2;
    double* Sb = p.S + (size_t)b * LD * LD;
    zacc = zacc + __shfl_xor(zacc, 32);           // both halves of the chunks' rows: lane l (and l + 32) holds column zcol + (l & 31)
#pragma unroll
    for (int s = 0; s < SI_NS; ++s) {
        if (!have[s]) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const double zj = __shfl(zacc, 16 * j + cl);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int r = rowbase[s] + 16 * i + kq + 4 * r4;   // C/D layout of the f64 MFMA: row = (lane>>4) + 4*reg
                    const int cc = colbase[s] + 16 * j + cl;
                    if (r > m2 || cc > r) continue;
                    double v = -acc[s][i][j][r4];
                    if (r < m2) {
                        if (cc == r) v += Db[3 * (r >> 1) + ((r & 1) ? 2 : 0)] + lambda;
                        else if ((cc >> 1) == (r >> 1)) v += Db[3 * (r >> 1) + 1];
                    } else {
                        v = -zj;                                       // row 2M comes from the VALU sum, not from the MFMA
                        if (cc < m2) v += glb[cc];
                    }
                    Sb[(size_t)r * LD + cc] = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// chain + SYRK FUSED (while every instance's lower triangle fits FC_TILES wavefront tiles): Y never
// goes to HBM.  NB = 2, 3 or 4 workgroups of 8 wavefronts per instance (the host's choice per trial: as many as leave every
// workgroup of the launch a CU of its own); all run the whole chain (the sequential 3x3
// recursion is the critical path of a trial and costs one lane), each keeps its share of the instance's 32x32 tiles of
// S = D + lambda I - Y^T Y as MFMA accumulators (NS tiles = 32 NS VGPRs per wavefront at two wavefronts per SIMD).
//   wavefront 0         PRODUCER, as in pgs_chain_kernel but in chunks of FC_P poses and with the next chunk's inputs
//                       fetched under the current chunk's recursion; works one chunk ahead
//   wavefronts 1..7     one column of Y per lane (448 >= 2M + 1): the column recurrence of chunk n into an LDS buffer
//                       of 3 FC_P rows (double-buffered) and the lane's term of the right-hand-side row gl - Y^T z
//   wavefronts 1-3, 5-7 one barrier later: v_mfma_f64_16x16x4_f64 over those rows for the wavefront's tiles (16-row halves
//                       trimmed by first detection)
//   wavefront 4         shares its SIMD with the producer and therefore holds NO tiles: on gfx950 the fp64 MFMA runs at the
//                       vector fp64 rate of its SIMD and a dependent fp64 chain beside it takes 27.5 instead of 11.5 cycles
//                       per link (tools/calib_mfma64; recursion 0.56 -> 0.79 ms).  It stages the bearing-range blocks of the
//                       next chunk instead.
// Time per workgroup ~ max(recursion + its staging, columns + MFMA of the busiest SIMD) per chunk.  Same arithmetic per
// element of Y and per tile as the unfused pair (the k order of the MFMA accumulation is the same; only row 2M is
// summed on the VALU instead of the matrix pipe).
// ------------------------------------------------------------------------------------------------------------
constexpr int FC_P = 4, FC_ROWS = 3 * FC_P, FC_TPB = 512, FC_TILES = 72;   // tiles an instance may have: NB workgroups x 6 wavefronts x NS
constexpr int FC_KP = 32, FC_LMAX = 224;           // factor slots per pose / landmarks the event staging is sized for
constexpr int FC_NF = FC_P * FC_KP / 64, FC_NE = FC_P * FC_KP * 3 / 64;   // per lane of the staging wavefront: factor slots, 16-byte pieces of E
typedef double dbl2_t __attribute__((ext_vector_type(2)));
template <int NS, int NB>
__global__ __launch_bounds__(FC_TPB) void pgs_chain_syrk_kernel(const PgsParams p) {
    constexpr int FC_NB = NB, FC_NW = 6 * NB;
    extern __shared__ double s_yb[];                // [2][FC_ROWS][ldl]
    __shared__ double s_in[2][FC_P][18];            // A (6 unique), C (9), gp (3)
    __shared__ double s_ring[2][FC_P][18];          // Linv (6), G (9), gp (3)
    // The E blocks of a chunk's bearing-range factors, staged by wavefront 4 (pose-major, as linearize
    // wrote them: one contiguous piece per chunk) and an index (pose of the chunk, landmark) -> factor slot, tagged with the
    // pose number so that it never needs clearing.  The column lanes pick their E entries from LDS: a lane that fetched its
    // next event from HBM when the previous one fired made its whole wavefront wait for that load at the next pose.
    __shared__ dbl2_t s_E[2][FC_P * FC_KP * 3 + 3];   // + one all-zero block: what a column without an event adds
    __shared__ int s_idx[2][FC_P][FC_LMAX];
    __shared__ int s_fail;
    const int bl = blockIdx.x / FC_NB, hb = blockIdx.x - bl * FC_NB;
    const int b = pgs_slot(p, bl), tid = threadIdx.x;
    if (p.state[b]) {
        if (p.prof && tid == 0) p.prof[(size_t)p.B * p.lanes_max * 8 + (size_t)b * 16 + 8 * hb + 1] = 0;   // debug: no stamp from this launch
        return;
    }
    const int N = pgs_N(p, b), LD = p.LD, m2 = 2 * p.M[b];
    const int nch = (N + FC_P - 1) / FC_P;
    const int ncol = (m2 + 1 + 31) & ~31, ldl = ncol + 16;
    if (tid == 0) s_fail = 0;
    if (p.prof && (p.syrk_notrim & 16)) {            // debug: which SIMD each wavefront of the workgroup runs on (HW_ID bits 5:4)
        if ((tid & 63) == 0) p.prof[(size_t)p.B * p.lanes_max * 8 + (size_t)b * 16 + 8 * hb + (tid >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        return;
    }
    for (int k = tid; k < 2 * FC_P * FC_LMAX; k += FC_TPB) (&s_idx[0][0][0])[k] = -1;
    if (tid < 6) s_E[tid / 3][FC_P * FC_KP * 3 + tid % 3] = (dbl2_t){0.0, 0.0};
    __syncthreads();
    if (tid < 64) {
        // ------------------------------------------------ producer ------------------------------------------------
        const unsigned long long t_begin = p.prof ? wall_clock64() : 0ull;
        unsigned long long t_rec = 0, t_pre = 0, t_post = 0;   // debug: time inside the recursion proper, before (loads issued) and after it (staging)
        const double lambda = p.lambda[b];
        const double* Ab = p.A + (size_t)b * p.N_max * 9;
        const double* Cb = p.C + (size_t)b * p.N_max * 9;
        const double* gpb = p.gp + (size_t)b * p.N_max * 3;
        double* Lb = p.Linv + (size_t)b * p.N_max * 6;
        double* Gb = p.G + (size_t)b * p.N_max * 9;
        double stg[18];
        auto load_in = [&](int ch) {                // inputs of pose ch * FC_P + tid into registers (lanes < FC_P)
            const int i = ch * FC_P + tid;
#pragma unroll
            for (int k = 0; k < 18; ++k) stg[k] = 0.0;
            if (tid < FC_P && i < N) {
                const double* A = Ab + 9 * i;
                stg[0] = A[0]; stg[1] = A[3]; stg[2] = A[4]; stg[3] = A[6]; stg[4] = A[7]; stg[5] = A[8];
                if (i > 0) {
                    const double* C = Cb + 9 * (i - 1);
#pragma unroll
                    for (int k = 0; k < 9; ++k) stg[6 + k] = C[k];
                }
                stg[15] = gpb[3 * i]; stg[16] = gpb[3 * i + 1]; stg[17] = gpb[3 * i + 2];
            }
        };
        auto store_in = [&](int buf) {
            if (tid < FC_P) {
#pragma unroll
                for (int k = 0; k < 18; ++k) s_in[buf][tid][k] = stg[k];
            }
        };
        load_in(0);
        store_in(0);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        double I0 = 0, I1 = 0, I2 = 0, I3 = 0, I4 = 0, I5 = 0;   // lane 0: Linv of the previous pose
#pragma unroll 1
        for (int it = 0; it <= nch + 1; ++it) {
            if (it < nch) {
                const int base = it * FC_P;
                const int n = (N - base) < FC_P ? (N - base) : FC_P;
                const unsigned long long tpa = p.prof ? wall_clock64() : 0ull;
                if (it + 1 < nch) load_in(it + 1);
                const unsigned long long tp0 = p.prof ? wall_clock64() : 0ull;
                t_pre += tp0 - tpa;
                if (tid == 0) {
                    double (*out)[18] = s_ring[it & 1];
                    const double (*sin)[18] = s_in[it & 1];
                    bool ok = s_fail == 0;
                    double in[18], nx[18];
#pragma unroll
                    for (int k = 0; k < 18; ++k) in[k] = sin[0][k];
#pragma unroll 1
                    for (int l = 0; l < n && ok; ++l) {
                        const int ln = l + 1 < n ? l + 1 : l;
#pragma unroll
                        for (int k = 0; k < 18; ++k) nx[k] = sin[ln][k];
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
                    for (int l = n; l < FC_P; ++l)   // past the last pose: Linv = G = 0, the columns then write zero rows
#pragma unroll
                        for (int k = 0; k < 18; ++k) out[l][k] = 0.0;
                }
                const unsigned long long tp1 = p.prof ? wall_clock64() : 0ull;
                t_rec += tp1 - tp0;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (it + 1 < nch) store_in((it + 1) & 1);
                if (hb == 0 && tid < n && s_fail == 0) {   // factor to HBM for the pose back-substitution
                    const double* o = s_ring[it & 1][tid];
                    double* L = Lb + 6 * (base + tid);
#pragma unroll
                    for (int k = 0; k < 6; ++k) L[k] = o[k];
                    double* Go = Gb + 9 * (base + tid);
#pragma unroll
                    for (int k = 0; k < 9; ++k) Go[k] = o[6 + k];
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                if (p.prof) t_post += wall_clock64() - tp1;
            }
            __syncthreads();
            if (s_fail) break;
        }
        if (tid == 0 && hb == 0) p.solve_ok[b] = s_fail ? 0 : 1;
        if (p.prof && tid == 0) {   // debug: [slots][2][8] after the chol timers: per workgroup begin, end (100 MHz), producer: before / after the recursion, recursion, wavefront 1: columns, tiles, barrier
            unsigned long long* o = p.prof + (size_t)p.B * p.lanes_max * 8 + (size_t)b * 16 + 8 * hb;
            o[0] = t_begin; o[1] = wall_clock64();
            o[2] = t_pre; o[3] = t_post;
            o[4] = t_rec;
        }
        return;
    }
    // -------------------------------------------------- consumers --------------------------------------------------
    const int w = tid >> 6, lane = tid & 63;
    const int c = tid - 64;                          // column of Y
    const int kq = lane >> 4, cl = lane & 15;
    // column recurrence state
    double y0 = 0.0, y1 = 0.0, y2 = 0.0, zacc = 0.0;
    const int KP = p.KP, myj = (c >> 1) < FC_LMAX ? (c >> 1) : 0, myd = c & 1;
    const bool is_z = c == m2;                       // the gradient column: its right-hand side is gp, it has no factors (s_idx[.][M] stays -1)
    unsigned long long tc[3] = {0, 0, 0}, tprev = p.prof ? wall_clock64() : 0ull;
    const int stamp_tid = 64 * (1 + ((p.syrk_notrim >> 8) & 7));   // debug: the consumer wavefront whose phases are timed (SLAM_PGS_NOTRIM bits 8-10; default wavefront 1)
#define FC_STAMP(i) do { if (p.prof && tid == stamp_tid) { const unsigned long long now_ = wall_clock64(); tc[i] += now_ - tprev; tprev = now_; } } while (0)
    auto columns = [&](int it) {
        if (it >= 1 && it <= nch && c <= m2) {       // column recurrence of chunk it - 1 -> s_yb[(it - 1) & 1]
            const int base = (it - 1) * FC_P;
            const int buf = (it - 1) & 1;
            const double (*rg)[18] = s_ring[buf];
            double* yo = s_yb + (size_t)buf * FC_ROWS * ldl + c;
            const double* Eq = reinterpret_cast<const double*>(&s_E[buf][0]) + myd;
            // branch-free: a column without a factor at pose i adds the all-zero block, poses past N have a zero ring entry
            int slot[FC_P];
#pragma unroll
            for (int l = 0; l < FC_P; ++l) {
                const int ent = s_idx[buf][l][myj];
                slot[l] = (ent >> 8) == base + l ? 6 * (l * KP + (ent & 255)) : 6 * FC_P * FC_KP;
            }
#pragma unroll
            for (int l = 0; l < FC_P; ++l) {
                const double* o = rg[l];
                const double* Ek = Eq + slot[l];
                double u0 = is_z ? o[15] : 0.0, u1 = is_z ? o[16] : 0.0, u2 = is_z ? o[17] : 0.0;
                u0 -= (o[6] * y0 + o[7] * y1) + o[8] * y2;      // G is zero for pose 0
                u1 -= (o[9] * y0 + o[10] * y1) + o[11] * y2;
                u2 -= (o[12] * y0 + o[13] * y1) + o[14] * y2;
                u0 += Ek[0]; u1 += Ek[2]; u2 += Ek[4];
                y0 = o[0] * u0;
                y1 = o[1] * u0 + o[2] * u1;
                y2 = (o[3] * u0 + o[4] * u1) + o[5] * u2;
                yo[(3 * l) * ldl] = y0; yo[(3 * l + 1) * ldl] = y1; yo[(3 * l + 2) * ldl] = y2;
            }
        }
        FC_STAMP(0);
    };
    auto zdot = [&](int it) {                        // the lane's term of row 2M over chunk it - 2 (complete in s_yb[it & 1])
        if (it >= 2 && hb == 0 && c <= m2) {
            const double* cbuf = s_yb + (size_t)(it & 1) * FC_ROWS * ldl;
#pragma unroll
            for (int r = 0; r < FC_ROWS; ++r) zacc = fma(cbuf[r * ldl + m2], cbuf[r * ldl + c], zacc);
        }
    };
    if (w == 4) {
        // ------------- wavefront 4: columns + the bearing-range blocks of the chunk the producer is working on -------------
        const int KP = p.KP;
        const int32_t* cntb = p.cnt + (size_t)b * p.N_max;
        const int32_t* mlmb = p.mlm + (size_t)b * p.N_max * KP;
        const dbl2_t* Eb2 = reinterpret_cast<const dbl2_t*>(p.E + (size_t)b * p.N_max * KP * 6);
        dbl2_t ev[FC_NE];
        int fl[FC_NF], fc[FC_NF];
        auto load_ev = [&](int ch) {                // the chunk's factor slots: landmark, count of its pose, E blocks
            const int base = ch * FC_P;
            const int nq = ((N - base) < FC_P ? (N - base) : FC_P) * KP;
#pragma unroll
            for (int u = 0; u < FC_NF; ++u) {
                const int q = lane + 64 * u;
                fl[u] = 0; fc[u] = 0;
                if (q < nq) { fl[u] = mlmb[(size_t)base * KP + q]; fc[u] = cntb[base + q / KP]; }
            }
#pragma unroll
            for (int u = 0; u < FC_NE; ++u) {
                const int v = lane + 64 * u;
                ev[u] = (dbl2_t){0.0, 0.0};
                if (v < 3 * nq) ev[u] = Eb2[(size_t)base * KP * 3 + v];
            }
        };
        auto store_ev = [&](int ch) {
            const int base = ch * FC_P, buf = ch & 1;
#pragma unroll
            for (int u = 0; u < FC_NE; ++u) s_E[buf][lane + 64 * u] = ev[u];
#pragma unroll
            for (int u = 0; u < FC_NF; ++u) {
                const int q = lane + 64 * u, l = q / KP, sl = q - l * KP;
                // the loaded words are first touched HERE: without the barrier the compiler masks / compares them where they
                // are loaded, i.e. waits for HBM before the recursion instead of after it (1 us per chunk)
                int f = fl[u], n = fc[u];
                asm volatile("" : "+v"(f), "+v"(n) : : "memory");
                if (sl < n) s_idx[buf][l][f & (kPgsFirstBit - 1)] = ((base + l) << 8) | sl;
            }
        };
#pragma unroll 1
        for (int it = 0; it <= nch + 1; ++it) {
            if (it < nch) load_ev(it);
            columns(it);
            zdot(it);
            if (it < nch) store_ev(it);
            __syncthreads();
            if (s_fail) break;
        }
    } else {
        // ------------------------------------- wavefronts 1-3, 5-7: columns + tiles -------------------------------------
        const int nt = (m2 + 31) >> 5, ntile = nt * (nt + 1) / 2;   // tiles over the landmark rows; row 2M is the VALU's
        const int mw = (w < 4 ? w - 1 : w - 2) * FC_NB + hb;   // MFMA wavefront number within the instance (wavefronts 1-3, 5-7)
        const int32_t* lmf = p.lm_first + (size_t)b * p.L_max;
        const bool trim = !(p.syrk_notrim & 1);
        // tile descriptors are wavefront-uniform: kept in SGPRs (readfirstlane) so that the phase below branches on scalars and the
        // operand reads of a tile can all be issued ahead of its MFMAs
        int rowbase[NS], colbase[NS], k0[NS][2];
        bool have[NS];
        dbl4_t acc[NS][2][2];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int t = __builtin_amdgcn_readfirstlane(mw + FC_NW * s);
            have[s] = t < ntile;
            int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (ti * (ti + 1) / 2 > t) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            const int tj = t - ti * (ti + 1) / 2;
            rowbase[s] = have[s] ? 32 * ti : 0; colbase[s] = have[s] ? 32 * tj : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rb = rowbase[s] + 16 * h;
                int kk = 0x7fffffff;                    // first row of Y where this half of the tile can be non-zero
                if (have[s] && rb < m2) kk = trim ? 3 * lmf[rb >> 1] : 0;
                k0[s][h] = __builtin_amdgcn_readfirstlane(kk);
            }
            rowbase[s] = __builtin_amdgcn_readfirstlane(rowbase[s]); colbase[s] = __builtin_amdgcn_readfirstlane(colbase[s]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[s][i][j] = (dbl4_t){0.0, 0.0, 0.0, 0.0};
        }
        auto tiles = [&](int it) {
            if (it >= 2) {                               // chunk it - 2 is complete in s_yb[it & 1]: tiles + right-hand-side row
                const double* cbuf = s_yb + (size_t)(it & 1) * FC_ROWS * ldl;
                const int kend = (it - 1) * FC_ROWS;     // one past the chunk's last row of Y
                const double* src = cbuf + kq * ldl + cl;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (kend <= k0[s][0]) continue;                          // scalar
                    const double* sa = src + rowbase[s];
                    const double* sb = src + colbase[s];
                    double a0[FC_ROWS / 4], b0[FC_ROWS / 4], b1[FC_ROWS / 4];
#pragma unroll
                    for (int ks = 0; ks < FC_ROWS / 4; ++ks) { a0[ks] = sa[ks * 4 * ldl]; b0[ks] = sb[ks * 4 * ldl]; b1[ks] = sb[ks * 4 * ldl + 16]; }
                    if (kend > k0[s][1]) {                                   // scalar: rows 16..31 of the tile have begun
                        double a1[FC_ROWS / 4];
#pragma unroll
                        for (int ks = 0; ks < FC_ROWS / 4; ++ks) a1[ks] = sa[ks * 4 * ldl + 16];
#pragma unroll
                        for (int ks = 0; ks < FC_ROWS / 4; ++ks) {
                            acc[s][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ks], b0[ks], acc[s][0][0], 0, 0, 0);
                            acc[s][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ks], b1[ks], acc[s][0][1], 0, 0, 0);
                            acc[s][1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ks], b0[ks], acc[s][1][0], 0, 0, 0);
                            acc[s][1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[ks], b1[ks], acc[s][1][1], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int ks = 0; ks < FC_ROWS / 4; ++ks) {
                            acc[s][0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ks], b0[ks], acc[s][0][0], 0, 0, 0);
                            acc[s][0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[ks], b1[ks], acc[s][0][1], 0, 0, 0);
                        }
                    }
                }
            }
            FC_STAMP(1);
        };
#pragma unroll 1
        for (int it = 0; it <= nch + 1; ++it) {
            columns(it);
            tiles(it);
            zdot(it);
            __syncthreads();
            FC_STAMP(2);
            if (s_fail) break;
        }
        if (p.prof && tid == stamp_tid) {
            unsigned long long* o = p.prof + (size_t)p.B * p.lanes_max * 8 + (size_t)b * 16 + 8 * hb;
            o[5] = tc[0]; o[6] = tc[1]; o[7] = tc[2];
        }
        if (!s_fail) {
            const double lambda = p.lambda[b];
            const double* Db = p.D + (size_t)b * p.L_max * 3;
            double* Sb = p.S + (size_t)b * LD * LD;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (!have[s]) continue;
                if (rowbase[s] == colbase[s]) {              // scalar: only a diagonal tile holds elements of D + lambda I
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int r = rowbase[s] + 16 * i + kq + 4 * r4;   // C/D layout of the f64 MFMA: row = (lane>>4) + 4*reg
                            const int rr = r < m2 ? r : 0;
                            const double dd = Db[3 * (rr >> 1) + ((rr & 1) ? 2 : 0)] + lambda, dx = Db[3 * (rr >> 1) + 1];
#pragma unroll
                            for (int j = 0; j <= i; ++j) {
                                const int cc = colbase[s] + 16 * j + cl;
                                if (r >= m2 || cc > r) continue;
                                double v = -acc[s][i][j][r4];
                                if (cc == r) v += dd;
                                else if ((cc >> 1) == (r >> 1)) v += dx;
                                Sb[(size_t)r * LD + cc] = v;
                            }
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4) {
                                const int r = rowbase[s] + 16 * i + kq + 4 * r4;
                                const int cc = colbase[s] + 16 * j + cl;
                                if (r < m2) Sb[(size_t)r * LD + cc] = -acc[s][i][j][r4];   // below the diagonal: cc < r, cc < 2M
                            }
                }
            }
        }
    }
#undef FC_STAMP
    if (!s_fail && hb == 0 && c <= m2) {
        const double* glb = p.gl + (size_t)b * p.L_max * 2;
        p.S[(size_t)b * LD * LD + (size_t)m2 * LD + c] = (c < m2 ? glb[c] : 0.0) - zacc;
    }
}

// Dense blocked Cholesky of S (2M x 2M, lower, in place; the right-hand-side row 2M rides along as one more panel row,
// which IS the forward substitution) followed by the blocked backward substitution; dl = S^-1 rhs.
// CTPB threads per instance: 1024 when few instances are active (the factorisation is a chain of short latency-bound
// phases: more wavefronts shorten each), 256 when many are (more instances resident per CU).
template <int CTPB>
__global__ __launch_bounds__(CTPB) void pgs_chol_kernel(const PgsParams p) {
    constexpr int NB = 16, NBL = 4;   // panel width: fewer, fatter panel steps (each costs several HBM/L2 round trips)
    extern __shared__ double s_dyn[];
    __shared__ double s_d[NB][NB + 1];
    __shared__ double s_diag[NB], s_rdiag[NB];
    __shared__ int s_fail;
    const int b = pgs_slot(p, blockIdx.x), tid = threadIdx.x;
    if (p.state[b] || !p.solve_ok[b]) return;
    const int LD = p.LD, m2 = 2 * p.M[b];
    if (m2 == 0) return;
    double* Sb = p.S + (size_t)b * LD * LD;
    double* s_p = s_dyn;                 // panel [(rows below the block)][NB + 1]
    double* s_y = s_dyn;                 // backward phase: y / x [m2]
    if (tid == 0) s_fail = 0;
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = p.prof ? wall_clock64() : 0ull;
#define PGS_STAMP(i) do { if (p.prof && tid == 0) { const unsigned long long now_ = wall_clock64(); tacc[i] += now_ - tprev; tprev = now_; } } while (0)
    __syncthreads();
    for (int j0 = 0; j0 < m2; j0 += NB) {
        const int nb = (m2 - j0) < NB ? (m2 - j0) : NB;
        {
            const int r = tid >> NBL, c = tid & (NB - 1);
            if (r < nb && c <= r) s_d[r][c] = Sb[(size_t)(j0 + r) * LD + j0 + c];
        }
        __syncthreads();
        PGS_STAMP(0);
        {   // factor the diagonal block on an NB x NB thread grid: column by column, two barriers each.  The diagonal
            // keeps its un-rooted pivot until the end; sqrt(pivot) and its reciprocal go to s_diag / s_rdiag.
            const int r = tid >> NBL, c2 = tid & (NB - 1);
            for (int c = 0; c < nb; ++c) {
                if (tid < NB * NB && c2 == c && r >= c && r < nb) {
                    const double d = s_d[c][c];
                    if (r == c) {
                        if (!(d > 0.0)) s_fail = 1;
                        const double sd = sqrt(d > 0.0 ? d : 1.0);
                        s_diag[c] = sd; s_rdiag[c] = 1.0 / sd;
                    } else {
                        s_d[r][c] = s_d[r][c] / sqrt(d > 0.0 ? d : 1.0);
                    }
                }
                __syncthreads();
                if (tid < NB * NB && r > c && c2 > c && c2 <= r && r < nb) s_d[r][c2] = s_d[r][c2] - s_d[r][c] * s_d[c2][c];
                __syncthreads();
            }
            if (tid < nb) s_d[tid][tid] = s_diag[tid];
        }
        __syncthreads();
        {   // write the factored block back
            const int r = tid >> NBL, c = tid & (NB - 1);
            if (r < nb && c <= r) Sb[(size_t)(j0 + r) * LD + j0 + c] = s_d[r][c];
        }
        PGS_STAMP(1);
        const int rb = j0 + nb;              // first row below the block
        const int R = m2 + 1 - rb;           // rows below, including the rhs row
        for (int rr = tid; rr < R; rr += CTPB) {   // panel: row (rb + rr) <- row * L_block^-T
            double* row = Sb + (size_t)(rb + rr) * LD + j0;
            double x[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) x[c] = c < nb ? row[c] : 0.0;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (c < nb) {
                    double v = x[c];
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        if (k < c) v -= x[k] * s_d[c][k];
                    x[c] = v * s_rdiag[c];
                }
                asm volatile("" ::: "memory");   // keep the LDS reads of later columns from being hoisted (register pressure)
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (c < nb) row[c] = x[c];
                s_p[rr * (NB + 1) + c] = x[c];
            }
        }
        __syncthreads();
        PGS_STAMP(2);
        // trailing update  C -= P P^T  on 16x16 tiles of the lower triangle below the block (rhs row included) with
        // v_mfma_f64_16x16x4_f64: NB / 4 k-steps per tile, operands from the LDS panel, C read-modify-written in HBM/L2
        {
            const int nt = (R + 15) >> 4;
            const int ntiles = nt * (nt + 1) / 2;
            const int w = tid >> 6, lane = tid & 63, kq = lane >> 4, cl = lane & 15;
            constexpr int NW = CTPB / 64, TG = 4;   // TG tiles per wavefront in flight (their C loads are issued together)
            for (int t0 = w; t0 < ntiles; t0 += NW * TG) {
                dbl4_t acc[TG];
                int trs[TG], tcs[TG];
#pragma unroll
                for (int g = 0; g < TG; ++g) {
                    const int t = t0 + g * NW;
                    int tr = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
                    while (tr * (tr + 1) / 2 > t) --tr;
                    while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
                    trs[g] = tr; tcs[g] = t - tr * (tr + 1) / 2;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int r = rb + 16 * tr + kq + 4 * r4, c = rb + 16 * tcs[g] + cl;
                        acc[g][r4] = (t < ntiles && r <= m2 && c <= r && c < m2) ? Sb[(size_t)r * LD + c] : 0.0;
                    }
                }
#pragma unroll
                for (int g = 0; g < TG; ++g) {
                    if (t0 + g * NW >= ntiles) continue;
                    const double* pa = s_p + (16 * trs[g] + cl) * (NB + 1) + kq;
                    const double* pb = s_p + (16 * tcs[g] + cl) * (NB + 1) + kq;
#pragma unroll
                    for (int q = 0; q < NB / 4; ++q) acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * q], pb[4 * q], acc[g], 0, 0, 0);
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int r = rb + 16 * trs[g] + kq + 4 * r4, c = rb + 16 * tcs[g] + cl;
                        if (r <= m2 && c <= r && c < m2) Sb[(size_t)r * LD + c] = acc[g][r4];
                    }
                }
            }
        }
        __syncthreads();
        PGS_STAMP(3);
    }
    if (s_fail) { if (tid == 0) p.solve_ok[b] = 0; return; }
    // backward substitution  L^T x = y  (y = row 2M of the factored matrix), blocks from the bottom
    for (int c = tid; c < m2; c += CTPB) s_y[c] = Sb[(size_t)m2 * LD + c];
    __syncthreads();
    const int nblk = (m2 + NB - 1) / NB;
    for (int bi = nblk - 1; bi >= 0; --bi) {
        const int j0 = bi * NB;
        const int nb = (m2 - j0) < NB ? (m2 - j0) : NB;
        {
            const int r = tid >> NBL, c = tid & (NB - 1);
            if (r < nb && c <= r) s_d[r][c] = Sb[(size_t)(j0 + r) * LD + j0 + c];
        }
        __syncthreads();
        if (tid < 64) {   // lane k owns y[j0 + k]; x_c is broadcast from lane c
            double yk = tid < nb ? s_y[j0 + tid] : 0.0;
            for (int c = nb - 1; c >= 0; --c) {
                const double xc = __shfl(yk, c, 64) / s_d[c][c];
                if (tid == c) yk = xc;
                if (tid < c) yk -= s_d[c][tid] * xc;
            }
            if (tid < nb) s_y[j0 + tid] = yk;
        }
        __syncthreads();
        for (int c = tid; c < j0; c += CTPB) {   // y[c] -= sum_k L[j0+k][c] x[j0+k]  (rows of L: coalesced over c)
            double v = s_y[c];
            for (int k = 0; k < nb; ++k) v -= Sb[(size_t)(j0 + k) * LD + c] * s_y[j0 + k];
            s_y[c] = v;
        }
        __syncthreads();
    }
    PGS_STAMP(4);
    double* dlb = p.dl + (size_t)b * p.L_max * 2;
    for (int c = tid; c < m2; c += CTPB) dlb[c] = s_y[c];
    if (p.prof && tid == 0)
        for (int i = 0; i < 6; ++i) p.prof[(size_t)b * 8 + i] = tacc[i];
#undef PGS_STAMP
}

// The same factorisation LEFT-LOOKING (round 4).  The right-looking kernel above reads, updates and writes back the whole trailing matrix
// at every panel step: ~22 dependent read-modify-write round trips through L2 per element, a panel solve that must wait for the trailing
// update before it, and 0.18 of the 0.50 ms of a trial in that update alone.  Here panel j is formed when it is needed,
//     C(rows >= j0, 16 columns)  =  S  -  L[rows, 0 : j0] L[j0 : j0+16, 0 : j0]^T ,
// as ONE chain of v_mfma_f64_16x16x4_f64 per 16 x 16 tile (the rows of L it reads were written panels ago; the 16 block rows are staged in
// LDS once per panel for all tiles), stays on chip through the factorisation of its diagonal block and its panel solve, and is written to
// memory once, as L.  Per element the arithmetic is the SAME chain of fused multiply-adds in ascending k as before (the right-looking
// kernel rounds to fp64 between panels exactly where this chain does), the diagonal block and the panel solve are the same code: the
// factor is bit-identical to the right-looking kernel's (SLAM_PGS_CHOL_LL=0 keeps the old one for the comparison).
#ifndef SLAM_PGS_LL_KU
#define SLAM_PGS_LL_KU 4
#endif
// CTPB_ threads: 768 by default since the end of round 5 - three wavefronts per SIMD have 168 registers per lane and the kernel no longer spills (at 1024 threads =
// 128 registers it kept 56 bytes per lane in scratch, most of it around the completion step): 13.4 -> 11.5 ms per solve on one box (docs/dev/sessions/gpu_r5aq.sh), the
// same factor bit for bit.  SLAM_PGS_CHOL_LL=1 keeps the 1024-thread instantiation.
template <int CTPB_>
__global__ __launch_bounds__(CTPB_) void pgs_chol_ll_kernel(const PgsParams p) {
    constexpr int CTPB = CTPB_, NB = 16, NBL = 4, NW = CTPB / 64;
    extern __shared__ double s_dyn[];
    __shared__ double s_diag[NB], s_rdiag[NB];
    __shared__ double s_xi[NB][NB + 1];   // inverse of the current diagonal block
    __shared__ int s_fail;
    const int b = pgs_slot(p, blockIdx.x), tid = threadIdx.x;
    if (p.state[b] || !p.solve_ok[b]) return;
    const int LD = p.LD, m2 = 2 * p.M[b];
    if (m2 == 0) return;
    double* Sb = p.S + (size_t)b * LD * LD;
    // Dynamic LDS, T = 2 (LD + 1) (NB + 1) doubles.  The panel C [rows j0 .. m2][NB + 1] (its first 16 rows are the diagonal block) of an even
    // panel sits at the bottom of it, of an odd panel at the top end: the panel solve leaves L in its panel's buffer, so the NEXT panel completes
    // its tiles (phase F: the last 16 k) out of LDS instead of reading back from memory what has just been written there (a round trip through
    // L2 per panel, 4 of the 14 us of a panel step).  The staged block rows of L [16][ldb] take the opposite end, over the panel before, which is
    // dead once phase F is through: R (NB + 1) + 16 ldb <= (m2 + 1) (NB + 1) + 112 and two consecutive panels need (2 R + 16) (NB + 1) <= T.
    const int T_dbl = 2 * (LD + 1) * (NB + 1);
    double* const s_y = s_dyn;           // backward phase: y / x [m2]
    if (tid == 0) s_fail = 0;
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = p.prof ? wall_clock64() : 0ull;
#define PGS_STAMP(i) do { if (p.prof && tid == 0) { const unsigned long long now_ = wall_clock64(); tacc[i] += now_ - tprev; tprev = now_; } } while (0)
    const int w = tid >> 6, lane = tid & 63, kq = lane >> 4, cl = lane & 15;
    typedef double dbl4v_t __attribute__((ext_vector_type(4)));
    constexpr int KU = SLAM_PGS_LL_KU;                      // 16-k blocks in flight per lane
    constexpr int NTW = NW - 1, TPW = (28 + NTW - 1) / NTW; // wavefronts that own tiles (1 .. NW - 1), tiles per wavefront (nt <= 28: LD <= 448)
    // PIPELINE over the panels.  A panel step is: complete the tiles (the last 16 k), factor the 16 x 16 diagonal block, solve the rows
    // below, write L.  The factorisation of the diagonal block is a 16-step dependent chain - one wavefront's work (wave-synchronous on LDS,
    // no workgroup barrier inside; it had thirty-two of them with sixteen wavefronts waiting at each) - and meanwhile wavefronts 1 .. 15 form
    // the NEXT panel's tiles over every k that is final already (all columns before this panel's), so that when this panel's L is written
    // only four MFMAs per tile are missing.  accn[] carries those partial sums (S minus the sum over k < j0) from one iteration to the next;
    // per element the products are subtracted in the same order as without the pipeline.
    dbl4_t accn[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) accn[q] = dbl4_t{0.0, 0.0, 0.0, 0.0};
    // tile t of the panel that starts at row jb: S entries (lower triangle, columns < m2, rows <= m2) as an MFMA accumulator
    auto tile_init = [&](int jb, int t) -> dbl4_t {
        dbl4_t a;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int r = jb + 16 * t + kq + 4 * r4, c = jb + cl;
            a[r4] = (r <= m2 && c <= r && c < m2) ? Sb[(size_t)r * LD + c] : 0.0;
        }
        return a;
    };
    if (w >= 1) {   // the first panel has no k range: its tiles are S itself
        const int nt0 = (m2 + 1 + 15) >> 4;
#pragma unroll
        for (int q = 0; q < TPW; ++q) { const int t = (w - 1) + NTW * q; if (t < nt0) accn[q] = tile_init(0, t); }
    }
    __syncthreads();
    double dg[NB];   // wavefront 0: row (lane & 15) of the diagonal block being factored
    auto dgl_rd = [](double v, int l) -> double {   // v of lane l as a wave-uniform value (two v_readlane_b32)
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
        return __hiloint2double(hi, lo);
    };
    for (int j0 = 0; j0 < m2; j0 += NB) {
        const int nb = (m2 - j0) < NB ? (m2 - j0) : NB;
        const int R = m2 + 1 - j0;                 // rows of the panel: the block rows, the rows below, the rhs row
        const int nt = (R + 15) >> 4;
        int ldb = (j0 + 3) & ~3;                   // row length of the staged block rows: a multiple of 4 with an odd quotient (bank spread)
        if (((ldb >> 2) & 1) == 0) ldb += 4;
        const bool odd = (j0 >> 4) & 1;
        double* const s_c = odd ? s_dyn + (T_dbl - R * (NB + 1)) : s_dyn;                            // this panel
        const double* const s_p = odd ? s_dyn : s_dyn + (T_dbl - (R + NB) * (NB + 1));              // the panel before (R + 16 rows), holding L
        double* const s_b = odd ? s_dyn : s_dyn + (T_dbl - 16 * ldb);                               // block rows staged for the next panel's tiles
        auto SD = [&](int r, int c) -> double& { return s_c[r * (NB + 1) + c]; };
        // ---- phase F: the tiles of this panel get the last 16 k (columns j0-16 .. j0-1, written by the previous panel's solve) ----
        if (w >= 1) {
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int t = (w - 1) + NTW * q;
                if (t >= nt) continue;
                dbl4_t acc = accn[q];
                if (j0 > 0) {   // rows j0 + 16 t + cl and j0 + cl (clamped to m2) of the previous panel's columns: its buffer's rows 16 + ..
                    const int la = 16 + 16 * t + cl < R + NB ? 16 + 16 * t + cl : R + NB - 1, lb = 16 + cl < R + NB ? 16 + cl : R + NB - 1;
                    const double* __restrict__ pa = s_p + la * (NB + 1) + 4 * kq;
                    const double* __restrict__ pb = s_p + lb * (NB + 1) + 4 * kq;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[qq], pb[qq], acc, 0, 0, 0);
                }
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rl = 16 * t + kq + 4 * r4;
                    if (rl < R) SD(rl, cl) = acc[r4];
                }
            }
        }
        __syncthreads();
        PGS_STAMP(3);   // completion of the panel
        // ---- phases D1 / D2: wavefront 0 factors the diagonal block (columns 0 .. 7, then 8 .. 15); wavefronts 1 .. 15 stage the NEXT
        //      panel's block rows L[j0+16 .. j0+31][0 .. j0) (D1) and run its tiles over k < j0 (D2) ----
        const int jn = j0 + NB;                    // next panel
        const bool has_next = jn < m2;
        const int ntn = has_next ? (m2 + 1 - jn + 15) >> 4 : 0;
        // The diagonal block in wavefront 0's REGISTERS (round 5): lane r (mod 16; the four lane groups hold replicas) keeps row r, the
        // pivot and the column entries l(c2, c) another row needs arrive by v_readlane.  Through LDS - lane = (row, column group), two
        // fenced round trips per column - a column cost ~1 300 cycles, 8.6 us per block, 190 of the 400 us of a factorisation.
        auto diag_cols = [&](auto lo_tag, auto hi_tag) {
            constexpr int c_lo = decltype(lo_tag)::value, c_hi = decltype(hi_tag)::value;
            const int r = lane & 15;
#pragma unroll
            for (int c = c_lo; c < c_hi; ++c) {
                if (c >= nb) break;   // wave-uniform
                const double d = dgl_rd(dg[c], c);   // the pivot: entry c of row c
                // 1 / sqrt(d) by v_rsq_f64 + two Newton steps (~1 ulp, like the pose chain's pivots): the column is scaled by a product, the
                // diagonal entry is d * rs - a square root and a division per column were 280 of its ~500 dependent cycles
                const double rs = rsqrt_nr(d > 0.0 ? d : 1.0);
                if (lane == c) {
                    if (!(d > 0.0)) s_fail = 1;
                    s_diag[c] = d * rs; s_rdiag[c] = rs;
                }
                const double lrc = dg[c] * rs;      // meaningful in the rows below c
                if (r > c) dg[c] = lrc;
#pragma unroll
                for (int c2 = c + 1; c2 < NB; ++c2) {
                    const double l2 = dgl_rd(dg[c], c2);   // l(c2, c), from row c2
                    if (r >= c2) dg[c2] = dg[c2] - lrc * l2;
                }
            }
        };
        if (w == 0) {
            const int r = lane & 15;
#pragma unroll
            for (int c = 0; c < NB; ++c) dg[c] = (r < nb && c <= r) ? SD(r, c) : 0.0;
            diag_cols(std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{});   // all sixteen columns here, the inverse in the second half
        } else if (has_next) {
            for (int e = tid - 64; e < 16 * j0; e += CTPB - 64) {
                const int r = e / j0, k = e - r * j0;
                const int rr = jn + r <= m2 ? jn + r : m2;
                s_b[r * ldb + k] = Sb[(size_t)rr * LD + k];
            }
#pragma unroll
            for (int q = 0; q < TPW; ++q) { const int t = (w - 1) + NTW * q; if (t < ntn) accn[q] = tile_init(jn, t); }
        }
        __syncthreads();
        if (w == 0) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // s_diag of every column is written
            if (lane < NB && lane < nb) {   // row `lane` of the factored block: to LDS for the panel solve, to memory as L
                const int r = lane;
                double* grow = Sb + (size_t)(j0 + r) * LD + j0;
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    const double v = c == r ? s_diag[c] : dg[c];
                    if (c <= r) { SD(r, c) = v; grow[c] = v; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the block's rows are in LDS
            // X = L_block^-1 (lower triangular), column `lane` per lane by forward substitution  x_r = -(sum_{k<r} l(r, k) x_k) / l(r, r)
            // through LDS (l(r, k): one broadcast read; x_k: the lane's own column of s_xi, written by itself).  The panel solve below is then
            // rows * X^T  on the matrix pipe and the backward substitution a product with X^T: round 4 measured both (panel solve 105 -> 18 us,
            // backward 67 -> 44 us per factorisation) and dropped them for the 5 us per block the inverse cost wavefront 0, then the long pole
            // of this phase; behind the register-resident factorisation it fits in the shadow of the other wavefronts' tiles.  (With the
            // column in 16 registers next to dg[] the kernel spilled: 12 us per block.)
            {   // (the column in registers - dg[] is dead by now, so they are free - and l(r, k) as broadcast LDS reads the compiler can issue
                // ahead of the dependent chain; through s_xi in LDS the chain paid a round trip per term: 4 us per block, the long pole)
                const int cx = lane & 15;
                double xv[NB];
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    double a = 0.0;
#pragma unroll
                    for (int k = 0; k < r; ++k) a += SD(r < nb ? r : 0, k) * xv[k];   // (x_k = 0 above the diagonal of X)
                    const double rdr = s_rdiag[r];
                    xv[r] = (r < nb && cx < nb) ? (r == cx ? rdr : (r > cx ? -(a * rdr) : 0.0)) : 0.0;
                }
                if (lane < NB) {
#pragma unroll
                    for (int r = 0; r < NB; ++r) s_xi[r][cx] = xv[r];   // X(r, c): row r, column c = lane
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // s_xi is complete
            if (lane < NB && lane < nb) {   // the strictly lower part of X goes into the (unused) strictly UPPER part of the block in S:
                const int k = lane;         // row j0 + k holds X(c, k), c > k - column k of X, what the backward substitution's lane k needs
                double* grow = Sb + (size_t)(j0 + k) * LD + j0;
#pragma unroll
                for (int c = 0; c < NB; ++c)
                    if (c > k && c < nb) grow[c] = s_xi[c][k];
            }
        } else if (has_next) {
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                const int t = (w - 1) + NTW * q;
                if (t >= ntn) continue;
                dbl4_t acc = accn[q];
                const int ar = jn + 16 * t + cl <= m2 ? jn + 16 * t + cl : m2;   // A-operand row of this lane (clamped)
                const double* __restrict__ arow = Sb + (size_t)ar * LD + 4 * kq;
                const double* __restrict__ brow = s_b + cl * ldb + 4 * kq;
                int k0 = 0;
#pragma unroll 1
                for (; k0 + 16 * KU <= j0; k0 += 16 * KU) {
                    dbl4v_t av[KU];
#pragma unroll
                    for (int u = 0; u < KU; ++u) av[u] = *reinterpret_cast<const dbl4v_t*>(arow + k0 + 16 * u);
#pragma unroll
                    for (int u = 0; u < KU; ++u)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[u][qq], brow[k0 + 16 * u + qq], acc, 0, 0, 0);
                }
#pragma unroll 1
                for (; k0 < j0; k0 += 16) {
                    const dbl4v_t a1 = *reinterpret_cast<const dbl4v_t*>(arow + k0);
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[qq], brow[k0 + qq], acc, 0, 0, 0);
                }
                accn[q] = acc;
            }
        }
        __syncthreads();
        PGS_STAMP(1);   // diagonal block (+ the next panel's tiles beside it)
        // (Round 4 measured the panel solve against the INVERSE of the diagonal block and dropped it: forming the inverse through LDS cost
        // wavefront 0, then the long pole of the diagonal phase, 5 us per block - 414 -> 457 us in all.  Round 5 forms it in registers.)
        // panel solve: rows * L_block^-T = rows * X^T, 16 x 16 tiles of the rows below the block as four MFMAs each (A = the rows of the
        // panel in LDS, B = X), written to memory as L.  (A thread per row walked a 16-step forward substitution out of LDS: 105 us per
        // factorisation.)
        for (int t = (nb == NB ? 1 : 0) + w; t < nt; t += NW) {   // (tile 0 = the block itself, unless the block is short: then it also holds rows below it)
            dbl4_t acc = dbl4_t{0.0, 0.0, 0.0, 0.0};
            const int ar = 16 * t + cl < R ? 16 * t + cl : R - 1;   // (rows past the panel: clamped, never stored)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(SD(ar, 4 * q + kq), s_xi[cl][4 * q + kq], acc, 0, 0, 0);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rl = 16 * t + kq + 4 * r4;
                if (rl >= nb && rl < R && cl < nb) {
                    Sb[(size_t)(j0 + rl) * LD + j0 + cl] = acc[r4];
                    SD(rl, cl) = acc[r4];   // L stays in the panel's buffer for the next panel's phase F (this wavefront has read the tile's rows above)
                }
            }
        }
        __syncthreads();   // L of this panel is in memory before the next panel's tiles read its columns; s_c is free again
        PGS_STAMP(2);
    }
    if (s_fail) { if (tid == 0) p.solve_ok[b] = 0; return; }
    // backward substitution  L^T x = y  (y = row 2M of the factored matrix), blocks from the bottom.  Nothing a block step loads depends on
    // the solution so far, so the loads leave the dependent chain: the 16 rows of L a thread needs for the update of its y[c] are requested
    // BEFORE the block's 16-step solve and used after it, the next block's diagonal block one iteration ahead (the right-looking kernel's
    // loop paid two memory round trips per block: 5.9 of its 6 us); the solve multiplies by reciprocals of the diagonal formed in parallel.
    __shared__ double s_d[NB][NB + 1];
    for (int c = tid; c < m2; c += CTPB) s_y[c] = Sb[(size_t)m2 * LD + c];
    const int nblk = (m2 + NB - 1) / NB;
    double dreg = 0.0;
    {
        const int j0 = (nblk - 1) * NB, nb = m2 - j0;
        const int r = tid >> NBL, c = tid & (NB - 1);
        if (tid < NB * NB && r < nb && c < nb) dreg = Sb[(size_t)(j0 + r) * LD + j0 + c];   // the whole block: L below / on the diagonal, X above it
    }
    __syncthreads();
    for (int bi = nblk - 1; bi >= 0; --bi) {
        const int j0 = bi * NB;
        const int nb = (m2 - j0) < NB ? (m2 - j0) : NB;
        {
            const int r = tid >> NBL, c = tid & (NB - 1);
            if (tid < NB * NB && r < nb && c < nb) s_d[r][c] = dreg;
        }
        double lrow[NB];                       // L[j0 + k][c] for this thread's column c < j0 (m2 <= CTPB: one column per thread)
#pragma unroll
        for (int k = 0; k < NB; ++k) lrow[k] = (tid < j0 && k < nb) ? Sb[(size_t)(j0 + k) * LD + tid] : 0.0;
        if (bi > 0) {                          // the next block's diagonal block
            const int r = tid >> NBL, c = tid & (NB - 1);
            dreg = (tid < NB * NB) ? Sb[(size_t)(j0 - NB + r) * LD + j0 - NB + c] : 0.0;
        }
        __syncthreads();
        if (tid < 64) {   // x_block = X^T y_block: lane k sums column k of X (the block's strictly upper part in S holds it, row k) against y
            double xk = 0.0;
            if (tid < nb) {
                xk = (1.0 / s_d[tid][tid]) * s_y[j0 + tid];   // X(k, k) = 1 / l(k, k)
#pragma unroll
                for (int c = 1; c < NB; ++c)
                    if (c > tid && c < nb) xk += s_d[tid][c] * s_y[j0 + c];   // X(c, k), staged from S[j0 + k][j0 + c]
            }
            __builtin_amdgcn_wave_barrier();   // every lane has read y before any lane overwrites it
            if (tid < nb) s_y[j0 + tid] = xk;
        }
        __syncthreads();
        if (tid < j0) {   // y[c] -= sum_k L[j0+k][c] x[j0+k]
            double v = s_y[tid];
#pragma unroll
            for (int k = 0; k < NB; ++k) v -= lrow[k] * s_y[j0 + (k < nb ? k : 0)];
            s_y[tid] = v;
        }
        __syncthreads();
    }
    PGS_STAMP(4);
    double* dlb = p.dl + (size_t)b * p.L_max * 2;
    for (int c = tid; c < m2; c += CTPB) dlb[c] = s_y[c];
    if (p.prof && tid == 0)
        for (int i = 0; i < 6; ++i) p.prof[(size_t)b * 8 + i] = tacc[i];
#undef PGS_STAMP
}

// Pose step: H_pp dp = gp - E dl through the chain factor: forward  z_i = v_i - M_i z_{i-1}  (v = Linv u, M = Linv G),
// backward  d_i = w_i - N_i d_{i+1}  (w = Linv^T z, N = Linv^T G_{i+1}^T).  Both are affine recurrences in a 3-vector,
// so they are evaluated as a SCAN instead of 2 x N dependent steps: every thread prepares (v, M) of its poses, then one
// wavefront composes the maps of 64 contiguous blocks (sequentially inside a block), scans the 64 composites with
// lane shuffles, and replays its block from the scanned entry value.  ~2 x (N/64 + 6) dependent steps instead of 2 N.
struct Affine3 { double a[3], B[9]; };   // z -> a + B z
__device__ __forceinline__ void affine_step(Affine3& f, const double* W) {   // f <- (z -> v - M z) o f, W = {v[3], M[9]}
    double na[3], nB[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        na[r] = W[r] - ((W[3 + 3 * r] * f.a[0] + W[4 + 3 * r] * f.a[1]) + W[5 + 3 * r] * f.a[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            nB[3 * r + c] = -((W[3 + 3 * r] * f.B[c] + W[4 + 3 * r] * f.B[3 + c]) + W[5 + 3 * r] * f.B[6 + c]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) f.a[k] = na[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) f.B[k] = nB[k];
}
// cur <- cur o prev  (prev is applied first)
__device__ __forceinline__ void affine_compose(Affine3& cur, const Affine3& prev) {
    double na[3], nB[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        na[r] = cur.a[r] + ((cur.B[3 * r] * prev.a[0] + cur.B[3 * r + 1] * prev.a[1]) + cur.B[3 * r + 2] * prev.a[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            nB[3 * r + c] = (cur.B[3 * r] * prev.B[c] + cur.B[3 * r + 1] * prev.B[3 + c]) + cur.B[3 * r + 2] * prev.B[6 + c];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) cur.a[k] = na[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) cur.B[k] = nB[k];
}
// One wavefront: x_i = W_i.v - W_i.M x_{i-1} over i = 0..N-1 (REV: i = N-1..0 with x_N = 0), x written to out[3 i].
// W [N][12] in HBM/L2 (just written by this workgroup).
template <bool REV>
__device__ __forceinline__ void affine_scan_wave(const double* W, double* out, int N, int lane) {
    const int BL = (N + 63) / 64;
    const int blk = REV ? 63 - lane : lane;            // block blk covers poses [blk*BL, min(N, (blk+1)*BL))
    const int lo = blk * BL, hi = (lo + BL) < N ? (lo + BL) : N;
    Affine3 f;
#pragma unroll
    for (int k = 0; k < 3; ++k) f.a[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) f.B[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (lo < N) {
        if (!REV) { for (int i = lo; i < hi; ++i) affine_step(f, W + 12 * (size_t)i); }
        else { for (int i = hi - 1; i >= lo; --i) affine_step(f, W + 12 * (size_t)i); }
    }
    // inclusive scan in processing order (lane 0 first)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Affine3 pv;
#pragma unroll
        for (int k = 0; k < 3; ++k) pv.a[k] = __shfl_up(f.a[k], off, 64);
#pragma unroll
        for (int k = 0; k < 9; ++k) pv.B[k] = __shfl_up(f.B[k], off, 64);
        if (lane >= off) affine_compose(f, pv);
    }
    // entry value of this lane's block = composite of all earlier blocks applied to 0 = their `a`
    double x0 = __shfl_up(f.a[0], 1, 64), x1 = __shfl_up(f.a[1], 1, 64), x2 = __shfl_up(f.a[2], 1, 64);
    if (lane == 0) { x0 = 0.0; x1 = 0.0; x2 = 0.0; }
    if (lo < N) {
        for (int t = 0; t < hi - lo; ++t) {
            const int i = REV ? hi - 1 - t : lo + t;
            const double* w = W + 12 * (size_t)i;
            const double n0 = w[0] - ((w[3] * x0 + w[4] * x1) + w[5] * x2);
            const double n1 = w[1] - ((w[6] * x0 + w[7] * x1) + w[8] * x2);
            const double n2 = w[2] - ((w[9] * x0 + w[10] * x1) + w[11] * x2);
            x0 = n0; x1 = n1; x2 = n2;
            out[3 * i] = x0; out[3 * i + 1] = x1; out[3 * i + 2] = x2;
        }
    }
}

constexpr int BTPB = 256;
__global__ __launch_bounds__(BTPB) void pgs_backsolve_kernel(const PgsParams p) {
    const int b = pgs_slot(p, blockIdx.x), tid = threadIdx.x;
    if (p.state[b] || !p.solve_ok[b]) return;
    const int N = pgs_N(p, b), KP = p.KP;
    const Inst g = inst_view(p, b);
    const double* gpb = p.gp + (size_t)b * p.N_max * 3;
    const double* Eb = p.E + (size_t)b * p.N_max * KP * 6;
    const double* Lb = p.Linv + (size_t)b * p.N_max * 6;
    const double* Gb = p.G + (size_t)b * p.N_max * 9;
    const double* dlb = p.dl + (size_t)b * p.L_max * 2;
    double* dpb = p.dp + (size_t)b * p.N_max * 3;
    double* Wb = p.Y + (size_t)b * p.y_stride;    // Y is dead once S has been formed: scratch for the (v, M) records
    for (int i = tid; i < N; i += BTPB) {          // forward records
        double u0 = gpb[3 * i], u1 = gpb[3 * i + 1], u2 = gpb[3 * i + 2];
        const int kc = g.cnt[i];
        for (int s = 0; s < kc; ++s) {
            const size_t k = (size_t)i * KP + s;
            const int j = g.mlm[k] & (kPgsFirstBit - 1);
            const double* E = Eb + 6 * k;
            const double d0 = dlb[2 * j], d1 = dlb[2 * j + 1];
            u0 -= E[0] * d0 + E[1] * d1; u1 -= E[2] * d0 + E[3] * d1; u2 -= E[4] * d0 + E[5] * d1;
        }
        const double* I = Lb + 6 * i;
        const double* G = Gb + 9 * i;      // G_0 = 0
        double* W = Wb + 12 * (size_t)i;
        W[0] = I[0] * u0;
        W[1] = I[1] * u0 + I[2] * u1;
        W[2] = (I[3] * u0 + I[4] * u1) + I[5] * u2;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            W[3 + cc] = I[0] * G[cc];
            W[6 + cc] = I[1] * G[cc] + I[2] * G[3 + cc];
            W[9 + cc] = (I[3] * G[cc] + I[4] * G[3 + cc]) + I[5] * G[6 + cc];
        }
    }
    __syncthreads();
    if (tid < 64) affine_scan_wave<false>(Wb, dpb, N, tid);      // z into dp
    __syncthreads();
    for (int i = tid; i < N; i += BTPB) {          // backward records: w = Linv^T z, Nx = Linv^T G_{i+1}^T
        const double* I = Lb + 6 * i;
        const double zz0 = dpb[3 * i], zz1 = dpb[3 * i + 1], zz2 = dpb[3 * i + 2];
        double* W = Wb + 12 * (size_t)i;
        W[0] = (I[0] * zz0 + I[1] * zz1) + I[3] * zz2;
        W[1] = I[2] * zz1 + I[4] * zz2;
        W[2] = I[5] * zz2;
        if (i + 1 < N) {
            const double* G = Gb + 9 * (i + 1);
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {   // column cc of G^T = row cc of G
                W[3 + cc] = (I[0] * G[3 * cc] + I[1] * G[3 * cc + 1]) + I[3] * G[3 * cc + 2];
                W[6 + cc] = I[2] * G[3 * cc + 1] + I[4] * G[3 * cc + 2];
                W[9 + cc] = I[5] * G[3 * cc + 2];
            }
        } else {
#pragma unroll
            for (int cc = 0; cc < 9; ++cc) W[3 + cc] = 0.0;
        }
    }
    __syncthreads();
    if (tid < 64) affine_scan_wave<true>(Wb, dpb, N, tid);       // dp
}

// p * Pose2(v): the retraction of one pose (the same expressions wherever a candidate pose is formed)
__device__ __forceinline__ void retract_pose(const double* ps, const double* d, double out[3]) {
    double s, c;
    det_sincos(ps[2], &s, &c);
    out[0] = ps[0] + (c * d[0] - s * d[1]);
    out[1] = ps[1] + (s * d[0] + c * d[1]);
    out[2] = remainder(ps[2] + d[2], kTwoPi);
}

// Evaluation, part 1: one thread per FACTOR (see pgs_lin_factor_kernel) - its two terms of the linearised cost 0.5 |J delta + e|^2 at the
// current values and its term of the true cost at the candidate (the factor forms the candidate pose / landmark itself, with the
// expressions part 2 stores them with).  PF[slot] = {0.5 v_0^2, 0.5 v_1^2, 0.5 |e(candidate)|^2}; part 2 adds them where the one-kernel
// version added them (bit-identical sums).
__global__ __launch_bounds__(LF_TPB) void pgs_eval_factor_kernel(const PgsParams p) {
    const int nfb = (p.nfact_max + LF_TPB - 1) / LF_TPB;
    const int bl = blockIdx.x / nfb, fb = blockIdx.x - bl * nfb;
    const int b = pgs_slot(p, bl);
    if (p.state[b] || !p.solve_ok[b]) return;
    const int e = fb * LF_TPB + threadIdx.x;
    const int M = p.M[b], KP = p.KP;
    if (e >= p.evt_start[(size_t)b * (p.L_max + 1) + M]) return;
    const Inst g = inst_view(p, b);
    const int i = p.evt_pose[(size_t)b * p.N_max * KP + e];
    const size_t k = (size_t)p.evt_slot[(size_t)b * p.N_max * KP + e];
    const double* pose = p.pw + (size_t)b * p.N_max * 3 + 3 * i;
    const double* dp = p.dp + (size_t)b * p.N_max * 3 + 3 * i;
    const int j = g.mlm[k] & (kPgsFirstBit - 1);
    const double* lm = p.lw + (size_t)b * p.L_max * 2 + 2 * j;
    const double* dl = p.dl + (size_t)b * p.L_max * 2 + 2 * j;
    const double bb = g.mb[k], rr = g.mr[k];
    double e2[2], Jp[6], Jl[4];
    bearing_range_factor<true>(p, pose, lm, bb, rr, e2, Jp, Jl);
    double* PF = p.PF + ((size_t)b * p.N_max * KP + k) * 12;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double v = (e2[r] + ((Jp[3 * r] * dp[0] + Jp[3 * r + 1] * dp[1]) + Jp[3 * r + 2] * dp[2])) + (Jl[2 * r] * dl[0] + Jl[2 * r + 1] * dl[1]);
        PF[r] = 0.5 * v * v;
    }
    double pn[3], ln[2], en[2];
    retract_pose(pose, dp, pn);
    ln[0] = lm[0] + dl[0]; ln[1] = lm[1] + dl[1];
    bearing_range_factor<false>(p, pn, ln, bb, rr, en, nullptr, nullptr);
    PF[2] = 0.5 * (en[0] * en[0] + en[1] * en[1]);
}

// linearised cost of the step, retraction, true cost of the candidate (part 2: the prior / between factors and the sums); GTSAM's
// tryLambda / iterate / defaultOptimize decisions follow in pgs_decide_kernel (LevenbergMarquardtOptimizer.cpp, NonlinearOptimizer.cpp).
__global__ __launch_bounds__(TPB) void pgs_evaluate_kernel(const PgsParams p) {
    __shared__ double s_buf[TPB];
    const int b = pgs_slot(p, blockIdx.x), tid = threadIdx.x;
    if (p.state[b]) return;
    const int N = pgs_N(p, b), KP = p.KP, M = p.M[b];
    const Inst g = inst_view(p, b);
    double* pose = p.pw + (size_t)b * p.N_max * 3;
    double* lm = p.lw + (size_t)b * p.L_max * 2;
    double* pose_n = p.pn + (size_t)b * p.N_max * 3;
    double* lm_n = p.ln + (size_t)b * p.L_max * 2;
    const double* dp = p.dp + (size_t)b * p.N_max * 3;
    const double* dl = p.dl + (size_t)b * p.L_max * 2;
    const double* PFb = p.PF + (size_t)b * p.N_max * KP * 12;
    const bool ok = p.solve_ok[b] != 0;
    double newLin = 0.0, newError = 0.0;
    if (ok) {
        double acc = 0.0;
        for (int i = tid; i < N; i += TPB) {   // 0.5 |J delta + e|^2 of the UNDAMPED linearisation
            double e[3], J1[9];
            if (i == 0) {
                prior_factor(p, pose, e);
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double v = e[k] + p.w_prior[k] * dp[k]; acc = acc + 0.5 * v * v; }
            }
            if (i + 1 < N) {
                between_factor<true>(p, pose + 3 * i, pose + 3 * (i + 1), p.cmds[2 * i], p.cmds[2 * i + 1], e, J1);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double v = (e[r] + ((J1[3 * r] * dp[3 * i] + J1[3 * r + 1] * dp[3 * i + 1]) + J1[3 * r + 2] * dp[3 * i + 2])) + p.w_btw[r] * dp[3 * (i + 1) + r];
                    acc = acc + 0.5 * v * v;
                }
            }
            const int kc = g.cnt[i];
            const double* PF = PFb + (size_t)i * KP * 12;
            constexpr int UB = 8;   // the factors' terms (pgs_eval_factor_kernel) are fetched eight factors at a time, added in slot order
            int s = 0;
#pragma unroll 1
            for (; s + UB <= kc; s += UB) {
                double w[UB][2];
#pragma unroll
                for (int u = 0; u < UB; ++u) { w[u][0] = PF[12 * (size_t)(s + u)]; w[u][1] = PF[12 * (size_t)(s + u) + 1]; }
#pragma unroll
                for (int u = 0; u < UB; ++u) { acc = acc + w[u][0]; acc = acc + w[u][1]; }
            }
            for (; s < kc; ++s) { acc = acc + PF[12 * (size_t)s]; acc = acc + PF[12 * (size_t)s + 1]; }
            double pn[3];
            retract_pose(pose + 3 * i, dp + 3 * i, pn);
            pose_n[3 * i] = pn[0]; pose_n[3 * i + 1] = pn[1]; pose_n[3 * i + 2] = pn[2];
        }
        for (int a = tid; a < 2 * M; a += TPB) lm_n[a] = lm[a] + dl[a];
        newLin = block_sum<TPB>(acc, s_buf);
        __syncthreads();   // candidate values are visible to the block
        double acc2 = 0.0;   // the true cost of the candidate: block_cost with the factors' terms taken from PF
        for (int i = tid; i < N; i += TPB) {
            double e[3], pc = 0.0;   // pose_cost's own accumulator: the pose's terms are summed first, then added to the thread's
            if (i == 0) {
                prior_factor(p, pose_n, e);
                pc = pc + 0.5 * ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
            }
            if (i + 1 < N) {
                between_factor<false>(p, pose_n + 3 * i, pose_n + 3 * (i + 1), p.cmds[2 * i], p.cmds[2 * i + 1], e, nullptr);
                pc = pc + 0.5 * ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
            }
            const int kc = g.cnt[i];
            const double* PF = PFb + (size_t)i * KP * 12 + 2;
            constexpr int UB = 8;
            int s = 0;
#pragma unroll 1
            for (; s + UB <= kc; s += UB) {
                double w[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) w[u] = PF[12 * (size_t)(s + u)];
#pragma unroll
                for (int u = 0; u < UB; ++u) pc = pc + w[u];
            }
            for (; s < kc; ++s) pc = pc + PF[12 * (size_t)s];
            acc2 = acc2 + pc;
        }
        newError = block_sum<TPB>(acc2, s_buf);
    }
    if (tid == 0) {   // the decision is pgs_decide_kernel's: it needs the slots of an instance in lambda order
        p.nok[b] = ok ? 1 : 0; p.nlin[b] = newLin; p.nerr[b] = newError;
        p.solve_ok[b] = 1;
    }
}

__global__ __launch_bounds__(TPB) void pgs_decide_kernel(const PgsParams p) {
    __shared__ int s_win, s_next;
    const int b = blockIdx.x + p.b_off, tid = threadIdx.x;
    const bool running = p.state[b] == 0;
    if (p.async_ticks) {
        // asynchronous ticks: a graph whose next solve is prepared (state 4: pgs_lm_begin_kernel on the tick stream, complete before this
        // launch) joins the next trial's list; the counters the host sizes the coming grids from ride along
        if (tid == 0) {
            if (blockIdx.x == 0) { p.n_active[5] = p.mono[0]; p.n_active[6] = p.mono[1]; }
            const int stt = p.state[b];
            if (stt == 4) {
                p.state[b] = 0;
                atomicAdd(p.n_active, 1); atomicMax(p.n_active + 1, 1);
                p.alist[atomicAdd(p.n_active + 2, 1)] = b;
            } else if (stt == 3 || stt == 5 || stt == 6) atomicAdd(p.n_active, 1);   // between two solves: still counts as unfinished
        }
        if (!running) return;
    }
    if (!running && p.slots_cap <= 0) return;
    const int N = pgs_N(p, b), M = p.M[b], B = p.B;
    if (running) {
    if (tid == 0) {
        const double lambdaFactor = 10.0, lambdaUpper = 1e5, minFidelity = 1e-3, relTol = 1e-5, absTol = 1e-5;
        const int maxIter = 100;
        double lambda = p.lambda[b], error = p.error[b];
        int iters = p.iters[b], trials = p.trials[b];
        const int nl = p.nl[b] > 0 ? p.nl[b] : 1;
        int win = -1, done = 0, fl = 0;
        bool end_inner = false;
        for (int j = 0; j < nl && !end_inner; ++j) {
            const int sl = j * B + b;
            const bool ok = p.nok[sl] != 0;
            const double newLin = p.nlin[sl], newError = p.nerr[sl];
            bool success = false, stop = false;
            if (ok) {
                const double oldLin = error;
                const double linChange = oldLin - newLin;
                if (linChange >= 0.0) {
                    const double costChange = error - newError;
                    if (linChange > 2.220446049250313e-16 * oldLin) success = (costChange / linChange) > minFidelity;
                    if (fabs(costChange) < relTol * error) stop = true;
                }
            }
            trials += 1;
            if (success) {
                lambda = lambda / lambdaFactor; error = newError; iters += 1; end_inner = true; win = j;
            } else if (!stop) {
                lambda = lambda * lambdaFactor;
                if (lambda >= lambdaUpper) end_inner = true;
            } else {
                end_inner = true;
            }
        }
        if (end_inner) {   // defaultOptimize's loop condition
            const double currentError = p.cur_error[b];
            const double absDec = currentError - error, relDec = absDec / currentError;
            if (!(fabs(error) <= 1.79769313486231570e308)) { done = 1; fl = PGS_FLAG_NONFINITE; }
            else if (iters >= maxIter) { done = 1; fl = PGS_FLAG_NOT_CONVERGED; }
            else if (error <= 0.0 || relDec <= relTol || absDec <= absTol) done = 1;
            else p.cur_error[b] = error;
        }
        if (p.async_ticks && !done && trials >= p.max_trials) { done = 1; fl = PGS_FLAG_NOT_CONVERGED; }   // (lockstep: the host's trial cap + pgs_lm_end_kernel)
        atomicAdd(p.work + (p.seg_on ? 2 : (p.fused ? 1 : 0)), (double)(trials - p.trials[b]) * p.inst_flop[b]);   // reporting only
        p.lambda[b] = lambda; p.error[b] = error; p.iters[b] = iters; p.trials[b] = trials;
        // the next trial runs the next `lanes_next` lambdas of the sequence GTSAM would walk if every one of them failed:
        // lambda, 10 lambda, ... (lambda_j < lambdaUpper for j >= 1: reaching the bound ends the inner loop before that trial)
        int nnext = 1;
        if (!done) {
            double lj = lambda;
            const int want = p.lanes_next < p.lanes_max ? p.lanes_next : p.lanes_max;
            while (nnext < want) {
                lj = lj * lambdaFactor;
                if (lj >= lambdaUpper) break;
                p.lambda[nnext * B + b] = lj;
                nnext += 1;
            }
        }
        for (int j = 1; j < p.lanes_max; ++j) p.state[j * B + b] = (!done && j < nnext) ? 0 : 1;
        p.nl[b] = nnext;
        if (done) {
            p.flags[b] |= fl;
            if (p.async_ticks) { p.state[b] = 3; atomicAdd(p.n_active, 1); }   // parked until pgs_tick_kernel has advanced it (or finished it)
            else p.state[b] = 1;
        }
        else {
            atomicAdd(p.n_active, 1); atomicMax(p.n_active + 1, nnext);
            const int at = atomicAdd(p.n_active + 2, nnext);
            for (int j = 0; j < nnext; ++j) p.alist[at + j] = j * B + b;
        }
        s_win = win; s_next = done ? 0 : nnext;
    }
    __syncthreads();
    double* pose = p.pw + (size_t)b * p.N_max * 3;
    double* lm = p.lw + (size_t)b * p.L_max * 2;
    if (s_win >= 0) {   // accept the winning slot's candidate
        if (tid < p.lanes_max) p.lin_ok[(size_t)tid * B + b] = 0;   // the values change: every slot of the instance linearises anew
        const int sl = s_win * B + b;
        const double* pose_n = p.pn + (size_t)sl * p.N_max * 3;
        const double* lm_n = p.ln + (size_t)sl * p.L_max * 2;
        for (int i = tid; i < 3 * N; i += TPB) pose[i] = pose_n[i];
        for (int a = tid; a < 2 * M; a += TPB) lm[a] = lm_n[a];
    }
    // clones that run in the next trial linearise at the instance's current values (after the accept above, if any: every
    // thread re-reads the elements it wrote itself)
    for (int j = 1; j < s_next; ++j) {
        double* cp = p.pw + (size_t)(j * B + b) * p.N_max * 3;
        double* cl = p.lw + (size_t)(j * B + b) * p.L_max * 2;
        for (int i = tid; i < 3 * N; i += TPB) cp[i] = pose[i];
        for (int a = tid; a < 2 * M; a += TPB) cl[a] = lm[a];
    }
    }   // running
    if (p.slots_cap > 0) {
        // Streaming: the LAST workgroup of the launch to arrive here (every workgroup counts, also those of finished and waiting
        // graphs) refills the list: waiting graphs take the running slots this trial freed, in index order.  Which graph runs when
        // touches no result - a graph's LM sequence depends on nothing but the graph.
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            const int arrived = atomicAdd(p.n_active + 3, 1);
            if (arrived == (int)gridDim.x - 1) {
                __threadfence();
                int nslots = atomicAdd(p.n_active + 2, 0), nact = atomicAdd(p.n_active, 0);
                int w = *p.wait_next;
                const int wend = p.b_off + p.b_cnt;
                while (nslots < p.slots_cap && w < wend) {
                    p.state[w] = 0;
                    p.alist[nslots] = w;
                    nslots += 1; nact += 1; w += 1;
                }
                *p.wait_next = w;
                p.n_active[0] = nact; p.n_active[2] = nslots; p.n_active[4] = w;
                if (nact > 0) atomicMax(p.n_active + 1, 1);
            }
        }
    }
}

// result <- current values (also for instances cut off by the trial cap)
__global__ __launch_bounds__(TPB) void pgs_lm_end_kernel(const PgsParams p) {
    const int b = blockIdx.x + p.b_off, tid = threadIdx.x;
    const int N = pgs_N(p, b), M = p.M[b];
    const double* pw = p.pw + (size_t)b * p.N_max * 3;
    const double* lw = p.lw + (size_t)b * p.L_max * 2;
    double* p1 = p.pose1 + (size_t)b * p.N_max * 3;
    double* l1 = p.lm1 + (size_t)b * p.L_max * 2;
    for (int i = tid; i < 3 * N; i += TPB) p1[i] = pw[i];
    for (int i = tid; i < 2 * M; i += TPB) l1[i] = lw[i];
    if (tid == 0 && p.state[b] != 1) { p.state[b] = 1; p.flags[b] |= PGS_FLAG_NOT_CONVERGED; }   // still running or still waiting at the trial cap
}

__global__ __launch_bounds__(TPB) void pgs_adopt_kernel(const PgsParams p) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = pgs_N(p, b), M = p.M[b];
    double* p0 = p.pose0 + (size_t)b * p.N_max * 3;
    double* l0 = p.lm0 + (size_t)b * p.L_max * 2;
    const double* p1 = p.pose1 + (size_t)b * p.N_max * 3;
    const double* l1 = p.lm1 + (size_t)b * p.L_max * 2;
    for (int i = tid; i < 3 * N; i += TPB) p0[i] = p1[i];
    for (int i = tid; i < 2 * M; i += TPB) l0[i] = l1[i];
    if (tid == 0 && p.tick_acc) { p.tick_acc[2 * b] += p.iters[b]; p.tick_acc[2 * b + 1] += p.trials[b]; }
    if (tid == 0 && p.tick_flop) {
        const double n = 2.0 * M, tr = (double)p.trials[b];
        p.tick_flop[2 * b] += tr * p.inst_flop[b];
        p.tick_flop[2 * b + 1] += tr * (n * n * n / 3.0 + 2.0 * n * n);
    }
}

// Asynchronous ticks: the step between two solves of ONE graph (pose_graph.cpp:258-264, then the next timer tick's :216-256).  result <-
// current values (pgs_lm_end_kernel), initial_estimate <- result (pgs_adopt_kernel), the sums over the ticks; then - unless the graph has
// reached T_end - the graph's next simulator tick, NaiveFilter::update and the append (pgs_run_sim_kernel's body for one timestep, with the
// graph's own timestep as the noise stream's step index).  State 3 (solve converged) / 5 (first tick: nothing to adopt) -> 6, or 1 = finished.
__global__ __launch_bounds__(256) void pgs_tick_kernel(const PgsParams p) {
    constexpr int KCAP = 64;
    __shared__ float s_meas[3 * KCAP];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int st = p.state[b];
    if (st != 3 && st != 5) return;
    const int N = p.Nv[b], M = p.M[b];
    if (st == 3) {
        const double* pw = p.pw + (size_t)b * p.N_max * 3;
        const double* lw = p.lw + (size_t)b * p.L_max * 2;
        double* p0 = p.pose0 + (size_t)b * p.N_max * 3;
        double* l0 = p.lm0 + (size_t)b * p.L_max * 2;
        double* p1 = p.pose1 + (size_t)b * p.N_max * 3;
        double* l1 = p.lm1 + (size_t)b * p.L_max * 2;
        for (int i = tid; i < 3 * N; i += 256) { const double v = pw[i]; p1[i] = v; p0[i] = v; }
        for (int i = tid; i < 2 * M; i += 256) { const double v = lw[i]; l1[i] = v; l0[i] = v; }
        if (tid == 0 && p.tick_acc) { p.tick_acc[2 * b] += p.iters[b]; p.tick_acc[2 * b + 1] += p.trials[b]; }
        if (tid == 0 && p.tick_flop) {
            const double n = 2.0 * M, tr = (double)p.trials[b];
            p.tick_flop[2 * b] += tr * p.inst_flop[b];
            p.tick_flop[2 * b + 1] += tr * (n * n * n / 3.0 + 2.0 * n * n);
        }
    }
    const int i = N - 1, t1 = N;   // the graph's timestep, the pose the tick adds
    if (i >= p.T_end || t1 >= p.N_max) {
        if (tid == 0) { p.state[b] = 1; if (i < p.T_end) p.flags[b] |= PGS_FLAG_POSE_CAP; }
        return;
    }
    if (tid >= 64) return;
    const int lane = tid;
    double tx = p.truth[3 * b], ty = p.truth[3 * b + 1], tth = p.truth[3 * b + 2];
    double lmx = 0.0, lmy = 0.0;
    if (lane < p.L) { lmx = p.map[2 * lane]; lmy = p.map[2 * lane + 1]; }
    const float fwd = p.cmds[2 * i], ang = p.cmds[2 * i + 1];
    int k = sim_wave<KCAP>(p, b, lane, fwd, ang, (uint32_t)i, tx, ty, tth, lmx, lmy, s_meas);
    if (k > KCAP) { k = KCAP; if (lane == 0) p.flags[b] |= PGS_FLAG_MEAS_CAP; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane == 0) {
        double s, c;
        const double th = p.cur[3 * b + 2];
        det_sincos(th, &s, &c);
        p.cur[3 * b] = p.cur[3 * b] + (double)fwd * c;
        p.cur[3 * b + 1] = p.cur[3 * b + 1] + (double)fwd * s;
        p.cur[3 * b + 2] = remainder(th + (double)ang, kTwoPi);
        double* th_hist = p.truth_hist + ((size_t)b * p.N_max + (t1 - 1)) * 2;
        th_hist[0] = tx; th_hist[1] = ty;
        append_step(p, b, t1, s_meas, k);
        p.Nv[b] = N + 1;
        atomicMax(p.mono + 1, N + 1);
        p.state[b] = 6;
    }
}

// compute_average_error as the pose-graph plot calls it (plotting_node.py:203-213,432-434): pose i of the message
// (i < timestep, float32 on the wire) against true_poses[i] = the true pose after step i+1.
__global__ __launch_bounds__(TPB) void pgs_avg_error_kernel(const PgsParams p, int which, double* out) {
    __shared__ double s_buf[TPB];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int ts = pgs_N(p, b) - 1;
    const double* pose = (which ? p.pose1 : p.pose0) + (size_t)b * p.N_max * 3;
    const double* th = p.truth_hist + (size_t)b * p.N_max * 2;
    double acc = 0.0;
    for (int i = tid; i < ts; i += TPB) {
        const double ex = (double)(float)pose[3 * i] - th[2 * i], ey = (double)(float)pose[3 * i + 1] - th[2 * i + 1];
        acc = acc + sqrt(ex * ex + ey * ey);
    }
    const double tot = block_sum<TPB>(acc, s_buf);
    if (tid == 0) out[b] = ts > 0 ? tot / ts : 0.0;
}

#include "pgs_seg_impl.h"

}  // namespace

hipError_t pgs_launch_init(const PgsParams& p, float x0, float y0, float yaw0, hipStream_t s) {
    hipLaunchKernelGGL(pgs_init_kernel, dim3((p.B + 255) / 256), dim3(256), 0, s, p, (double)x0, (double)y0, (double)yaw0);
    return hipGetLastError();
}

hipError_t pgs_launch_append(const PgsParams& p, const float* d_meas, const int32_t* d_count, int k_stride, const double* d_sec, hipStream_t s) {
    hipLaunchKernelGGL(pgs_append_kernel, dim3((p.B + 63) / 64), dim3(64), 0, s, p, d_meas, d_count, k_stride, d_sec);
    return hipGetLastError();
}

hipError_t pgs_launch_run_sim(const PgsParams& p, int T, uint32_t step0, hipStream_t s) {
    hipLaunchKernelGGL(pgs_run_sim_kernel, dim3(p.B), dim3(64), 0, s, p, T, step0);
    return hipGetLastError();
}

hipError_t pgs_launch_seg_plan(const PgsParams& p, hipStream_t s) {
    hipLaunchKernelGGL(pgs_seg_plan_kernel, dim3(p.B), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t pgs_launch_lm_begin(const PgsParams& p, hipStream_t s) {
    hipLaunchKernelGGL(pgs_lm_begin_kernel, dim3(p.b_cnt), dim3(TPB), 0, s, p);
    return hipGetLastError();
}

hipError_t pgs_launch_trial_kernel(const PgsParams& p, int which, hipStream_t s) {
    const int nslot = pgs_nslot(p);   // slots covered: the instances of the group and their active lambda lanes, or the compacted list
    if (which == 6) {   // the decide kernel alone (split_decide): one workgroup per graph of the group, whatever the list holds
        hipLaunchKernelGGL(pgs_decide_kernel, dim3(p.b_cnt), dim3(TPB), 0, s, p);
        return hipGetLastError();
    }
    if (nslot <= 0) return hipSuccess;
    switch (which) {
    case 0:
        if (p.nfact_max > 0) hipLaunchKernelGGL(pgs_lin_factor_kernel, dim3(nslot * ((p.nfact_max + LF_TPB - 1) / LF_TPB)), dim3(LF_TPB), 0, s, p);
        hipLaunchKernelGGL(pgs_linearize_kernel, dim3(nslot), dim3(TPB), 0, s, p);
        break;
    case 1: {
        if (p.seg_on) {   // segmented elimination: the interiors of all segments side by side, then the separator chain
            const int nseg = seg_ns(p.N, p.seg_len) + 1;
            hipLaunchKernelGGL(pgs_seg_chain_kernel, dim3(nslot), dim3(64 * ((nseg + 63) / 64)), 0, s, p);   // the segments' 3x3 chains: a lane each
            hipLaunchKernelGGL(pgs_seg_kernel, dim3(nslot * nseg), dim3(SG_TPB), 0, s, p);                     // their column recurrences: a workgroup each
            if (nseg > 1) hipLaunchKernelGGL(pgs_sep_kernel, dim3(nslot), dim3(64 + p.LD), 0, s, p);
            break;
        }
        if (p.fused) {   // chain + SYRK in one launch, Y stays in LDS
            // 22 KB static + the Y buffers (89 KB at LD = 448) of the 160 KB: per device, checked (lds_attr.h)
            const void* fk = p.fused == 4 ? (const void*)pgs_chain_syrk_kernel<3, 4> : p.fused == 3 ? (const void*)pgs_chain_syrk_kernel<4, 3> : (const void*)pgs_chain_syrk_kernel<6, 2>;
            if (const hipError_t e = slam_allow_full_lds(fk); e != hipSuccess) return e;
            const size_t lds = sizeof(double) * 2 * FC_ROWS * (size_t)(p.LD + 16);
            // p.fused = workgroups per instance: the more, the fewer tiles (and MFMA time) per workgroup next to the recursion
            if (p.fused == 4) hipLaunchKernelGGL((pgs_chain_syrk_kernel<3, 4>), dim3(4 * nslot), dim3(FC_TPB), lds, s, p);
            else if (p.fused == 3) hipLaunchKernelGGL((pgs_chain_syrk_kernel<4, 3>), dim3(3 * nslot), dim3(FC_TPB), lds, s, p);
            else hipLaunchKernelGGL((pgs_chain_syrk_kernel<6, 2>), dim3(2 * nslot), dim3(FC_TPB), lds, s, p);
            break;
        }
        hipLaunchKernelGGL(pgs_chain_kernel, dim3(nslot), dim3(64 + p.LD), 0, s, p);
        break;
    }
    case 2: {
        if (p.seg_on) {   // the segments' Gram matrices side by side, then [D + lambda I; g_l^T] - Ysep^T Ysep - sum_p T_p by the tile kernel (its epilogue)
            PgsParams q = p;
            q.syrk_row0 = p.yr_sep; q.syrk_rows = 3 * seg_ns(p.N, p.seg_len); q.syrk_first = p.sep_first;
            const int nt = (p.LD + 63) / 64;
            hipLaunchKernelGGL(pgs_seg_gram_kernel, dim3(nslot * (seg_ns(p.N, p.seg_len) + 1)), dim3(GR_TPB), 0, s, p);
            hipLaunchKernelGGL(pgs_syrk_kernel<32>, dim3(8 * (nt * (nt + 1) / 2) * ((nslot + 7) / 8)), dim3(256), 0, s, q);
            break;
        }
        if (p.fused) break;   // done by the chain launch
        if (p.syrk_wave_tile == 1) {   // instance-resident accumulators
            if (const hipError_t e = slam_allow_full_lds((const void*)pgs_syrk_inst_kernel); e != hipSuccess) return e;
            const size_t lds = sizeof(double) * 2 * SI_ROWS * (size_t)(p.LD + 16);
            hipLaunchKernelGGL(pgs_syrk_inst_kernel, dim3(8 * SI_NB * ((nslot + 7) / 8)), dim3(SI_TPB), lds, s, p);
            break;
        }
        if (p.syrk_wave_tile == 64) {
            const int nt = (p.LD + 127) / 128;
            hipLaunchKernelGGL(pgs_syrk_kernel<64>, dim3(8 * (nt * (nt + 1) / 2) * ((nslot + 7) / 8)), dim3(256), 0, s, p);
        } else {
            const int nt = (p.LD + 63) / 64;
            hipLaunchKernelGGL(pgs_syrk_kernel<32>, dim3(8 * (nt * (nt + 1) / 2) * ((nslot + 7) / 8)), dim3(256), 0, s, p);
        }
        break;
    }
    case 3: {
        const size_t lds = sizeof(double) * (size_t)(p.LD + 16) * 17;   // panel rows x (NB + 1)
        // panels of L_max > 235 need more than the default 64 KiB of dynamic LDS (gfx950: 160 KiB); solve groups launch from several
        // host threads and a single-process host may hold several devices: per (kernel, device), checked (lds_attr.h)
        const void* ck = p.chol_threads == 256 ? (const void*)pgs_chol_kernel<256> : (p.chol_ll == 1 && p.LD <= 448) ? (const void*)pgs_chol_ll_kernel<1024>
                       : (p.chol_ll && p.LD <= 448) ? (const void*)pgs_chol_ll_kernel<768> : (const void*)pgs_chol_kernel<1024>;
        if (const hipError_t e = slam_allow_full_lds(ck); e != hipSuccess) return e;
        if (p.chol_threads == 256) { hipLaunchKernelGGL(pgs_chol_kernel<256>, dim3(nslot), dim3(256), lds, s, p); break; }
        if (p.chol_ll && p.LD <= 448) {   // left-looking: two consecutive panels [<= LD + 1][17] each, the staged block rows over the older one (its staging registers are sized for LD <= 448)
            const size_t lds_ll = sizeof(double) * 2 * (size_t)(p.LD + 1) * 17;
            if (p.chol_ll == 1) hipLaunchKernelGGL(pgs_chol_ll_kernel<1024>, dim3(nslot), dim3(1024), lds_ll, s, p);
            else hipLaunchKernelGGL(pgs_chol_ll_kernel<768>, dim3(nslot), dim3(768), lds_ll, s, p);
            break;
        }
        hipLaunchKernelGGL(pgs_chol_kernel<1024>, dim3(nslot), dim3(1024), lds, s, p);
        break;
    }
    case 4:
        if (p.seg_on && p.N <= kSegBackLdsPoses && !p.seg_back_global) {   // the chains out of LDS (12 doubles per pose)
            if (const hipError_t e = slam_allow_full_lds((const void*)pgs_seg_backsolve_lds_kernel); e != hipSuccess) return e;
            hipLaunchKernelGGL(pgs_seg_backsolve_lds_kernel, dim3(nslot), dim3(SBL_TPB), sizeof(double) * 12 * (size_t)p.N, s, p);
        } else if (p.seg_on) hipLaunchKernelGGL(pgs_seg_backsolve_kernel, dim3(nslot), dim3(SB_TPB), 0, s, p);
        else hipLaunchKernelGGL(pgs_backsolve_kernel, dim3(nslot), dim3(BTPB), 0, s, p);
        break;
    default:   // the candidates of every slot, then GTSAM's accept / lambda / convergence logic per instance
        if (p.nfact_max > 0) hipLaunchKernelGGL(pgs_eval_factor_kernel, dim3(nslot * ((p.nfact_max + LF_TPB - 1) / LF_TPB)), dim3(LF_TPB), 0, s, p);
        hipLaunchKernelGGL(pgs_evaluate_kernel, dim3(nslot), dim3(TPB), 0, s, p);
        if (!p.split_decide) hipLaunchKernelGGL(pgs_decide_kernel, dim3(p.b_cnt), dim3(TPB), 0, s, p);
        break;
    }
    return hipGetLastError();
}

hipError_t pgs_launch_tick(const PgsParams& p, hipStream_t s) {
    hipLaunchKernelGGL(pgs_tick_kernel, dim3(p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(pgs_seg_plan_kernel, dim3(p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(pgs_lm_begin_kernel, dim3(p.B), dim3(TPB), 0, s, p);
    return hipGetLastError();
}

hipError_t pgs_launch_lm_end(const PgsParams& p, hipStream_t s) {
    hipLaunchKernelGGL(pgs_lm_end_kernel, dim3(p.b_cnt), dim3(TPB), 0, s, p);
    return hipGetLastError();
}

hipError_t pgs_launch_adopt(const PgsParams& p, hipStream_t s) {
    hipLaunchKernelGGL(pgs_adopt_kernel, dim3(p.B), dim3(TPB), 0, s, p);
    return hipGetLastError();
}

hipError_t pgs_launch_avg_error(const PgsParams& p, int which, double* out, hipStream_t s) {
    hipLaunchKernelGGL(pgs_avg_error_kernel, dim3(p.B), dim3(TPB), 0, s, p, which, out);
    return hipGetLastError();
}

}  // namespace slam
