// pgs_graph.h — graph building: PoseGraph::init / updateNaiveVehPoseEstimate / update / onLandmarkMeasurement (pose_graph.cpp:68-256) and the device-side simulator loop.
// Part of pgs_kernel.hip (round 6: split by phase, pure moves); included there inside namespace slam { namespace {.  DESIGN.md 4.4.
#pragma once

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
    // Every detection of the message reaches append_step (a map has at most 255 landmarks): a detection beyond the k_per_pose factor slots of its
    // pose is dropped THERE, after its landmark was created - like the reference's loop (pose_graph.cpp:249-256) and the oracle.  Until round 6 the
    // message was cut at 64 detections before that (tools/gpu_soak_pgs.py wide: landmarks created at another pose, metres apart).
    constexpr int KCAP = 256;
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
