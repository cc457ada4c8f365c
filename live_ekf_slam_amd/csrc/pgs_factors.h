// pgs_factors.h — the factors (whitened residuals / Jacobians of pose_graph.cpp's Prior / Between / BearingRange factors), per-instance views, deterministic block sums, the cost.
// Part of pgs_kernel.hip (round 6: split by phase, pure moves); included there inside namespace slam { namespace {.  DESIGN.md 4.4.
#pragma once

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
