// pgs_backsolve.h — the pose step through the sequential chain's factor (affine scans).
// Part of pgs_kernel.hip (round 6: split by phase, pure moves); included there inside namespace slam { namespace {.  DESIGN.md 4.4.
#pragma once

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
