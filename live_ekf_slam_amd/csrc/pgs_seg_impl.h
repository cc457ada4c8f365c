// pgs_seg_impl.h — the pose chain eliminated SEGMENT BY SEGMENT (round 5).  Included by pgs_kernel.hip inside namespace slam { namespace {.
//
// Reference: the linear solve inside gtsam::LevenbergMarquardtOptimizer::tryLambda, called from PoseGraph::solvePoseGraph
// (ekf_ws/src/localization_pkg/src/pose_graph.cpp:273-300).  GTSAM eliminates in a COLAMD order; any exact elimination order gives the
// same step up to rounding (oracle/slam_oracle_pgs.cpp: LIN_SCHUR == LIN_DENSE == LIN_SEG).  Rounds 1-4 eliminated the poses in
// trajectory order 0, 1, 2, ...: a block-bidiagonal factor, ONE dependent 3x3 recursion of N = 1000 links per trial (0.56 ms whatever
// the batch: 46 of the 90 ms of a batch-256 solve), and a Y = L^-1 [H_pl | g_p] whose rows are dense from a landmark's first detection
// on (3N x (2M+1): 175 MFLOP of Schur-complement SYRK per instance-trial at 1000 x 170).
//
// Here the chain is cut at the SEPARATOR poses SL, 2 SL, ... (nested dissection with one level; the separators of a chain are single
// poses).  Order: the interiors of all segments (independent of each other: one workgroup per (slot, segment), pgs_seg_kernel), then the
// NS = (N - 2) / SL separators as a short chain (pgs_sep_kernel), then the landmarks (the dense Cholesky, unchanged).
//   * depth: SL + NS (63 at N = 1000, SL = 32) dependent 3x3 steps instead of N;
//   * fill: an interior row of Y only has the columns of the landmarks its own segment sees (<= 49 of 170 at SL = 32 on BASELINE
//     configs[4]), only the 3 NS separator rows are dense: 8 MFLOP of SYRK per instance-trial instead of 175, a seventh of the
//     column-recurrence work; the price is the SPIKE - every interior pose also couples to its segment's left separator (one more 3x3
//     block per pose and column step).
// Block algebra (oracle: Pgs::solve_seg, statement for statement).  Segment p, left separator a = p SL (p >= 1), right separator
// b = (p + 1) SL (p < NS), interior poses i = lo .. hi - 1:
//   Ginn_i = H[i][i-1] L_{i-1}^-T (zero at i = lo),  L_i = chol(A_i + lambda I - Ginn_i Ginn_i^T),
//   spike:  B_lo = H[a][lo] = C_a^T,  B_i = -(Gs_{i-1} Ginn_i^T),  Gs_i = B_i L_i^-T,   separator a:  A_a -= sum_i Gs_i Gs_i^T,
//   right end: Gr = C_{hi-1} L_{hi-1}^-T,  A_b -= Gr Gr^T,  H[b][a] = -(Gr Gs_{hi-1}^T);
//   Y_i = L_i^-1 ([E_i | g_i] - Ginn_i Y_{i-1}) on the segment's columns,  R_a -= sum_i Gs_i Y_i,  R_b -= Gr Y_{hi-1};
//   separators k = 1 .. NS: the same chain recurrence on (A_s - ..., H[s_k][s_{k-1}], R_s) with ALL columns;
//   S_ext = [D + lambda I; g_l^T] - Ysep^T Ysep - sum_p Y_p^T Y_p  (tile kernel on the 3 NS dense rows, then the segments in order).
// Everything an instance's result depends on is accumulated in a fixed order (no atomics on data): run-to-run and shard-invariant.
#pragma once

// (seg_ns / seg_lo / seg_hi: the geometry of the segments, pgs_kernel.h)

// inverse of the Cholesky factor of the symmetric 3x3 (T0; T3 T4; T6 T7 T8): I = i00 i10 i11 i20 i21 i22.  Pivots through v_rsq_f64 +
// two Newton steps like the sequential chain (rsqrt_nr); false if a pivot is not positive.
__device__ __forceinline__ bool chol_inv3(double T0, double T3, double T4, double T6, double T7, double T8, double I[6]) {
    if (!(T0 > 0.0)) return false;
    I[0] = rsqrt_nr(T0);
    const double l10 = T3 * I[0], l20 = T6 * I[0];
    const double t11 = T4 - l10 * l10;
    if (!(t11 > 0.0)) return false;
    I[2] = rsqrt_nr(t11);
    const double l21 = (T7 - l20 * l10) * I[2];
    const double t22 = (T8 - l20 * l20) - l21 * l21;
    if (!(t22 > 0.0)) return false;
    I[5] = rsqrt_nr(t22);
    I[1] = -(l10 * I[0]) * I[2];
    I[4] = -(l21 * I[2]) * I[5];
    I[3] = -(l20 * I[0] + l21 * I[1]) * I[5];
    return true;
}
__device__ __forceinline__ void mul_linvT(const double* X, const double* I, double* G) {   // G = X Linv^T
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        G[3 * r + 0] = X[3 * r] * I[0];
        G[3 * r + 1] = X[3 * r] * I[1] + X[3 * r + 1] * I[2];
        G[3 * r + 2] = (X[3 * r] * I[3] + X[3 * r + 1] * I[4]) + X[3 * r + 2] * I[5];
    }
}
__device__ __forceinline__ void mul_abT(const double* X, const double* Z, double* P) {   // P = X Z^T
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) P[3 * a + c] = (X[3 * a] * Z[3 * c] + X[3 * a + 1] * Z[3 * c + 1]) + X[3 * a + 2] * Z[3 * c + 2];
}
__device__ __forceinline__ void sym_ggT(const double* G, double* o) {   // lower triangle 00 10 11 20 21 22 of G G^T
    o[0] = (G[0] * G[0] + G[1] * G[1]) + G[2] * G[2];
    o[1] = (G[3] * G[0] + G[4] * G[1]) + G[5] * G[2];
    o[2] = (G[3] * G[3] + G[4] * G[4]) + G[5] * G[5];
    o[3] = (G[6] * G[0] + G[7] * G[1]) + G[8] * G[2];
    o[4] = (G[6] * G[3] + G[7] * G[4]) + G[8] * G[5];
    o[5] = (G[6] * G[6] + G[7] * G[7]) + G[8] * G[8];
}
#define SEG_SUB_GV(G, v0, v1, v2, u0, u1, u2)                          \
    do {                                                               \
        u0 -= ((G)[0] * (v0) + (G)[1] * (v1)) + (G)[2] * (v2);         \
        u1 -= ((G)[3] * (v0) + (G)[4] * (v1)) + (G)[5] * (v2);         \
        u2 -= ((G)[6] * (v0) + (G)[7] * (v1)) + (G)[8] * (v2);         \
    } while (0)
#define SEG_SUB_GTV(G, v0, v1, v2, u0, u1, u2)                         \
    do {                                                               \
        u0 -= ((G)[0] * (v0) + (G)[3] * (v1)) + (G)[6] * (v2);         \
        u1 -= ((G)[1] * (v0) + (G)[4] * (v1)) + (G)[7] * (v2);         \
        u2 -= ((G)[2] * (v0) + (G)[5] * (v1)) + (G)[8] * (v2);         \
    } while (0)

// ---- the plan: which landmarks each segment's interior poses see.  From the graph alone (cnt, mlm), once per solve and instance; the
//      lambda lanes get copies (clone_instances).  Also the first separator row that can be non-zero per landmark (k range of the dense
//      tile SYRK) and the largest column set (the host takes the sequential path if a segment sees more than kPgsSegMaxLm landmarks). ----
__global__ __launch_bounds__(256) void pgs_seg_plan_kernel(const PgsParams p) {
    __shared__ int s_has[256], s_first[256];   // L_max <= 255
    __shared__ int s_nf;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (p.async_ticks && p.state[b] != 6) return;   // asynchronous ticks: only the graphs whose next tick was just appended
    const int N = pgs_N(p, b), KP = p.KP, M = p.M[b], L_max = p.L_max;
    const int32_t* cnt = p.cnt + (size_t)b * p.N_max;
    const int32_t* mlm = p.mlm + (size_t)b * p.N_max * KP;
    {   // factors of the instance (the grid of the per-factor kernels)
        if (tid == 0) s_nf = 0;
        __syncthreads();
        int nf = 0;
        for (int i = tid; i < N; i += 256) nf += cnt[i];
        atomicAdd(&s_nf, nf);
        __syncthreads();
        if (tid == 0) { p.fact_cnt[b] = s_nf; if (p.mono) atomicMax(p.mono, s_nf); }
    }
    if (p.seg_len <= 0) return;
    const int SL = p.seg_len, NS = seg_ns(N, SL), nseg = NS + 1;
    if (NS > kPgsSegMaxSep) {   // (the host takes the sequential path; asynchronous ticks: the graph is finished and flagged)
        if (tid == 0) { p.seg_umax[b] = 0x7fffffff; if (p.async_ticks) { p.flags[b] |= PGS_FLAG_SEG_LIMIT; p.state[b] = 1; } }
        return;
    }
    int32_t* ncol = p.seg_ncol + (size_t)b * p.nseg_max;
    int32_t* slm = p.seg_lm + (size_t)b * p.nseg_max * L_max;
    int32_t* sinv = p.seg_inv + (size_t)b * p.nseg_max * L_max;
    if (tid < L_max) s_first[tid] = NS;
    int umax = 0;
    for (int ps = 0; ps < nseg; ++ps) {
        if (tid < L_max) s_has[tid] = 0;
        __syncthreads();
        const int lo = seg_lo(ps, SL), hi = seg_hi(ps, SL, NS, N);
        for (int idx = tid; idx < (hi - lo) * KP; idx += 256) {
            const int i = lo + idx / KP, s = idx - (idx / KP) * KP;
            if (s < cnt[i]) s_has[mlm[(size_t)i * KP + s] & (kPgsFirstBit - 1)] = 1;   // (every writer stores the same value)
        }
        __syncthreads();
        if (tid < 64) {   // compaction in landmark order, 64 landmarks per ballot
            int base = 0;
            int32_t* blk = p.seg_blk + ((size_t)b * p.nseg_max + ps) * seg_nb1(L_max);
            for (int j0 = 0; j0 < L_max; j0 += 64) {
                const int j = j0 + lane;
                const bool has = j < M && s_has[j < L_max ? j : 0] != 0;
                const unsigned long long m = __ballot(has);
                const int lc = base + __popcll(m & ((1ull << lane) - 1ull));
                if (lane < 4) blk[(j0 >> 4) + lane] = base + __popcll(m & ((1ull << (16 * lane)) - 1ull));   // local landmarks below 16 * block
                if (j < L_max) sinv[(size_t)ps * L_max + j] = has ? lc : -1;
                if (has) {
                    slm[(size_t)ps * L_max + lc] = j;
                    // segment ps feeds separator ps (its left one, 0-based row ps - 1) and separator ps + 1 (0-based row ps)
                    const int fs = ps >= 1 ? ps - 1 : 0;
                    if (fs < s_first[j]) s_first[j] = fs;
                }
                base += __popcll(m);
            }
            if (lane == 0) { blk[seg_nb1(L_max) - 2] = base; blk[seg_nb1(L_max) - 1] = base; ncol[ps] = base; }
            umax = base > umax ? base : umax;
        }
        __syncthreads();
    }
    for (int idx = tid; idx < NS * KP; idx += 256) {   // detections AT the separator poses
        const int k = idx / KP, s = idx - k * KP, i = (k + 1) * SL;
        if (s < cnt[i]) atomicMin(&s_first[mlm[(size_t)i * KP + s] & (kPgsFirstBit - 1)], k);
    }
    __syncthreads();
    // The tile SYRK takes the k range of a 32-row wavefront tile from the tile's FIRST row, i.e. it needs first[] non-decreasing in the
    // landmark index.  Landmarks are numbered by first detection, but a detection beyond the KP factor slots of its pose is dropped
    // (PGS_FLAG_MEAS_CAP) while its landmark is created: such a landmark's first FACTOR can come long after those of the landmarks numbered
    // after it.  So: the suffix minimum (found by tools/gpu_soak_pgs.py on KP = 4 graphs: steps wrong by metres).
    if (tid == 0) {
        int run = NS;
        for (int j = L_max - 1; j >= 0; --j) { run = s_first[j] < run ? s_first[j] : run; s_first[j] = run; }
    }
    __syncthreads();
    if (tid < L_max) p.sep_first[(size_t)b * L_max + tid] = s_first[tid];
    if (tid == 0) {
        p.seg_umax[b] = umax;
        // asynchronous ticks: nobody on the host looks at the plan before the solve runs - a graph the segmented elimination cannot hold
        // (kPgsSegMaxLm landmarks per segment) is finished here with its last adopted result and flagged
        if (p.async_ticks && umax > kPgsSegMaxLm) { p.flags[b] |= PGS_FLAG_SEG_LIMIT; p.state[b] = 1; }
    }
}

// ---- interiors of the segments, part 1: the 3x3 chains with their spikes.  ONE workgroup per slot, one LANE per segment: the chains of a
//      slot's segments are independent and run the same instruction stream, so a wavefront carries 64 of them in lockstep for the price of
//      one (a workgroup per segment with one busy lane each: 8 192 workgroups at a full batch, 310 us against 36 for the chain itself).  Every
//      lane fetches its next pose's blocks a step ahead, leaves the factor (Linv, Ginn, Gs) of its poses in global memory for part 2 and the
//      pose step, and the segment's contributions to its two separators in segout. ----
__global__ __launch_bounds__(64 * ((kPgsSegMaxSep + 1 + 63) / 64)) void pgs_seg_chain_kernel(const PgsParams p) {
    __shared__ int s_fail;
    const int b = pgs_slot(p, blockIdx.x), ps = threadIdx.x;
    if (p.state[b]) return;
    const int SL = p.seg_len, N = pgs_N(p, b), NS = seg_ns(N, SL), nseg = NS + 1;
    if (ps == 0) s_fail = 0;
    __syncthreads();
    if (ps < nseg) {
        const int lo = seg_lo(ps, SL), hi = seg_hi(ps, SL, NS, N), len = hi - lo;
        const double lambda = p.lambda[b];
        const double* Ab = p.A + (size_t)b * p.N_max * 9;
        const double* Cb = p.C + (size_t)b * p.N_max * 9;
        double* Lb = p.Linv + (size_t)b * p.N_max * 6;
        double* Gb = p.G + (size_t)b * p.N_max * 9;
        double* Gsb = p.Gs + (size_t)b * p.N_max * 9;
        auto fetch = [&](int i, double (&in)[15]) {   // A (00 10 11 20 21 22), H[i][i-1] (zero block before pose 0)
            const double* A = Ab + 9 * (size_t)i;
            in[0] = A[0]; in[1] = A[3]; in[2] = A[4]; in[3] = A[6]; in[4] = A[7]; in[5] = A[8];
            const double* C = Cb + 9 * (size_t)(i > 0 ? i - 1 : 0);
#pragma unroll
            for (int k = 0; k < 9; ++k) in[6 + k] = i > 0 ? C[k] : 0.0;
        };
        double in[15], nx[15];
        fetch(lo, in);
        double I[6] = {0, 0, 0, 0, 0, 0}, Gsp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Bc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, aL[6] = {0, 0, 0, 0, 0, 0};
        if (ps >= 1) {   // H[a][lo] = C_a^T (for the segment's first pose in[6..14] is C_a, the coupling to the left separator)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Bc[3 * r + c] = in[6 + 3 * c + r];
        }
        bool ok = true;
#pragma unroll 1
        for (int l = 0; l < len; ++l) {
            const int i = lo + l;
            fetch(l + 1 < len ? i + 1 : i, nx);   // the next pose's blocks travel under this pose's chain
            double G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (l > 0) mul_linvT(in + 6, I, G);
            const double T0 = (in[0] + lambda) - ((G[0] * G[0] + G[1] * G[1]) + G[2] * G[2]);
            const double T3 = in[1] - ((G[3] * G[0] + G[4] * G[1]) + G[5] * G[2]);
            const double T4 = (in[2] + lambda) - ((G[3] * G[3] + G[4] * G[4]) + G[5] * G[5]);
            const double T6 = in[3] - ((G[6] * G[0] + G[7] * G[1]) + G[8] * G[2]);
            const double T7 = in[4] - ((G[6] * G[3] + G[7] * G[4]) + G[8] * G[5]);
            const double T8 = (in[5] + lambda) - ((G[6] * G[6] + G[7] * G[7]) + G[8] * G[8]);
            if (ps >= 1 && l > 0) {   // fill of eliminating the previous pose
                double P[9];
                mul_abT(Gsp, G, P);
#pragma unroll
                for (int k = 0; k < 9; ++k) Bc[k] = -P[k];
            }
            if (!chol_inv3(T0, T3, T4, T6, T7, T8, I)) { ok = false; break; }
            if (ps >= 1) {
                mul_linvT(Bc, I, Gsp);
                double q[6];
                sym_ggT(Gsp, q);
#pragma unroll
                for (int k = 0; k < 6; ++k) aL[k] += q[k];
            }
            double* L = Lb + 6 * (size_t)i;
            double* Go = Gb + 9 * (size_t)i;
            double* Gso = Gsb + 9 * (size_t)i;
#pragma unroll
            for (int k = 0; k < 6; ++k) L[k] = I[k];
#pragma unroll
            for (int k = 0; k < 9; ++k) { Go[k] = G[k]; Gso[k] = Gsp[k]; }
#pragma unroll
            for (int k = 0; k < 15; ++k) in[k] = nx[k];
        }
        double* so = p.segout + ((size_t)b * p.nseg_max + ps) * 32;
        if (ok) {
            double Gr[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, aR[6] = {0, 0, 0, 0, 0, 0}, Hba[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (ps < NS) {   // the right separator couples to the last interior pose through H[hi][hi-1]
                double ce[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) ce[k] = Cb[9 * (size_t)(hi - 1) + k];
                mul_linvT(ce, I, Gr);
                sym_ggT(Gr, aR);
                if (ps >= 1) mul_abT(Gr, Gsp, Hba);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) { so[k] = aL[k]; so[6 + k] = aR[k]; }
#pragma unroll
            for (int k = 0; k < 9; ++k) { so[12 + k] = Gr[k]; so[21 + k] = Hba[k]; }
        } else {
            s_fail = 1;   // (every failing lane stores the same value)
        }
    }
    __syncthreads();
    if (ps == 0 && s_fail) p.solve_ok[b] = 0;
}

// ---- interiors of the segments, part 2: the column recurrence, one workgroup per (slot, segment), one thread per local column ----
constexpr int SG_TPB = 128;   // >= 2 * kPgsSegMaxLm + 1 columns, one per thread
__global__ __launch_bounds__(SG_TPB) void pgs_seg_kernel(const PgsParams p) {
    // (a segment holds at most SL poses - except the ONLY segment of a graph of exactly SL + 1 poses, which has no separator yet: SL + 1;
    // found by tools/gpu_soak_pgs.py on 33-pose graphs)
    __shared__ double s_fac[kPgsSegMaxLen + 1][28];   // Linv (6), Ginn (9), Gs (9), g_p (3)
    __shared__ double s_gr[9];
    const int SL = p.seg_len, nsegl = seg_ns(p.N, SL) + 1;   // segments per slot of the LAUNCH (p.N: the most poses any graph has)
    const int bl = blockIdx.x / nsegl, ps = blockIdx.x - bl * nsegl;
    const int b = pgs_slot(p, bl), tid = threadIdx.x;
    if (p.state[b] || !p.solve_ok[b]) return;
    const int N = pgs_N(p, b), NS = seg_ns(N, SL), nseg = NS + 1;
    if (ps >= nseg) return;
    const int lo = seg_lo(ps, SL), hi = seg_hi(ps, SL, NS, N), len = hi - lo, LD = p.LD;
    {   // the segment's factor (written by pgs_seg_chain_kernel: L2) and gradient blocks into LDS, coalesced
        const double* Lb = p.Linv + (size_t)b * p.N_max * 6 + 6 * (size_t)lo;
        const double* Gb = p.G + (size_t)b * p.N_max * 9 + 9 * (size_t)lo;
        const double* Gsb = p.Gs + (size_t)b * p.N_max * 9 + 9 * (size_t)lo;
        const double* gpb = p.gp + (size_t)b * p.N_max * 3 + 3 * (size_t)lo;
        for (int e = tid; e < 6 * len; e += SG_TPB) s_fac[e / 6][e % 6] = Lb[e];
        for (int e = tid; e < 9 * len; e += SG_TPB) { s_fac[e / 9][6 + e % 9] = Gb[e]; s_fac[e / 9][15 + e % 9] = Gsb[e]; }
        for (int e = tid; e < 3 * len; e += SG_TPB) s_fac[e / 3][24 + e % 3] = gpb[e];
        if (tid < 9) s_gr[tid] = p.segout[((size_t)b * p.nseg_max + ps) * 32 + 12 + tid];
    }
    __syncthreads();
    // ---- columns: local column lc = tid; 2 ncol landmark columns, then the gradient column ----
    const int ncol = p.seg_ncol[(size_t)b * p.nseg_max + ps], nc = 2 * ncol + 1;
    if (tid >= nc) return;
    const bool grad = tid == nc - 1;
    const int myd = tid & 1;
    int cur = 0, end = 0, next_i = 0x7fffffff;
    const double* Elmb = p.Elm + (size_t)b * p.N_max * p.KP * 6;
    const int32_t* evt_pose = p.evt_pose + (size_t)b * p.N_max * p.KP;
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;
    if (!grad) {
        const size_t li = ((size_t)b * p.nseg_max + ps) * p.L_max + (tid >> 1);
        const int j = p.seg_lm[li];
        cur = p.seg_evt[li];
        end = p.evt_start[(size_t)b * (p.L_max + 1) + j + 1];
        if (cur < end) {
            next_i = evt_pose[cur];
            e0 = Elmb[6 * (size_t)cur + myd]; e1 = Elmb[6 * (size_t)cur + 2 + myd]; e2 = Elmb[6 * (size_t)cur + 4 + myd];
        }
    }
    double* Yb = p.Y + (size_t)b * p.y_stride;
    double* Yi = Yb + (size_t)3 * lo * LD + tid;
    double y0 = 0.0, y1 = 0.0, y2 = 0.0, r0 = 0.0, r1 = 0.0, r2 = 0.0;
#pragma unroll 2
    for (int l = 0; l < len; ++l) {
        const int i = lo + l;
        const double* o = s_fac[l];
        double u0 = 0.0, u1 = 0.0, u2 = 0.0;
        if (grad) { u0 = o[24]; u1 = o[25]; u2 = o[26]; }
        SEG_SUB_GV(o + 6, y0, y1, y2, u0, u1, u2);   // Ginn is zero at the segment's first pose
        while (i == next_i) {   // (a message may hold the same landmark twice: two factors at one pose)
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
        Yi[0] = y0; Yi[LD] = y1; Yi[2 * (size_t)LD] = y2;
        Yi += 3 * (size_t)LD;
        // R_a -= Gs_i Y_i, kept with a plus sign (Gs is zero in segment 0)
        r0 += (o[15] * y0 + o[16] * y1) + o[17] * y2;
        r1 += (o[18] * y0 + o[19] * y1) + o[20] * y2;
        r2 += (o[21] * y0 + o[22] * y1) + o[23] * y2;
    }
    double* Rc = Yb + (size_t)(p.yr_rc + 6 * ps) * LD + tid;
    Rc[0] = r0; Rc[LD] = r1; Rc[2 * (size_t)LD] = r2;
    Rc[3 * (size_t)LD] = (s_gr[0] * y0 + s_gr[1] * y1) + s_gr[2] * y2;   // Gr is zero in the last segment
    Rc[4 * (size_t)LD] = (s_gr[3] * y0 + s_gr[4] * y1) + s_gr[5] * y2;
    Rc[5 * (size_t)LD] = (s_gr[6] * y0 + s_gr[7] * y1) + s_gr[8] * y2;
}

// ---- the separators: a chain of NS poses over ALL columns (threads 64 .. 64 + LD - 1 own one global column each) ----
__global__ __launch_bounds__(1024) void pgs_sep_kernel(const PgsParams p) {
    __shared__ double s_sin[kPgsSegMaxSep][30];   // A (6), Gr Gr^T of the segment before (6), sum Gs Gs^T of the segment after (6), Gr Gs_e^T (9), g_p (3)
    __shared__ double s_sf[kPgsSegMaxSep][16];    // Linv (6), G (9)
    __shared__ int s_fail;
    const int b = pgs_slot(p, blockIdx.x), tid = threadIdx.x;
    if (p.state[b] || !p.solve_ok[b]) return;
    const int SL = p.seg_len, N = pgs_N(p, b), NS = seg_ns(N, SL), LD = p.LD, KP = p.KP, m2 = 2 * p.M[b];
    if (NS == 0) return;
    const double lambda = p.lambda[b];
    const double* Ab = p.A + (size_t)b * p.N_max * 9;
    const double* gpb = p.gp + (size_t)b * p.N_max * 3;
    const double* sob = p.segout + (size_t)b * p.nseg_max * 32;
    for (int k = tid; k < NS; k += blockDim.x) {   // separator k + 1 (0-based k), pose s
        const int s = (k + 1) * SL;
        const double* A = Ab + 9 * (size_t)s;
        double* o = s_sin[k];
        o[0] = A[0]; o[1] = A[3]; o[2] = A[4]; o[3] = A[6]; o[4] = A[7]; o[5] = A[8];
#pragma unroll
        for (int q = 0; q < 6; ++q) { o[6 + q] = sob[(size_t)k * 32 + 6 + q]; o[12 + q] = sob[(size_t)(k + 1) * 32 + q]; }
#pragma unroll
        for (int q = 0; q < 9; ++q) o[18 + q] = sob[(size_t)k * 32 + 21 + q];
        o[27] = gpb[3 * s]; o[28] = gpb[3 * s + 1]; o[29] = gpb[3 * s + 2];
    }
    if (tid == 0) s_fail = 0;
    __syncthreads();
    const int c = tid - 64;
    double* Yb = p.Y + (size_t)b * p.y_stride;
    double* Ys = Yb + (size_t)p.yr_sep * LD + c;
    if (tid == 0) {
        double I[6] = {0, 0, 0, 0, 0, 0};
        bool ok = true;
#pragma unroll 1
        for (int k = 0; k < NS && ok; ++k) {
            const double* in = s_sin[k];
            double G[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (k >= 1) {   // H[s_k][s_{k-1}] = -(Gr Gs_e^T) of the segment between them
                double Hk[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) Hk[q] = -in[18 + q];
                mul_linvT(Hk, I, G);
            }
            const double T0 = (((in[0] + lambda) - in[6]) - in[12]) - ((G[0] * G[0] + G[1] * G[1]) + G[2] * G[2]);
            const double T3 = ((in[1] - in[7]) - in[13]) - ((G[3] * G[0] + G[4] * G[1]) + G[5] * G[2]);
            const double T4 = (((in[2] + lambda) - in[8]) - in[14]) - ((G[3] * G[3] + G[4] * G[4]) + G[5] * G[5]);
            const double T6 = ((in[3] - in[9]) - in[15]) - ((G[6] * G[0] + G[7] * G[1]) + G[8] * G[2]);
            const double T7 = ((in[4] - in[10]) - in[16]) - ((G[6] * G[3] + G[7] * G[4]) + G[8] * G[5]);
            const double T8 = (((in[5] + lambda) - in[11]) - in[17]) - ((G[6] * G[6] + G[7] * G[7]) + G[8] * G[8]);
            if (!chol_inv3(T0, T3, T4, T6, T7, T8, I)) { ok = false; break; }
            double* o = s_sf[k];
#pragma unroll
            for (int q = 0; q < 6; ++q) o[q] = I[q];
#pragma unroll
            for (int q = 0; q < 9; ++q) o[6 + q] = G[q];
        }
        if (!ok) s_fail = 1;
    } else if (c >= 0 && c <= m2 && c < LD) {
        // pass 1, BESIDE lane 0's chain (it needs nothing of it; no dependence between the separators either, so the loads of several
        // of them are in flight at once): the right-hand side before the chain term, (g - right contribution of segment k) - left
        // contribution of segment k + 1, + E at the separator's pose (the landmark's event there: sep_evt)
        const bool grad = c == m2;
        const int j = c >> 1, d = c & 1;
        const int32_t* ncolb = p.seg_ncol + (size_t)b * p.nseg_max;
        const int32_t* sinv = p.seg_inv + (size_t)b * p.nseg_max * p.L_max;
        const int32_t* sevt = p.sep_evt + (size_t)b * p.nseg_max * p.L_max;
        const int32_t* evt_pose = p.evt_pose + (size_t)b * p.N_max * KP;
        const int evt_end = grad ? 0 : p.evt_start[(size_t)b * (p.L_max + 1) + j + 1];
        const double* Elmb = p.Elm + (size_t)b * p.N_max * KP * 6;
        const double* Rc = Yb + (size_t)p.yr_rc * LD;
        constexpr int PB = 4;   // separators per batch: index loads, then the loads they address, then the arithmetic
#pragma unroll 1
        for (int k0 = 0; k0 < NS; k0 += PB) {
            int lr[PB], ll[PB], ev[PB];
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const int k = k0 + q < NS ? k0 + q : NS - 1;
                if (grad) { lr[q] = 2 * ncolb[k]; ll[q] = 2 * ncolb[k + 1]; ev[q] = -1; }
                else {
                    const int ir = sinv[(size_t)k * p.L_max + j], il = sinv[(size_t)(k + 1) * p.L_max + j];
                    lr[q] = ir >= 0 ? 2 * ir + d : -1; ll[q] = il >= 0 ? 2 * il + d : -1;
                    ev[q] = sevt[(size_t)k * p.L_max + j];
                }
            }
            double rr[PB][3], rl[PB][3], ee[PB][3];
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const int k = k0 + q < NS ? k0 + q : NS - 1;
                const double* qr = Rc + (size_t)(6 * k + 3) * LD + (lr[q] >= 0 ? lr[q] : 0);
                const double* ql = Rc + (size_t)(6 * (k + 1)) * LD + (ll[q] >= 0 ? ll[q] : 0);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    rr[q][r] = lr[q] >= 0 ? qr[(size_t)r * LD] : 0.0;
                    rl[q][r] = ll[q] >= 0 ? ql[(size_t)r * LD] : 0.0;
                    ee[q][r] = ev[q] >= 0 ? Elmb[6 * (size_t)ev[q] + 2 * r + d] : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const int k = k0 + q;
                if (k >= NS) break;
                const int s = (k + 1) * SL;
                double u0 = 0.0, u1 = 0.0, u2 = 0.0;
                if (grad) { u0 = s_sin[k][27]; u1 = s_sin[k][28]; u2 = s_sin[k][29]; }
                u0 = (u0 - rr[q][0]) - rl[q][0]; u1 = (u1 - rr[q][1]) - rl[q][1]; u2 = (u2 - rr[q][2]) - rl[q][2];
                if (ev[q] >= 0) {
                    u0 += ee[q][0]; u1 += ee[q][1]; u2 += ee[q][2];
                    for (int e = ev[q] + 1; e < evt_end && evt_pose[e] == s; ++e) {   // the same landmark twice in one message (rare)
                        u0 += Elmb[6 * (size_t)e + d]; u1 += Elmb[6 * (size_t)e + 2 + d]; u2 += Elmb[6 * (size_t)e + 4 + d];
                    }
                }
                Ys[(size_t)(3 * k) * LD] = u0; Ys[(size_t)(3 * k + 1) * LD] = u1; Ys[(size_t)(3 * k + 2) * LD] = u2;
            }
        }
    }
    __syncthreads();
    if (s_fail) {
        if (tid == 0) p.solve_ok[b] = 0;
        return;
    }
    for (int k = tid; k < NS; k += blockDim.x) {
        double* o = p.sepfac + ((size_t)b * p.nseg_max + k) * 16;
#pragma unroll
        for (int q = 0; q < 15; ++q) o[q] = s_sf[k][q];
    }
    if (c < 0 || c >= LD) return;
    if (c > m2) {   // columns the tile kernel's operand loads touch but never store
        for (int k = 0; k < 3 * NS; ++k) Ys[(size_t)k * LD] = 0.0;
        return;
    }
    // pass 2: the chain recurrence (each thread re-reads what it wrote itself)
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
#pragma unroll 4
    for (int k = 0; k < NS; ++k) {
        const double* o = s_sf[k];
        double u0 = Ys[(size_t)(3 * k) * LD], u1 = Ys[(size_t)(3 * k + 1) * LD], u2 = Ys[(size_t)(3 * k + 2) * LD];
        SEG_SUB_GV(o + 6, y0, y1, y2, u0, u1, u2);   // G is zero at the first separator
        y0 = o[0] * u0;
        y1 = o[1] * u0 + o[2] * u1;
        y2 = (o[3] * u0 + o[4] * u1) + o[5] * u2;
        Ys[(size_t)(3 * k) * LD] = y0; Ys[(size_t)(3 * k + 1) * LD] = y1; Ys[(size_t)(3 * k + 2) * LD] = y2;
    }
}

// ---- T_p = Y_p^T Y_p, the Gram matrix of one segment's columns (one workgroup per (slot, segment)): Y_p ([3 len][2 ncol + 1], just
//      written by pgs_seg_kernel: L2) goes through LDS in chunks of 16 rows (the next chunk's loads are in flight while this one feeds
//      v_mfma_f64_16x16x4_f64), the 16x16 tiles of the lower triangle are dealt to the four wavefronts.  The tile SYRK kernel subtracts
//      T_p from the rows / columns of S_ext the segment's local columns map to, segment after segment (pgs_syrk_kernel's epilogue:
//      a fixed order per element of S). ----
constexpr int GR_TPB = 256, GR_CH = 16, GR_LDL = 128 + 16, GR_TW = 9;   // 36 tiles at 128 columns / 4 wavefronts
__global__ __launch_bounds__(GR_TPB) void pgs_seg_gram_kernel(const PgsParams p) {
    __shared__ double s_c[GR_CH * GR_LDL];
    const int SL = p.seg_len, nsegl = seg_ns(p.N, SL) + 1;   // (the launch's decomposition, see pgs_seg_kernel)
    const int bl = blockIdx.x / nsegl, ps = blockIdx.x - bl * nsegl;
    const int b = pgs_slot(p, bl), tid = threadIdx.x;
    if (p.state[b] || !p.solve_ok[b]) return;
    const int N = pgs_N(p, b), NS = seg_ns(N, SL);
    if (ps > NS) return;
    const int lo = seg_lo(ps, SL), hi = seg_hi(ps, SL, NS, N), LD = p.LD;
    const int nc = 2 * p.seg_ncol[(size_t)b * p.nseg_max + ps] + 1, nr = 3 * (hi - lo), ncp = (nc + 15) & ~15;
    const int w = tid >> 6, lane = tid & 63, kq = lane >> 4, cl = lane & 15;
    const double* Yp = p.Y + (size_t)b * p.y_stride + (size_t)3 * lo * LD;
    const int nt = ncp >> 4, ntile = nt * (nt + 1) / 2;
    int ti[GR_TW], tj[GR_TW];
    dbl4_t acc[GR_TW];
#pragma unroll
    for (int q = 0; q < GR_TW; ++q) {
        int a = 0, tt = w + 4 * q;
        if (tt >= ntile) tt = 0;
        while (tt >= a + 1) { tt -= a + 1; a += 1; }
        ti[q] = a; tj[q] = tt;
        acc[q] = (dbl4_t){0.0, 0.0, 0.0, 0.0};
    }
    constexpr int PE = GR_CH * 128 / GR_TPB;   // elements of a chunk per thread
    double pre[PE];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int u = 0; u < PE; ++u) {
            const int idx = tid + GR_TPB * u, r = idx >> 7, cc = idx & 127;
            pre[u] = (r0 + r < nr && cc < nc) ? Yp[(size_t)(r0 + r) * LD + cc] : 0.0;
        }
    };
    fetch(0);
#pragma unroll 1
    for (int r0 = 0; r0 < nr; r0 += GR_CH) {
#pragma unroll
        for (int u = 0; u < PE; ++u) {
            const int idx = tid + GR_TPB * u;
            s_c[(idx >> 7) * GR_LDL + (idx & 127)] = pre[u];
        }
        __syncthreads();
        if (r0 + GR_CH < nr) fetch(r0 + GR_CH);
#pragma unroll
        for (int q = 0; q < GR_TW; ++q) {
            if (w + 4 * q >= ntile) break;   // wave-uniform
#pragma unroll
            for (int k = 0; k < GR_CH; k += 4)
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(s_c[(k + kq) * GR_LDL + 16 * ti[q] + cl], s_c[(k + kq) * GR_LDL + 16 * tj[q] + cl], acc[q], 0, 0, 0);
        }
        __syncthreads();
    }
    const int TLD = p.seg_tld;   // leading dimension of a segment's Gram matrix (>= its columns, a multiple of 16)
    double* Tp = p.segT + ((size_t)b * p.nseg_max + ps) * ((size_t)TLD * TLD);
#pragma unroll
    for (int q = 0; q < GR_TW; ++q) {
        if (w + 4 * q >= ntile) break;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) Tp[(size_t)(16 * ti[q] + kq + 4 * r4) * TLD + 16 * tj[q] + cl] = acc[q][r4];   // C/D layout: row = (lane >> 4) + 4 reg
    }
}

// ---- pose step H_pp dp = g_p - H_pl dl through the segmented factor: forward over the interiors (one lane per segment), the
//      separators (one lane), backward over the separators, backward over the interiors ----
constexpr int SB_TPB = 256;
__global__ __launch_bounds__(SB_TPB) void pgs_seg_backsolve_kernel(const PgsParams p) {
    __shared__ double s_sf[kPgsSegMaxSep][16];
    __shared__ double s_r[kPgsSegMaxSep + 1][6];
    __shared__ double s_ds[kPgsSegMaxSep + 2][3];
    const int b = pgs_slot(p, blockIdx.x), tid = threadIdx.x;
    if (p.state[b] || !p.solve_ok[b]) return;
    const int SL = p.seg_len, N = pgs_N(p, b), NS = seg_ns(N, SL), nseg = NS + 1, KP = p.KP;
    const Inst g = inst_view(p, b);
    const double* gpb = p.gp + (size_t)b * p.N_max * 3;
    const double* Eb = p.E + (size_t)b * p.N_max * KP * 6;
    const double* Lb = p.Linv + (size_t)b * p.N_max * 6;
    const double* Gb = p.G + (size_t)b * p.N_max * 9;
    const double* Gsb = p.Gs + (size_t)b * p.N_max * 9;
    const double* sob = p.segout + (size_t)b * p.nseg_max * 32;
    const double* dlb = p.dl + (size_t)b * p.L_max * 2;
    double* dpb = p.dp + (size_t)b * p.N_max * 3;
    for (int i = tid; i < N; i += SB_TPB) {   // u_i = g_i - E_i dl (into dp, overwritten in place by z and then by the step)
        double u0 = gpb[3 * i], u1 = gpb[3 * i + 1], u2 = gpb[3 * i + 2];
        const int kc = g.cnt[i];
        for (int s = 0; s < kc; ++s) {
            const size_t k = (size_t)i * KP + s;
            const int j = g.mlm[k] & (kPgsFirstBit - 1);
            const double* E = Eb + 6 * k;
            const double d0 = dlb[2 * j], d1 = dlb[2 * j + 1];
            u0 -= E[0] * d0 + E[1] * d1; u1 -= E[2] * d0 + E[3] * d1; u2 -= E[4] * d0 + E[5] * d1;
        }
        dpb[3 * i] = u0; dpb[3 * i + 1] = u1; dpb[3 * i + 2] = u2;
    }
    for (int k = tid; k < NS; k += SB_TPB) {
        const double* o = p.sepfac + ((size_t)b * p.nseg_max + k) * 16;
#pragma unroll
        for (int q = 0; q < 15; ++q) s_sf[k][q] = o[q];
    }
    __syncthreads();
    for (int ps = tid; ps < nseg; ps += SB_TPB) {   // forward over the interior poses of segment ps
        const int lo = seg_lo(ps, SL), hi = seg_hi(ps, SL, NS, N);
        double z0 = 0.0, z1 = 0.0, z2 = 0.0, r0 = 0.0, r1 = 0.0, r2 = 0.0;
        double f[24], fn[24];
#pragma unroll
        for (int q = 0; q < 6; ++q) fn[q] = Lb[6 * (size_t)lo + q];
#pragma unroll
        for (int q = 0; q < 9; ++q) { fn[6 + q] = Gb[9 * (size_t)lo + q]; fn[15 + q] = Gsb[9 * (size_t)lo + q]; }
        double un0 = dpb[3 * lo], un1 = dpb[3 * lo + 1], un2 = dpb[3 * lo + 2];
#pragma unroll 1
        for (int i = lo; i < hi; ++i) {
#pragma unroll
            for (int q = 0; q < 24; ++q) f[q] = fn[q];
            double u0 = un0, u1 = un1, u2 = un2;
            const int in = i + 1 < hi ? i + 1 : i;   // the next pose's factor is fetched under this pose's arithmetic
#pragma unroll
            for (int q = 0; q < 6; ++q) fn[q] = Lb[6 * (size_t)in + q];
#pragma unroll
            for (int q = 0; q < 9; ++q) { fn[6 + q] = Gb[9 * (size_t)in + q]; fn[15 + q] = Gsb[9 * (size_t)in + q]; }
            un0 = dpb[3 * in]; un1 = dpb[3 * in + 1]; un2 = dpb[3 * in + 2];
            SEG_SUB_GV(f + 6, z0, z1, z2, u0, u1, u2);   // Ginn is zero at the segment's first pose
            z0 = f[0] * u0;
            z1 = f[1] * u0 + f[2] * u1;
            z2 = (f[3] * u0 + f[4] * u1) + f[5] * u2;
            dpb[3 * i] = z0; dpb[3 * i + 1] = z1; dpb[3 * i + 2] = z2;
            r0 += (f[15] * z0 + f[16] * z1) + f[17] * z2;
            r1 += (f[18] * z0 + f[19] * z1) + f[20] * z2;
            r2 += (f[21] * z0 + f[22] * z1) + f[23] * z2;
        }
        const double* Gr = sob + (size_t)ps * 32 + 12;   // zero in the last segment
        s_r[ps][0] = r0; s_r[ps][1] = r1; s_r[ps][2] = r2;
        s_r[ps][3] = (Gr[0] * z0 + Gr[1] * z1) + Gr[2] * z2;
        s_r[ps][4] = (Gr[3] * z0 + Gr[4] * z1) + Gr[5] * z2;
        s_r[ps][5] = (Gr[6] * z0 + Gr[7] * z1) + Gr[8] * z2;
    }
    __syncthreads();
    if (tid == 0) {
        double z0 = 0.0, z1 = 0.0, z2 = 0.0;
#pragma unroll 1
        for (int k = 0; k < NS; ++k) {   // forward over the separators
            const int s = (k + 1) * SL;
            const double* o = s_sf[k];
            double u0 = (dpb[3 * s] - s_r[k][3]) - s_r[k + 1][0];
            double u1 = (dpb[3 * s + 1] - s_r[k][4]) - s_r[k + 1][1];
            double u2 = (dpb[3 * s + 2] - s_r[k][5]) - s_r[k + 1][2];
            SEG_SUB_GV(o + 6, z0, z1, z2, u0, u1, u2);
            z0 = o[0] * u0;
            z1 = o[1] * u0 + o[2] * u1;
            z2 = (o[3] * u0 + o[4] * u1) + o[5] * u2;
            s_ds[k + 1][0] = z0; s_ds[k + 1][1] = z1; s_ds[k + 1][2] = z2;
        }
        s_ds[NS + 1][0] = 0.0; s_ds[NS + 1][1] = 0.0; s_ds[NS + 1][2] = 0.0;
        s_ds[0][0] = 0.0; s_ds[0][1] = 0.0; s_ds[0][2] = 0.0;
#pragma unroll 1
        for (int k = NS - 1; k >= 0; --k) {   // backward: s_ds[k + 1] = step of separator k (0-based)
            double v0 = s_ds[k + 1][0], v1 = s_ds[k + 1][1], v2 = s_ds[k + 1][2];
            if (k + 1 < NS) { const double* Gn = s_sf[k + 1] + 6; SEG_SUB_GTV(Gn, s_ds[k + 2][0], s_ds[k + 2][1], s_ds[k + 2][2], v0, v1, v2); }
            const double* I = s_sf[k];
            const double d2 = I[5] * v2;
            const double d1 = I[2] * v1 + I[4] * v2;
            const double d0 = (I[0] * v0 + I[1] * v1) + I[3] * v2;
            s_ds[k + 1][0] = d0; s_ds[k + 1][1] = d1; s_ds[k + 1][2] = d2;
            const int s = (k + 1) * SL;
            dpb[3 * s] = d0; dpb[3 * s + 1] = d1; dpb[3 * s + 2] = d2;
        }
    }
    __syncthreads();
    for (int ps = tid; ps < nseg; ps += SB_TPB) {   // backward over the interior poses of segment ps
        const int lo = seg_lo(ps, SL), hi = seg_hi(ps, SL, NS, N);
        const double a0 = s_ds[ps][0], a1 = s_ds[ps][1], a2 = s_ds[ps][2];               // step of the left separator (zero for segment 0)
        double n0 = s_ds[ps + 1][0], n1 = s_ds[ps + 1][1], n2 = s_ds[ps + 1][2];          // step of the pose after i: first the right separator
        double Gn[9];   // coupling of pose i to the pose after it: Gright for the last interior pose, Ginn_{i+1} else
#pragma unroll
        for (int q = 0; q < 9; ++q) Gn[q] = sob[(size_t)ps * 32 + 12 + q];
        double fI[6], fS[9], fG[9], nI[6], nS[9], nG[9];
        const int e = hi - 1;
#pragma unroll
        for (int q = 0; q < 6; ++q) nI[q] = Lb[6 * (size_t)e + q];
#pragma unroll
        for (int q = 0; q < 9; ++q) { nS[q] = Gsb[9 * (size_t)e + q]; nG[q] = Gb[9 * (size_t)e + q]; }
        double zn0 = dpb[3 * e], zn1 = dpb[3 * e + 1], zn2 = dpb[3 * e + 2];
#pragma unroll 1
        for (int i = e; i >= lo; --i) {
#pragma unroll
            for (int q = 0; q < 6; ++q) fI[q] = nI[q];
#pragma unroll
            for (int q = 0; q < 9; ++q) { fS[q] = nS[q]; fG[q] = nG[q]; }
            double v0 = zn0, v1 = zn1, v2 = zn2;
            const int ip = i - 1 >= lo ? i - 1 : i;
#pragma unroll
            for (int q = 0; q < 6; ++q) nI[q] = Lb[6 * (size_t)ip + q];
#pragma unroll
            for (int q = 0; q < 9; ++q) { nS[q] = Gsb[9 * (size_t)ip + q]; nG[q] = Gb[9 * (size_t)ip + q]; }
            zn0 = dpb[3 * ip]; zn1 = dpb[3 * ip + 1]; zn2 = dpb[3 * ip + 2];
            SEG_SUB_GTV(Gn, n0, n1, n2, v0, v1, v2);
            SEG_SUB_GTV(fS, a0, a1, a2, v0, v1, v2);   // Gs is zero in segment 0
            n2 = fI[5] * v2;
            n1 = fI[2] * v1 + fI[4] * v2;
            n0 = (fI[0] * v0 + fI[1] * v1) + fI[3] * v2;
            dpb[3 * i] = n0; dpb[3 * i + 1] = n1; dpb[3 * i + 2] = n2;
#pragma unroll
            for (int q = 0; q < 9; ++q) Gn[q] = fG[q];   // Ginn_i couples pose i - 1 to pose i
        }
    }
}

// ---- the same pose step with the chains in LDS (N <= kSegBackLdsPoses).  Everything that does not depend on the neighbouring pose is
//      formed by all threads first - v_i = Linv_i u_i, M_i = Linv_i Ginn_i - so that a segment's forward chain is z_i = v_i - M_i z_{i-1}
//      out of LDS (9 multiply-adds per pose and an LDS round trip instead of 24 global loads); likewise backward with
//      w_i = Linv_i^T (z_i - Gs_i^T d_a), N_i = Linv_i^T Gn_i^T.  The kernel above keeps the oracle's association, (u - G z) first; this one
//      differs from it in the last bits, like the sequential path's scan (pgs_backsolve_kernel) always did. ----
constexpr int SBL_TPB = 1024, kSegBackLdsPoses = 1365;   // 12 doubles per pose: 128 KiB of dynamic LDS
__global__ __launch_bounds__(SBL_TPB) void pgs_seg_backsolve_lds_kernel(const PgsParams p) {
    extern __shared__ double s_vm[];   // [N][12]
    __shared__ double s_sf[kPgsSegMaxSep][16];
    __shared__ double s_r[kPgsSegMaxSep + 1][6];
    __shared__ double s_ds[kPgsSegMaxSep + 2][3];
    const int b = pgs_slot(p, blockIdx.x), tid = threadIdx.x;
    if (p.state[b] || !p.solve_ok[b]) return;
    const int SL = p.seg_len, N = pgs_N(p, b), NS = seg_ns(N, SL), nseg = NS + 1, KP = p.KP;
    const Inst g = inst_view(p, b);
    const double* gpb = p.gp + (size_t)b * p.N_max * 3;
    const double* Eb = p.E + (size_t)b * p.N_max * KP * 6;
    const double* Lb = p.Linv + (size_t)b * p.N_max * 6;
    const double* Gb = p.G + (size_t)b * p.N_max * 9;
    const double* Gsb = p.Gs + (size_t)b * p.N_max * 9;
    const double* sob = p.segout + (size_t)b * p.nseg_max * 32;
    const double* dlb = p.dl + (size_t)b * p.L_max * 2;
    double* dpb = p.dp + (size_t)b * p.N_max * 3;
    for (int k = tid; k < NS; k += SBL_TPB) {
        const double* o = p.sepfac + ((size_t)b * p.nseg_max + k) * 16;
#pragma unroll
        for (int q = 0; q < 15; ++q) s_sf[k][q] = o[q];
    }
    for (int i = tid; i < N; i += SBL_TPB) {   // u_i = g_i - E_i dl, then (v_i, M_i); a separator pose keeps its u
        double I[6], G[9];
#pragma unroll
        for (int q = 0; q < 6; ++q) I[q] = Lb[6 * (size_t)i + q];     // (issued before the factor loop's dependent loads)
#pragma unroll
        for (int q = 0; q < 9; ++q) G[q] = Gb[9 * (size_t)i + q];
        double u0 = gpb[3 * i], u1 = gpb[3 * i + 1], u2 = gpb[3 * i + 2];
        const int kc = g.cnt[i];
        for (int s = 0; s < kc; ++s) {
            const size_t k = (size_t)i * KP + s;
            const int j = g.mlm[k] & (kPgsFirstBit - 1);
            const double* E = Eb + 6 * k;
            const double d0 = dlb[2 * j], d1 = dlb[2 * j + 1];
            u0 -= E[0] * d0 + E[1] * d1; u1 -= E[2] * d0 + E[3] * d1; u2 -= E[4] * d0 + E[5] * d1;
        }
        double* o = s_vm + 12 * (size_t)i;
        const bool sep = i >= SL && i % SL == 0 && i / SL <= NS;
        if (sep) { o[0] = u0; o[1] = u1; o[2] = u2; continue; }
        const int ps = i / SL - ((i % SL == 0 && i > 0) ? 1 : 0);   // (an interior pose is no multiple of SL except pose 0)
        const bool first = i == seg_lo(ps, SL);
        o[0] = I[0] * u0;
        o[1] = I[1] * u0 + I[2] * u1;
        o[2] = (I[3] * u0 + I[4] * u1) + I[5] * u2;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {   // M = Linv Ginn (zero at the segment's first pose)
            o[3 + cc] = first ? 0.0 : I[0] * G[cc];
            o[6 + cc] = first ? 0.0 : I[1] * G[cc] + I[2] * G[3 + cc];
            o[9 + cc] = first ? 0.0 : (I[3] * G[cc] + I[4] * G[3 + cc]) + I[5] * G[6 + cc];
        }
    }
    __syncthreads();
    for (int ps = tid; ps < nseg; ps += SBL_TPB) {   // forward chains, one lane per segment
        const int lo = seg_lo(ps, SL), hi = seg_hi(ps, SL, NS, N);
        double z0 = 0.0, z1 = 0.0, z2 = 0.0;
#pragma unroll 2
        for (int i = lo; i < hi; ++i) {
            double* o = s_vm + 12 * (size_t)i;
            const double n0 = o[0] - ((o[3] * z0 + o[4] * z1) + o[5] * z2);
            const double n1 = o[1] - ((o[6] * z0 + o[7] * z1) + o[8] * z2);
            const double n2 = o[2] - ((o[9] * z0 + o[10] * z1) + o[11] * z2);
            z0 = n0; z1 = n1; z2 = n2;
            o[0] = z0; o[1] = z1; o[2] = z2;
        }
        const double* Gr = sob + (size_t)ps * 32 + 12;   // zero in the last segment
        s_r[ps][3] = (Gr[0] * z0 + Gr[1] * z1) + Gr[2] * z2;
        s_r[ps][4] = (Gr[3] * z0 + Gr[4] * z1) + Gr[5] * z2;
        s_r[ps][5] = (Gr[6] * z0 + Gr[7] * z1) + Gr[8] * z2;
    }
    __syncthreads();
    for (int i = tid; i < N; i += SBL_TPB) {   // Gs_i z_i of every interior pose (summed per segment below, in pose order)
        const bool sep = i >= SL && i % SL == 0 && i / SL <= NS;
        if (sep) continue;
        double* o = s_vm + 12 * (size_t)i;
        const double* Gs = Gsb + 9 * (size_t)i;
        const double z0 = o[0], z1 = o[1], z2 = o[2];
        o[3] = (Gs[0] * z0 + Gs[1] * z1) + Gs[2] * z2;
        o[4] = (Gs[3] * z0 + Gs[4] * z1) + Gs[5] * z2;
        o[5] = (Gs[6] * z0 + Gs[7] * z1) + Gs[8] * z2;
    }
    __syncthreads();
    for (int ps = tid; ps < nseg; ps += SBL_TPB) {
        const int lo = seg_lo(ps, SL), hi = seg_hi(ps, SL, NS, N);
        double r0 = 0.0, r1 = 0.0, r2 = 0.0;
        for (int i = lo; i < hi; ++i) { const double* o = s_vm + 12 * (size_t)i; r0 += o[3]; r1 += o[4]; r2 += o[5]; }
        s_r[ps][0] = r0; s_r[ps][1] = r1; s_r[ps][2] = r2;
    }
    __syncthreads();
    if (tid == 0) {
        double z0 = 0.0, z1 = 0.0, z2 = 0.0;
#pragma unroll 1
        for (int k = 0; k < NS; ++k) {   // forward over the separators
            const double* us = s_vm + 12 * (size_t)((k + 1) * SL);
            const double* o = s_sf[k];
            double u0 = (us[0] - s_r[k][3]) - s_r[k + 1][0];
            double u1 = (us[1] - s_r[k][4]) - s_r[k + 1][1];
            double u2 = (us[2] - s_r[k][5]) - s_r[k + 1][2];
            SEG_SUB_GV(o + 6, z0, z1, z2, u0, u1, u2);
            z0 = o[0] * u0;
            z1 = o[1] * u0 + o[2] * u1;
            z2 = (o[3] * u0 + o[4] * u1) + o[5] * u2;
            s_ds[k + 1][0] = z0; s_ds[k + 1][1] = z1; s_ds[k + 1][2] = z2;
        }
        s_ds[NS + 1][0] = 0.0; s_ds[NS + 1][1] = 0.0; s_ds[NS + 1][2] = 0.0;
        s_ds[0][0] = 0.0; s_ds[0][1] = 0.0; s_ds[0][2] = 0.0;
#pragma unroll 1
        for (int k = NS - 1; k >= 0; --k) {   // backward: s_ds[k + 1] = step of separator k (0-based)
            double v0 = s_ds[k + 1][0], v1 = s_ds[k + 1][1], v2 = s_ds[k + 1][2];
            if (k + 1 < NS) { const double* Gn = s_sf[k + 1] + 6; SEG_SUB_GTV(Gn, s_ds[k + 2][0], s_ds[k + 2][1], s_ds[k + 2][2], v0, v1, v2); }
            const double* I = s_sf[k];
            const double d2 = I[5] * v2;
            const double d1 = I[2] * v1 + I[4] * v2;
            const double d0 = (I[0] * v0 + I[1] * v1) + I[3] * v2;
            s_ds[k + 1][0] = d0; s_ds[k + 1][1] = d1; s_ds[k + 1][2] = d2;
            double* us = s_vm + 12 * (size_t)((k + 1) * SL);
            us[0] = d0; us[1] = d1; us[2] = d2;
        }
    }
    __syncthreads();
    for (int i = tid; i < N; i += SBL_TPB) {   // w_i = Linv_i^T (z_i - Gs_i^T d_a), N_i = Linv_i^T Gn_i^T
        const bool sep = i >= SL && i % SL == 0 && i / SL <= NS;
        if (sep) continue;
        const int ps = i / SL - ((i % SL == 0 && i > 0) ? 1 : 0);
        const int hi = seg_hi(ps, SL, NS, N);
        double I[6], Gs[9], Gn[9];
#pragma unroll
        for (int q = 0; q < 6; ++q) I[q] = Lb[6 * (size_t)i + q];
#pragma unroll
        for (int q = 0; q < 9; ++q) { Gs[q] = Gsb[9 * (size_t)i + q]; Gn[q] = i + 1 < hi ? Gb[9 * (size_t)(i + 1) + q] : sob[(size_t)ps * 32 + 12 + q]; }
        double* o = s_vm + 12 * (size_t)i;
        double v0 = o[0], v1 = o[1], v2 = o[2];
        SEG_SUB_GTV(Gs, s_ds[ps][0], s_ds[ps][1], s_ds[ps][2], v0, v1, v2);   // Gs is zero in segment 0
        o[0] = (I[0] * v0 + I[1] * v1) + I[3] * v2;
        o[1] = I[2] * v1 + I[4] * v2;
        o[2] = I[5] * v2;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {   // column cc of Gn^T = row cc of Gn
            o[3 + cc] = (I[0] * Gn[3 * cc] + I[1] * Gn[3 * cc + 1]) + I[3] * Gn[3 * cc + 2];
            o[6 + cc] = I[2] * Gn[3 * cc + 1] + I[4] * Gn[3 * cc + 2];
            o[9 + cc] = I[5] * Gn[3 * cc + 2];
        }
    }
    __syncthreads();
    for (int ps = tid; ps < nseg; ps += SBL_TPB) {   // backward chains
        const int lo = seg_lo(ps, SL), hi = seg_hi(ps, SL, NS, N);
        double d0 = s_ds[ps + 1][0], d1 = s_ds[ps + 1][1], d2 = s_ds[ps + 1][2];   // step of the right separator (zero after the last segment)
#pragma unroll 2
        for (int i = hi - 1; i >= lo; --i) {
            double* o = s_vm + 12 * (size_t)i;
            const double n0 = o[0] - ((o[3] * d0 + o[4] * d1) + o[5] * d2);
            const double n1 = o[1] - ((o[6] * d0 + o[7] * d1) + o[8] * d2);
            const double n2 = o[2] - ((o[9] * d0 + o[10] * d1) + o[11] * d2);
            d0 = n0; d1 = n1; d2 = n2;
            o[0] = d0; o[1] = d1; o[2] = d2;
        }
    }
    __syncthreads();
    for (int i = tid; i < N; i += SBL_TPB) {
        const double* o = s_vm + 12 * (size_t)i;
        dpb[3 * i] = o[0]; dpb[3 * i + 1] = o[1]; dpb[3 * i + 2] = o[2];
    }
}
