// pgs_linearize.h — slot mapping, pgs_lm_begin_kernel, the linearisation (one thread per factor, then per pose / landmark).
// Part of pgs_kernel.hip (round 6: split by phase, pure moves); included there inside namespace slam { namespace {.  DESIGN.md 4.4.
#pragma once

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
