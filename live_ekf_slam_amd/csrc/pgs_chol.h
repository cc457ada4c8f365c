// pgs_chol.h — dense Cholesky of the Schur complement + substitutions: right-looking and left-looking builds.
// Part of pgs_kernel.hip (round 6: split by phase, pure moves); included there inside namespace slam { namespace {.  DESIGN.md 4.4.
#pragma once

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
