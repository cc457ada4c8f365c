// pgs_syrk.h — the Schur complement on v_mfma_f64_16x16x4_f64: tile kernel (with the segmented path's epilogue), instance-resident kernel, fused chain + SYRK kernel.
// Part of pgs_kernel.hip (round 6: split by phase, pure moves); included there inside namespace slam { namespace {.  DESIGN.md 4.4.
#pragma once

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
            const int TLD = p.seg_tld;
            const size_t TSZ = (size_t)TLD * TLD;
            const double* Tb = p.segT + (size_t)b * p.nseg_max * TSZ;
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
                            t[u] = q[u] >= 0 ? Tb[(size_t)ps * TSZ + (size_t)(2 * nl[u]) * TLD + 2 * q[u] + d] : 0.0;
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
                    const double* Tp = Tb + (size_t)ps * TSZ;
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4)
                                if (lr[i][r4] >= 0 && lc[j] >= 0 && lc[j] <= lr[i][r4]) acc[i][j][r4] = acc[i][j][r4] - Tp[(size_t)lr[i][r4] * TLD + lc[j]];
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
