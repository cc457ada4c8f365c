// ekf_step_stream.h — part of the BODY of ekf_step_kernel (ekf_kernel_impl.h includes it inside the kernel function; round 6: the 1 840-line kernel split into its
// parts, pure moves - every object file byte-identical).  The passes over P in HBM: the in-place bulk stream with the deferred rank-2 updates (`stream_pass`), layout changes through the second buffer (`mid_pass`), vehicle rows / columns (`write_vehicle`), gathers of newly visible landmarks' rows (`pregather`).
// Lambdas and statements here capture the kernel's locals (p, tid, lane, the LDS arrays ...): not a stand-alone header.  DESIGN.md 4.1.

    struct PassArgs {
        const ST* src; ST* dst; double* mid;
        int nf, ldd, lds, nsrc, nu;   // state size / leading dimension written, leading dimension / valid size of the source, updates
        int lo;                       // RING passes (decoupled loop): update w of the pass lives in slot (lo + w) % KG
    };
    // ---- BULK: stream P once, in strips of R = UNR consecutive rows.  Work item `it` = (strip s, vector column j): the
    //      lane owns the 16-byte vectors (R*s + i, VEC*j .. VEC*j + VEC-1), i < R.  Its (H P) operands (VEC per update)
    //      are read once per strip and re-used for the R rows; K[r] (one 16-byte read per row and update) is the
    //      same address for every lane of the strip (LDS broadcast): (VEC + R) operand reads per R*VEC elements and
    //      update instead of two per element.  64 consecutive items form a chunk; chunks are handed to wavefronts
    //      dynamically.  Every vector is read and rewritten by the same lane, so the update is in place unless the
    //      step changes the leading dimension (insertions), which writes the other buffer.
    //      Thin patches.  The thin copies in LDS undergo, element for element, the same operations in the same order
    //      as the stream applies (the downdates), EXCEPT where the prediction touched them (rows / columns 0, 1 and
    //      the (2,2) element) and where a landmark is new.  So the common pass (same layout, single group) patches
    //      only those from LDS (FAST); passes that insert landmarks or belong to a multi-group step patch every thin
    //      row / column (general), like the thin phase assumes. ----
    constexpr int R = UNR;
    auto stream_pass = [&](auto fast_tag, const PassArgs& pa) {
        // mode 1: FAST (same layout, patches only where the prediction touches); 0: general (every thin row / column
        // patched, layout may change); 2: RING = FAST without any patch, updates taken from the ring of the decoupled loop
        constexpr int MODE = decltype(fast_tag)::value;
        constexpr bool FAST = MODE != 0;
        constexpr bool RING = MODE == 2;
        const ST* __restrict__ srcb = pa.src;
        const int nf = pa.nf, ldd = pa.ldd, lds = pa.lds, nsrc = pa.nsrc, nu = pa.nu;
        const int nv = ldd / VEC;                       // vectors per row
        const int nstrip = (nf + R - 1) / R;
        const int nitem = SLAM_DBG(p.dbg & 1) ? 0 : nstrip * nv;
        const float inv_nv = 1.0f / (float)nv;
        const VT* __restrict__ src2 = reinterpret_cast<const VT*>(srcb);
        VT* __restrict__ dst2 = reinterpret_cast<VT*>(pa.dst);
        const int lsv = lds / VEC;
        auto next_chunk = [&]() -> int {
            int ch = 0;
            if (lane == 0) ch = atomicAdd(&s_chunk, 1);
            return __builtin_amdgcn_readfirstlane(ch);
        };
        // item -> (strip, vector column) without an integer division: (it + 0.5) / nv is at least 0.5 / nv away from
        // an integer and the float product is off by < 1e-5 at these magnitudes.
        // FAST passes are branch-free: items beyond the end are clamped to the last item and rows beyond the last row
        // of the last strip to the last row, so those lanes redo a neighbour's work and store the same bits to the same
        // addresses (within one wave-instruction, after all loads of the chunk).  With every load and store issued
        // unconditionally the compiler can count them, so its s_waitcnt for the loads of a chunk leaves the stores and
        // the prefetch of the next chunk in flight (a store behind a divergent branch forces vmcnt(0) instead).
        auto decode = [&](int ch, int& it, int& sidx, int& j) {
            it = ch * 64 + opaque(lane);
            if (FAST) it = it < nitem ? it : nitem - 1;
            sidx = (int)(((float)it + 0.5f) * inv_nv);
            j = it - sidx * nv;
        };
        // the loads of one chunk: R 16-byte vectors per lane, issued back to back
        auto issue = [&](int ch, VT (&raw)[R]) {
            int it, sidx, j;
            decode(ch, it, sidx, j);
            const int r0 = sidx * R;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int r = r0 + i;
                if constexpr (FAST) {
                    raw[i] = src2[(r < nf ? r : nf - 1) * nv + j];
                } else {
                    const bool ok = it < nitem && r < nsrc && j < lsv;
                    VT z;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) z[e] = (ST)0;
                    raw[i] = ok ? src2[r * lsv + j] : z;
                }
            }
        };
        // downdates, patches, storage rounding and the stores of one chunk
        auto process = [&](int ch, const VT (&raw)[R]) {
            int it, sidx, j;
            decode(ch, it, sidx, j);
            const bool act = FAST || it < nitem;
            const int r0 = sidx * R, c0 = j * VEC;
            int rr[R];   // row of vector i (FAST: clamped to the last row)
#pragma unroll
            for (int i = 0; i < R; ++i) rr[i] = (FAST && r0 + i >= nf) ? nf - 1 : r0 + i;
            double val[R][VEC];
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    val[i][e] = (double)raw[i][e];
                    if (!FAST && !(rr[i] < nsrc && c0 + e < nsrc)) val[i][e] = 0.0;   // nothing there yet
                }
#pragma unroll
            for (int w = 0; w < KG; ++w) {
                if (w >= nu) break;  // wave-uniform
                const int sw = RING ? (pa.lo + w) % KG : w;   // slot of update w
                double2 hp[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) hp[e] = s_HP[sw * HPW + e * HS + j];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const double2 kk = s_K[sw * LDP + rr[i]];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) val[i][e] = val[i][e] - (kk.x * hp[e].x + kk.y * hp[e].y);
                }
                if constexpr (!kWide) {
                    // fp32 storage rounds P at the end of every timestep; a group that spans several timesteps rounds
                    // where they end (wave-uniform flag per update)
                    if (s_wend[sw]) {
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int e = 0; e < VEC; ++e) val[i][e] = (double)(ST)val[i][e];
                    }
                }
            }
            if constexpr (RING) {
                // no patches: rows / columns 0, 1 and (2,2) of P in HBM are not maintained inside the decoupled loop
                // (nobody reads them there; they are written from the thin copies when the loop ends)
            } else if constexpr (FAST) {
                if (j == 0) {   // columns 0, 1 (the prediction changed them)
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        val[i][0] = s_C[rr[i]];
                        val[i][1] = s_C[LDP + rr[i]];
                    }
                }
                if (sidx == 0) {   // rows 0, 1
                    static_assert(R >= 2, "rows 0 and 1 must share a strip");
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        val[0][e] = s_R[c0 + e];
                        val[1][e] = s_R[LDP + c0 + e];
                    }
                }
                if (sidx == 2 / R && c0 <= 2 && 2 < c0 + VEC) {   // (2,2)
                    const double p22 = s_R[2 * LDP + 2];
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            if (rr[i] == 2 && c0 + e == 2) val[i][e] = p22;
                }
            } else {
                int sc[VEC], sr[R];
#pragma unroll
                for (int e = 0; e < VEC; ++e) sc[e] = s_slot[c0 + e];
#pragma unroll
                for (int i = 0; i < R; ++i) sr[i] = s_slot[rr[i]];
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        if (sc[e] >= 0) val[i][e] = s_C[sc[e] * LDP + rr[i]];
                        if (sr[i] >= 0) val[i][e] = s_R[sr[i] * LDP + c0 + e];
                        if (c0 + e >= nf) val[i][e] = 0.0;   // pad columns stay zero
                    }
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                VT o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const ST stored = (ST)val[i][e];   // storage rounding (identity for fp64)
                    const unsigned ha = hi_abs((double)stored);
                    hiacc = ((FAST || (act && rr[i] < nf)) && hiacc < ha) ? ha : hiacc;
                    o[e] = stored;
                }
                if (FAST || (act && rr[i] < nf)) dst2[rr[i] * nv + j] = o;
            }
        };
        if constexpr (PIPE && FAST) {
            // Software pipeline over two register sets: the loads of the next chunk are in flight while this one is
            // updated and stored.  A chunk index beyond the end loads (clamped) the last item and is never processed.
            VT bufA[R], bufB[R];
            int ca = next_chunk();
            issue(ca, bufA);
#pragma unroll 1
            while (ca * 64 < nitem) {
                const int cb = next_chunk();
                issue(cb, bufB);
                process(ca, bufA);
                if (cb * 64 >= nitem) break;
                ca = next_chunk();
                issue(ca, bufA);
                process(cb, bufB);
            }
        } else {
#pragma unroll 1
            for (;;) {
                const int ch = next_chunk();
                if (ch * 64 >= nitem) break;
                VT raw[R];
                issue(ch, raw);
                process(ch, raw);
            }
        }
    };
    // fp32 storage with more than KG detections in one step (rare): the matrix between the groups stays fp64 in the
    // per-instance scratch slab so that storage rounding happens exactly once per step.  Element-wise, one vector
    // of one row per lane, every thin row / column patched.
    auto mid_pass = [&](bool src_mid, bool dst_mid, const PassArgs& pa) {
        const int nf = pa.nf, ldd = pa.ldd, lds = pa.lds, nsrc = pa.nsrc, nu = pa.nu;
        double* const Pmid = pa.mid;
        const ST* const Pin = pa.src;
        ST* const Pout = pa.dst;
        const int nv = ldd / VEC;
        const int nitem = SLAM_DBG(p.dbg & 1) ? 0 : nf * nv;
#pragma unroll 1
        for (;;) {
            int ch = 0;
            if (lane == 0) ch = atomicAdd(&s_chunk, 1);
            ch = __builtin_amdgcn_readfirstlane(ch);
            if (ch * 64 >= nitem) break;
            const int it = ch * 64 + opaque(lane);
            if (it < nitem) {
                const int r = it / nv, c0 = (it - r * nv) * VEC;
                const int srw = s_slot[r];
                VT o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int c = c0 + e;
                    double v = 0.0;
                    if (r < nsrc && c < nsrc) v = src_mid ? Pmid[(size_t)r * lds + c] : (double)Pin[(size_t)r * lds + c];
                    for (int w = 0; w < nu; ++w) {
                        const double2 kk = s_K[w * LDP + r], hh = s_HP[w * HPW + hpi(c)];
                        v = v - (kk.x * hh.x + kk.y * hh.y);
                    }
                    const int scl = s_slot[c];
                    if (scl >= 0) v = s_C[scl * LDP + r];
                    if (srw >= 0) v = s_R[srw * LDP + c];
                    if (c >= nf) v = 0.0;
                    if (dst_mid) {
                        Pmid[(size_t)r * ldd + c] = v;
                    } else {
                        const ST stored = (ST)v;
                        const unsigned ha = hi_abs((double)stored);
                        hiacc = hiacc > ha ? hiacc : ha;
                        o[e] = stored;
                    }
                }
                if (!dst_mid) reinterpret_cast<VT*>(Pout)[it] = o;
            }
        }
    };
    // The three vehicle rows and columns of P from their LDS copies (what the prediction changes, ekf.cpp:61 with the sparse
    // F_x, F_v) into a matrix of state size n: all a step without update or insertion has to write.
    auto write_vehicle = [&](ST* Pbuf, int n) {
        const int ldn = ekf_ld(n, ESZ);
        const int tsk = opaque(tid);
        if (tsk == 0) count_other(s_cnt, 6 * n - 9);
#pragma unroll 1
        for (int i = tsk; i < 3 * n; i += TPB) {
            const int r = i / n, c = i - r * n;
            const ST sv = (ST)s_R[r * LDP + c];                       // P[r][c], r < 3
            const unsigned ha = hi_abs((double)sv);
            hiacc = hiacc > ha ? hiacc : ha;
            Pbuf[(size_t)r * ldn + c] = sv;
        }
#pragma unroll 1
        for (int i = tsk; i < 3 * (n - 3); i += TPB) {
            const int c = i / (n - 3), r = 3 + (i - c * (n - 3));
            const ST sv = (ST)s_C[c * LDP + r];                       // P[r][c], c < 3 <= r
            const unsigned ha = hi_abs((double)sv);
            hiacc = hiacc > ha ? hiacc : ha;
            Pbuf[(size_t)r * ldn + c] = sv;
        }
    };

    __syncthreads();
    SLAM_STAMP(0);   // initial loads
    // The vehicle rows / columns of P are needed by every launch: the wavefronts that do not run the pre-step fetch them
    // meanwhile (a single-wavefront workgroup does it first), so the first group formation finds them resident.
    auto pregather = [&](int i0, int istride) {
        const int ldi = ekf_ld(n_init, ESZ);
#pragma unroll 1
        for (int i = i0; i < 3 * LDP; i += istride) {
            const int sl = i / LDP, j = i - sl * LDP;
            double rv = 0.0, cv = 0.0;
            if (j < n_init) {
                rv = (double)PA[(size_t)sl * ldi + j];   // P[sl][j]
                cv = (double)PA[(size_t)j * ldi + sl];   // P[j][sl]
            }
            s_R[i] = rv;
            s_C[i] = cv;
        }
    };
