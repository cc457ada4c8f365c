// ekf_step_lockstep.h — part of the BODY of ekf_step_kernel (ekf_kernel_impl.h includes it inside the kernel function; round 6: the 1 840-line kernel split into its
// parts, pure moves - every object file byte-identical).  The barrier-synchronised timestep: insertions, unknown ids, freezes, more detections than the ring holds, single-step launches.
// Lambdas and statements here capture the kernel's locals (p, tid, lane, the LDS arrays ...): not a stand-alone header.  DESIGN.md 4.1.

    int nf = n_old + 2 * n_ins;           // leading dimension of the matrix written this step
    nf = nf < NMAX ? nf : NMAX;
    ST* const Pout = (nf != n_old) ? (Pcur == PA ? PB : PA) : Pcur;   // in place unless the layout changes
    double* const Pmid = kWide ? reinterpret_cast<double*>(Pout) : (p.scratch + (size_t)b * p.pstride);

    // x_pred of the vehicle (ekf.cpp:56-59) was computed by the pre-step; it is needed before the first group because
    // unknown-id association (ekf.cpp:82-98) projects detections with the PREDICTED pose.  The covariance part of the
    // prediction runs on the thin rows/cols of the first group.
    if (tid < 3) s_xp[tid] = s_ps[10 * pb + tid];

    // ------------------------------------------------------------------------------------------------------
    // groups of <= KG detections
    // ------------------------------------------------------------------------------------------------------
    int l0 = 0;
    bool first = true;
    while (first || l0 < k) {
        // Source of this group's P: the old buffer (leading dimension n_old) for the first group; afterwards the
        // matrix written by the previous group's bulk pass (leading dimension nf).  Intermediate results between
        // groups stay in fp64: for fp32 storage they live in a per-instance fp64 scratch slab, so storage rounding
        // happens exactly once per step (when the last group writes P_out).
        const int ldd = ekf_ld(nf, ESZ);                       // leading dimension of the matrix this step writes
        const int lds = first ? ekf_ld(n_old, ESZ) : ldd;      // leading dimension of the source
        const int nsrc = first ? n_old : na;                   // rows/cols of the source that hold data

        // ---- form the group: thread 0 decides, everybody reads.  Thin rows/cols of landmarks that are detected
        //      again stay where they are (their LDS copy IS the current P row); the others give their slot up. ----
        __syncthreads();
        if (p.id_known) {
            // Known ids: the landmark of every detection is known from the pre-step (didx), so the whole formation is
            // lane-parallel in wavefront 0: lane l <-> detection l0 + l of the group AND thin slot pair l; votes via
            // ballot, a handful of LDS round trips instead of a serial chain of them on the critical path of every step.
            if (tid < 64) {
                int fb = 0, lim = KP;
                if (first && nu > 0) {   // pre-flush decision (see the serial path below for the rules)
                    int kupd = 0;
#pragma unroll 1
                    for (int q0 = 0; q0 < k; q0 += 64) {
                        const int q = q0 + lane;
                        const bool isupd = q < k && didx_t[q < k ? q : 0] >= 0 && didx_t[q < k ? q : 0] < M;
                        kupd += __popcll(__ballot(isupd));
                    }
                    fb = (frz_top || n_ins > 0 || nu + kupd > KG) ? 1 : 0;
                    lim = fb ? KP : (KG - nu < KP ? KG - nu : KP);
                }
                int l1g, nTg;
                const int needg = form_known(didx_t, k, l0, lim, nsrc, l1g, nTg);
                if (lane == 0) {
                    s_chunk = 0;
                    s_misc[4] = l1g;
                    s_misc[5] = nTg;
                    s_misc[2] = 0;
                    s_misc[7] = (first && nu > 0 && (fb || needg)) ? 1 : 0;
                }
            }
        } else if (tid == 0) {
            int l1 = l0, na_g = na, M_g = M;
            int frz = 0;
            int want[KP], nw = 0;
#pragma unroll
            for (int w = 0; w < KP; ++w) want[w] = -1;
            // Pre-flush: the open group (nu pending updates of earlier timesteps) is streamed into P BEFORE this step if the
            // step cannot join it: it inserts landmarks (layout change), its updates do not fit into the free slots, it
            // needs a thin row / column from HBM (which must then be current), or the instance freezes.
            int fb = 0, lim = KP;
            if (first && nu > 0) {
                int kupd = 0;
                for (int l = 0; l < k; ++l) kupd += (didx_t[l] >= 0 && didx_t[l] < M_g) ? 1 : 0;
                fb = (frz_top || n_ins > 0 || nu + kupd > KG) ? 1 : 0;
                lim = fb ? KP : (KG - nu < KP ? KG - nu : KP);
            }
            int needg = 0;
#pragma unroll 1
            while (l1 < k && l1 - l0 < lim) {
                int idx;
                if (p.id_known) {
                    idx = didx_t[l1];
                } else if (l1 == l0) {
                    // unknown ids (ekf.cpp:82-98): associate against the CURRENT x_pred, one detection per group
                    const float r_m = meas_t[3 * l1 + 1], b_m = meas_t[3 * l1 + 2];
                    double s, c;
                    det_sincos(s_xp[2] + (double)b_m, &s, &c);
                    const float x_det = (float)(s_xp[0] + (double)r_m * c);
                    const float y_det = (float)(s_xp[1] + (double)r_m * s);
                    idx = -2;
#pragma unroll 1
                    for (int j = 0; j < M_g; ++j) {
                        const float xd = assoc_abs((double)x_det - s_xp[3 + 2 * j], p.abs_is_int);       // ekf.cpp:91-92: which `abs`
                        const float yd = assoc_abs((double)y_det - s_xp[3 + 2 * j + 1], p.abs_is_int);
                        if (xd < p.min_sep && yd < p.min_sep) { idx = j; break; }
                    }
                    if (idx == -2) idx = (M_g < p.L_max && M_g < LMAX && na_g + 2 <= nf) ? M_g : -1;
                    if (idx == -1) s_misc[3] = 1;
                    if (idx >= 0 && idx < M_g && 2 * idx + 4 >= n_old) frz = 1;  // matched a landmark inserted this step
                    didx_t[l1] = idx;
                } else {
                    break;
                }
                if (idx >= 0) {
                    const int ii = 3 + 2 * idx;
                    bool have = false;
#pragma unroll
                    for (int w = 0; w < KP; ++w) have = have || (want[w] == ii);
                    if (!have) {   // at most KG detections per group, so a pair is always free
#pragma unroll
                        for (int w = 0; w < KP; ++w)
                            if (w == nw) want[w] = ii;
                        nw += 1;
                    }
                    if (idx >= M_g) { M_g += 1; na_g += 2; }
                }
                l1 += 1;
            }
            // release the pairs this group does not touch: the last bulk pass already wrote them to HBM
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const int ii = s_T[3 + 2 * j];
                if (ii >= 0) {
                    bool keep = false;
#pragma unroll
                    for (int w = 0; w < KP; ++w) keep = keep || (want[w] == ii);
                    if (!keep) {
                        s_slot[ii] = (signed char)-1; s_slot[ii + 1] = (signed char)-1;
                        s_T[3 + 2 * j] = -1; s_T[4 + 2 * j] = -1;
                    }
                }
            }
            // every wanted landmark without a slot takes a free pair
#pragma unroll
            for (int w = 0; w < KP; ++w) {
                const int ii = want[w];
                if (ii >= 0 && s_slot[ii] < 0) {
                    int j = 0;
                    while (j < KP - 1 && s_T[3 + 2 * j] >= 0) ++j;
                    s_T[3 + 2 * j] = ii; s_T[4 + 2 * j] = ii + 1;
                    s_slot[ii] = (signed char)(3 + 2 * j); s_slot[ii + 1] = (signed char)(4 + 2 * j);
                    const signed char nd = (signed char)(ii < nsrc ? 1 : 2);   // known landmark: gather, new one: zeros
                    s_need[3 + 2 * j] = nd; s_need[4 + 2 * j] = nd;
                    needg |= (nd == 1) ? 1 : 0;
                }
            }
            int nT = 3;
#pragma unroll
            for (int j = 0; j < KP; ++j)
                if (s_T[3 + 2 * j] >= 0) nT = 5 + 2 * j;
            s_chunk = 0;
            s_misc[4] = l1;
            s_misc[5] = nT;      // high-water mark: slots [3, nT) may contain free pairs (s_T < 0)
            s_misc[2] = frz;
            s_misc[7] = (first && nu > 0 && (fb || needg)) ? 1 : 0;
        }
        __syncthreads();
        const int l1 = s_misc[4], nT = s_misc[5];
        SLAM_STAMP(3);   // x_pred + group formation
        if (first && s_misc[7]) {
            // ---- pre-flush: stream the open group into P in place (layout of the previous step); the patches of rows /
            //      columns 0, 1 and (2,2) come from the thin copies, which hold the END of the previous step (this step's
            //      prediction has not touched them yet) ----
            PassArgs pa;
            pa.lo = 0;
            pa.src = Pin; pa.dst = const_cast<ST*>(Pin); pa.mid = nullptr;
            pa.nf = n_old; pa.ldd = lds; pa.lds = lds; pa.nsrc = n_old; pa.nu = nu;
            __builtin_amdgcn_s_setprio(0);
            if (tid == 0) count_pass(s_cnt, 2 * n_old * (lds / VEC), nu);
            stream_pass(std::integral_constant<int, 1>{}, pa);
            __builtin_amdgcn_s_setprio(SLAM_PRIO_THIN);
            nu = 0;
            __syncthreads();   // P in HBM is current (the gather below reads it); every wave is done with s_chunk / s_wend
            if (tid == 0) s_chunk = 0;
            if (tid < KG) s_wend[tid] = 0;
            SLAM_STAMP(9);   // pre-flush pass
        }
        if (first && frz_top) {   // duplicate new id (ekf.cpp:115 would index out of range): freeze in the pre-step state
            // rows / columns the deferred predictions changed (at the first step of a launch P in HBM is current and the
            // thin copies have not been gathered yet)
            if (!s_misc[7] && t > 0) write_vehicle(const_cast<ST*>(Pin), n_old);
            frz_at = t; frz_M = M_old; frz_n = n_old; frz_P = Pin;
            break;   // one exit for freezing instances, after the timestep loop
        }
        if (s_misc[2]) {
            // unknown-id quirk: the reference throws.  Freeze in the pre-step state.
            frz_at = t; frz_M = M_old; frz_n = n_old; frz_P = Pin;
            break;
        }
        if (s_misc[3]) flags |= SLAM_INST_CAPACITY;

        // ---- thin gather: HBM -> LDS.  Rows are contiguous, columns are strided 8-byte loads.  All loads of a
        //      lane are issued before the first LDS store so their latencies overlap. ----
        {
            const bool src_mid = !first;
            const int tg = opaque(tid);
            const ST* srcS = (kWide && src_mid) ? reinterpret_cast<const ST*>(Pmid) : Pin;
            constexpr int GI = (TS * LDP + TPB - 1) / TPB;
            // the loaded values stay in their storage type until every load of the lane has been issued: a conversion
            // next to its load would make each load wait for the previous one
            auto gather = [&](auto zero, const auto* __restrict__ base) {
                typedef decltype(zero) LT;
                LT rv[GI], cv[GI];
#pragma unroll
                for (int u = 0; u < GI; ++u) {
                    const int i = tg + TPB * u;
                    rv[u] = (LT)0; cv[u] = (LT)0;
                    if (i < nT * LDP) {
                        const int sl = i / LDP, j = i - sl * LDP;
                        const int t_s = s_T[sl];
                        if (s_need[sl] == 1 && j < nsrc && t_s < nsrc) {
                            rv[u] = base[(size_t)t_s * lds + j];   // P[t_s][j]
                            cv[u] = base[(size_t)j * lds + t_s];   // P[j][t_s]
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < GI; ++u) {
                    const int i = tg + TPB * u;
                    if (i < nT * LDP && s_need[i / LDP] != 0) { s_R[i] = (double)rv[u]; s_C[i] = (double)cv[u]; }
                }
            };
            if (!kWide && src_mid) gather(0.0, Pmid);
            else gather((ST)0, srcS);
            if (tid < nT && s_need[tid] == 1 && s_T[tid] < nsrc) count_other(s_cnt, 2 * nsrc);
        }
        __syncthreads();
        // Entries of a gathered row / column against the vehicle states come from the RESIDENT vehicle columns / rows: a timestep
        // without update or insertion writes nothing to HBM (its prediction lives in the thin copies only), so P[t_s][0..2] and
        // P[0..2][t_s] in HBM may lack the predictions since the last pass.  (Round 3's soak saw this as a wrong vehicle row in the
        // one-wavefront variant - five timesteps without detections, then an update of a mapped landmark, in one launch; the
        // kernels with a decoupled loop reach the same code only through two consecutive steps that skip it, e.g. capacity skips.)
        if (tid < 3) {
#pragma unroll 1
            for (int sl = 3; sl < nT; ++sl) {
                const int t_s = s_T[sl];
                if (s_need[sl] == 1 && (unsigned)t_s < (unsigned)nsrc) {
                    s_R[sl * LDP + tid] = s_C[tid * LDP + t_s];   // P[t_s][c], c < 3
                    s_C[sl * LDP + tid] = s_R[tid * LDP + t_s];   // P[r][t_s], r < 3
                }
            }
        }
        __syncthreads();
        if (tid < TS) s_need[tid] = 0;
        SLAM_STAMP(4);   // thin gather
        // ---- prediction stage on the thin copies (first group only), ekf.cpp:41-61.  The operands are row 2 / column 2 of
        //      P_t as they are BEFORE the prediction; of those only P[2][0..2] and P[0..2][2] change, all of them owned by the
        //      thread of thin slot 2, which keeps its new values in registers until everybody has read the old ones (one
        //      barrier, no copies of the operand row / column). ----
        if (first) {
            const int tp = opaque(tid);
            const double* const ps = s_ps + 10 * pb;   // F_x(0,2), F_x(1,2), F_v V F_v^T from the pre-step
            const double* const r2o = s_R + 2 * LDP;   // P_t[2][.]
            const double* const c2o = s_C + 2 * LDP;   // P_t[.][2]
            const double fa = ps[3], fb = ps[4];
            const double p22 = r2o[2];
            auto predicted = [&](double t, int r, int cc) -> double {
                const double f_r = r == 0 ? fa : fb;
                if (r < 2) t = t + f_r * r2o[cc];                  // rows 0,1 of F_x * P
                if (cc < 2) {                                      // cols 0,1 of (F_x P) F_x^T
                    double a2 = c2o[r];
                    if (r < 2) a2 = a2 + f_r * p22;
                    t = t + a2 * (cc == 0 ? fa : fb);
                }
                if (r < 2 && cc < 2) t = t + ps[5 + 2 * r + cc];    // + F_v V F_v^T
                if (r == 2 && cc == 2) t = t + p.V11;
                return t;
            };
            // only rows 0,1 / cols 0,1 / (2,2) of P change: thin rows 0,1 and thin cols 0,1 entirely ...
#pragma unroll 1
            for (int i = tp; i < 2 * LDP; i += TPB) {
                const int sl = i >= LDP ? 1 : 0, j = i - sl * LDP;
                if (j < na) {
                    s_R[i] = predicted(s_R[i], sl, j);     // R[sl][j] = P[sl][j]
                    s_C[i] = predicted(s_C[i], j, sl);     // C[sl][j] = P[j][sl]
                }
            }
            // ... and entries 0,1 (+ the (2,2) element) of every other thin row / col
            double n_r0 = 0.0, n_r1 = 0.0, n_c0 = 0.0, n_c1 = 0.0, n_22 = 0.0;
            if (tp >= 2 && tp < nT) {
                const int t_s = s_T[tp];
                if ((unsigned)t_s < (unsigned)na) {
                    n_r0 = predicted(s_R[tp * LDP + 0], t_s, 0);
                    n_r1 = predicted(s_R[tp * LDP + 1], t_s, 1);
                    n_c0 = predicted(s_C[tp * LDP + 0], 0, t_s);
                    n_c1 = predicted(s_C[tp * LDP + 1], 1, t_s);
                    if (tp == 2) {
                        n_22 = predicted(p22, 2, 2);
                    } else {
                        s_R[tp * LDP + 0] = n_r0; s_R[tp * LDP + 1] = n_r1;
                        s_C[tp * LDP + 0] = n_c0; s_C[tp * LDP + 1] = n_c1;
                    }
                }
            }
            __syncthreads();
            if (tp == 2) {   // slot 2 is state index 2 for the whole launch
                s_R[2 * LDP + 0] = n_r0; s_R[2 * LDP + 1] = n_r1; s_R[2 * LDP + 2] = n_22;
                s_C[2 * LDP + 0] = n_c0; s_C[2 * LDP + 1] = n_c1; s_C[2 * LDP + 2] = n_22;
            }
        }

        SLAM_STAMP(5);   // predict
        // ---- detections of the group in message order ----
#pragma unroll 1
        for (int l = l0; l < l1; ++l) {
            const int td = opaque(tid);   // keeps per-lane index arithmetic from being hoisted out of the loops
            const int idx = didx_t[l];
            if (idx < 0 || SLAM_DBG(p.dbg & 2)) continue;  // dropped (capacity)
            const float r_m = meas_t[3 * l + 1], b_m = meas_t[3 * l + 2];
            const int ii = 3 + 2 * idx;
            if (idx < M) {
                // ---------------- landmark update, ekf.cpp:110-140 ----------------
                // Three barriers per update.  Everything that is a scalar chain in the reference (Jacobian entries with their
                // float truncations, the innovation, S and its PartialPivLU inverse) is evaluated by wavefront 0 WITHOUT a
                // barrier in between: the eight quotients of H on eight lanes at once, atan2 beside them (independent
                // chains in one instruction stream), the five columns of H P that S needs on five lanes, the results passed
                // between lanes as wave-uniform values (v_readlane).  The other wavefronts join for the O(n) parts.
                const int si = s_slot[ii];
                if (tid < 64) {
                    double H[8], Si[4], nu0, nu1;
                    const bool okS = leader_chain(ii, si, r_m, b_m, H, nu0, nu1, Si);
                    if (lane < 8) {   // broadcast to the other wavefronts through LDS
                        double hv = H[0];
#pragma unroll
                        for (int q = 1; q < 8; ++q) hv = lane == q ? H[q] : hv;
                        s_sc[lane] = hv;
                    }
                    if (lane == 0) {
                        if (!okS) s_misc[6] = 1;
                        s_sc[8] = nu0; s_sc[9] = nu1;
                        s_sc[10] = Si[0]; s_sc[11] = Si[1]; s_sc[12] = Si[2]; s_sc[13] = Si[3];
                    }
                }
                __syncthreads();
                double2* __restrict__ HPu = s_HP + nu * HPW;   // entry c at hpi(c)
                double2* __restrict__ Ku = s_K + nu * LDP;
                {   // every state index: its column of H P, its row of P H^T, K = P H^T S^-1, x_pred += K nu
                    const double h00 = s_sc[0], h01 = s_sc[1], h03 = s_sc[2], h04 = s_sc[3];
                    const double h10 = s_sc[4], h11 = s_sc[5], h12 = -1.0, h13 = s_sc[6], h14 = s_sc[7];
                    const double si0 = s_sc[10], si1 = s_sc[11], si2 = s_sc[12], si3 = s_sc[13];
                    const double nu0 = s_sc[8], nu1 = s_sc[9];
                    const double* Ri = s_R + si * LDP;
                    const double* Rj = s_R + (si + 1) * LDP;
                    const double* Ci = s_C + si * LDP;
                    const double* Cj = s_C + (si + 1) * LDP;
#pragma unroll
                    for (int u = 0; u < (LDP + TPB - 1) / TPB; ++u) {
                        const int c = td + TPB * u;
                        double2 hp = make_double2(0.0, 0.0), kk = make_double2(0.0, 0.0);
                        if (c < na) {
                            const double p0 = s_R[c], p1 = s_R[LDP + c], p2 = s_R[2 * LDP + c], pi = Ri[c], pj = Rj[c];
                            hp.x = ((h00 * p0 + h01 * p1) + h03 * pi) + h04 * pj;
                            hp.y = (((h10 * p0 + h11 * p1) + h12 * p2) + h13 * pi) + h14 * pj;
                            const double q0 = s_C[c], q1 = s_C[LDP + c], q2 = s_C[2 * LDP + c], qi = Ci[c], qj = Cj[c];
                            const double phx = ((q0 * h00 + q1 * h01) + qi * h03) + qj * h04;
                            const double phy = (((q0 * h10 + q1 * h11) + q2 * h12) + qi * h13) + qj * h14;
                            kk.x = phx * si0 + phy * si2;
                            kk.y = phx * si1 + phy * si3;
                            double xv = s_xp[c] + (kk.x * nu0 + kk.y * nu1);
                            if (c == 2) xv = rem2pi(xv);
                            s_xp[c] = xv;
                        }
                        if (c < LDP) { HPu[hpi(c)] = hp; Ku[c] = kk; }
                    }
                }
                __syncthreads();
                // thin copies follow the same downdate  P -= K (H P)
                {
                    constexpr int JW = TPB < 128 ? TPB : 128;   // threads along a thin row; the others take other slots
                    thin_downdate(td % JW, JW, td / JW, (TPB + JW - 1) / JW, nT, na, Ku, HPu);
                }
                nu += 1;
                __syncthreads();
            } else {
                // ---------------- landmark insertion, ekf.cpp:141-173 ----------------
                const int sa = s_slot[ii], sb = sa + 1;
                const int no = na;
                if (tid == 0) {  // leader: G_x, G_z entries and the new landmark position
                    const double phi = s_xp[2] + (double)b_m;
                    double s, c;
                    det_sincos(phi, &s, &c);
                    const double rd = (double)r_m;
                    s_sc[0] = -rd * s; s_sc[1] = rd * c; s_sc[2] = c; s_sc[3] = s;
                    s_sc[4] = s_xp[0] + rd * c; s_sc[5] = s_xp[1] + rd * s;
                }
                __syncthreads();
                const double g02 = s_sc[0], g12 = s_sc[1];
                // new rows G_x P[0:3,:] and new cols P[:,0:3] G_x^T
#pragma unroll 1
                for (int j = td; j < no; j += TPB) {
                    s_R[sa * LDP + j] = s_R[j] + g02 * s_R[2 * LDP + j];
                    s_R[sb * LDP + j] = s_R[LDP + j] + g12 * s_R[2 * LDP + j];
                    s_C[sa * LDP + j] = s_C[j] + s_C[2 * LDP + j] * g02;
                    s_C[sb * LDP + j] = s_C[LDP + j] + s_C[2 * LDP + j] * g12;
                }
                __syncthreads();
                if (tid == 0) {  // corner: (G_x P_vv) G_x^T + (G_z W) G_z^T
                    const double c = s_sc[2], s = s_sc[3];
                    const double gw00 = c * p.W00, gw01 = g02 * p.W11;   // (G_z W) row 0
                    const double gw10 = s * p.W00, gw11 = g12 * p.W11;   // (G_z W) row 1
                    const double* Ra = s_R + sa * LDP;
                    const double* Rb = s_R + sb * LDP;
                    const double v00 = ((Ra[0] + Ra[2] * g02) + gw00 * c) + gw01 * g02;
                    const double v01 = ((Ra[1] + Ra[2] * g12) + gw00 * s) + gw01 * g12;
                    const double v10 = ((Rb[0] + Rb[2] * g02) + gw10 * c) + gw11 * g02;
                    const double v11 = ((Rb[1] + Rb[2] * g12) + gw10 * s) + gw11 * g12;
                    s_R[sa * LDP + no] = v00; s_R[sa * LDP + no + 1] = v01;
                    s_R[sb * LDP + no] = v10; s_R[sb * LDP + no + 1] = v11;
                    s_C[sa * LDP + no] = v00; s_C[sa * LDP + no + 1] = v10;
                    s_C[sb * LDP + no] = v01; s_C[sb * LDP + no + 1] = v11;
                    s_xp[no] = s_sc[4];
                    s_xp[no + 1] = s_sc[5];
                    s_ids[M] = p.id_known ? (int)meas_t[3 * l] : M;
                }
                if (td >= 64 - TS && td < 64) {  // cross entries of the other thin rows / cols
                    const int sl = td - (64 - TS);
                    if (sl < nT && sl != sa && sl != sb) {
                        const int t_s = s_T[sl];
                        if ((unsigned)t_s < (unsigned)no) {
                            s_R[sl * LDP + no] = s_C[sa * LDP + t_s];       // P[t_s][no]
                            s_R[sl * LDP + no + 1] = s_C[sb * LDP + t_s];   // P[t_s][no+1]
                            s_C[sl * LDP + no] = s_R[sa * LDP + t_s];       // P[no][t_s]
                            s_C[sl * LDP + no + 1] = s_R[sb * LDP + t_s];   // P[no+1][t_s]
                        }
                    }
                }
                M += 1;
                na += 2;
                __syncthreads();
            }
        }

        SLAM_STAMP(6);   // detections
        // ---- the last wavefront first closes the books of this step and prepares the next one; it joins the stream
        //      when it is done (chunks are handed out dynamically, so the others simply take more of them) ----
        if (l1 >= k && (tid >> 6) == W - 1) {
            if (p.sim && lane == 0) {  // plotting_node.py:209-212 with the float32 wire format of EKFState.x_v / y_v
                const double* tru = s_tru + (t % SD) * 6 + 3;   // true pose after this timestep
                const double ex = (double)(float)s_xp[0] - tru[0], ey = (double)(float)s_xp[1] - tru[1];
                s_keep[3] = s_keep[3] + sqrt(ex * ex + ey * ey);
            }
            if (t + 1 < T) prestep(t + 1);
        }
        // ---- what goes to HBM now.  Updates are DEFERRED: the group (K, H P of up to KG updates) stays open across
        //      timesteps and P is streamed once per group instead of once per step; the thin rows / columns in LDS are
        //      always current, so nothing on the critical path needs P itself.  The stream runs now if the group cannot
        //      stay open: more groups of this step follow, the step changed the layout (insertions), unknown-id
        //      association (every detection is its own group), or the launch ends.  (A pending group is flushed at the
        //      START of a step that needs HBM to be current: see the pre-flush above.) ----
        const bool more = l1 < k;   // further groups of this step follow
        const bool pass_now = more || !first || nf != n_old || !p.id_known || t + 1 >= T || SLAM_DBG(p.dbg & 16);
        if (pass_now) {
            __syncthreads();   // the thin copies are final for this pass (the prediction's late stores of slot 2 included)
            __builtin_amdgcn_s_setprio(0);
            PassArgs pa;
            pa.lo = 0;
            pa.src = first ? Pin : Pout; pa.dst = Pout; pa.mid = Pmid;
            pa.nf = nf; pa.ldd = ldd; pa.lds = lds; pa.nsrc = nsrc; pa.nu = nu;
            if (first && !more && nf == n_old) {
                if (nu == 0 && !SLAM_DBG(p.dbg & 16)) {   // nothing pending: only the prediction's rows / columns
                    write_vehicle(Pout, nf);
                } else {
                    if (tid == 0) count_pass(s_cnt, 2 * nf * (ldd / VEC), nu);
                    stream_pass(std::integral_constant<int, 1>{}, pa);
                }
            } else if (kWide || (first && !more)) {
                if (tid == 0) count_pass(s_cnt, nsrc * (lds / VEC) + nf * (ldd / VEC), nu);
                stream_pass(std::integral_constant<int, 0>{}, pa);
            } else {
                pa.src = Pin;
                // the fp64 slab between the groups of one fp32-storage step moves 8-byte elements
                if (tid == 0) count_pass(s_cnt, (nsrc * lds * (first ? ESZ : 8) + nf * ldd * (more ? 8 : ESZ)) / 16, nu);
                mid_pass(!first, more, pa);
            }
            nu = 0;
        } else if (!kWide && nu > 0 && tid == 0) {
            s_wend[nu - 1] = 1;   // fp32 storage: P is rounded where this timestep ends
        }
        l0 = l1;
        first = false;
    }
    if (frz_at >= 0) break;
    __syncthreads();
    SLAM_STAMP(7);   // bulk stream
    if (nu == 0 && tid < KG) s_wend[tid] = 0;
