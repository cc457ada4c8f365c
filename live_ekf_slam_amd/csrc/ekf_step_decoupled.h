// ekf_step_decoupled.h — part of the BODY of ekf_step_kernel (ekf_kernel_impl.h includes it inside the kernel function; round 6: the 1 840-line kernel split into its
// parts, pure moves - every object file byte-identical).  The DECOUPLED steady-state loop: wavefront 0 runs the thin phases of consecutive timesteps and publishes updates in the ring, the other wavefronts stream them into P.
// Lambdas and statements here capture the kernel's locals (p, tid, lane, the LDS arrays ...): not a stand-alone header.  DESIGN.md 4.1.

    // =====================================================================================================================
    // DECOUPLED STEADY-STATE LOOP.  As long as the steps ahead neither insert landmarks nor freeze, overflow or exceed KG
    // detections, the workgroup leaves the barrier-synchronised step above: wavefront 0 (CONTROL) runs every thin phase of
    // consecutive timesteps by itself - pre-step, group formation, prediction, per detection the scalar chain, K / H P, the
    // state update and the downdate of the thin copies, all wave-synchronous, no workgroup barrier - and publishes each
    // update's K / H P in a ring of KG slots; the other wavefronts (STREAMERS) apply the published updates to P in passes
    // of up to KG updates, concurrently.  The thin copies in LDS are always current, so the control wavefront never waits for
    // P except when a landmark comes into view whose row / column it must gather: then it has the streamers drain the ring
    // first.  Rows / columns 0, 1 and (2,2) of P in HBM are not maintained inside the loop (a gathered row takes those
    // entries from the resident vehicle columns); they are written when the loop ends.  Every element of P sees the same
    // operations in the same order as in the synchronised path, so the results are bit-identical.
    // =====================================================================================================================
    if constexpr (MULTI && W >= 2) {
        auto fastable = [&](int tq) -> bool {   // step tq (its pre-step results are in the parity buffers) can run decoupled
            const int* nx = s_next + 4 * (tq & 1);
            // fp32 storage rounds P once per timestep, so a pass may only end where a step ends (s_wend): the updates of a step
            // must fit the ring, or the control wavefront waits for a slot that only a pass could free while no pass can be cut
            // (the several-groups-per-step loop of round 3 let steps of up to 2 KP detections in: a deadlock the watchdog turned
            // into SLAM_INST_WATCHDOG, found by tools/gpu_soak_ekf.py).  fp64 passes may end anywhere.
            constexpr int kStepMax = sizeof(ST) == 8 ? KLOOP : (KLOOP < KG ? KLOOP : KG);
            return nx[0] <= kStepMax && nx[1] == 0 && nx[2] == 0 && nx[3] == 0;
        };
        const bool fast_ok = p.id_known && p.meas_out == nullptr && fastable(t) &&
                             !SLAM_DBG(p.dbg & (2 | 16 | 64));
        if (fast_ok) {
            const int n = na, ldn = ekf_ld(n, ESZ);
            ST* const Pbuf = Pcur;
            constexpr int kFirstStreamer = 1;
            constexpr int NS = W - kFirstStreamer;    // streamers
            constexpr bool kGen = W >= 2;             // the last streamer also runs the measurement generator ahead of the filter (with two
                                                      // wavefronts that is the pass leader: it generates while no pass is due)
            if (tid == 0) {
                s_ring[0] = nu; s_ring[1] = 0; s_ring[2] = 0; s_ring[3] = 0; s_ring[4] = t; s_ring[5] = 0; s_ring[6] = 0; s_ring[7] = 0;
                s_pass[0] = 0; s_pass[1] = 0; s_pass[2] = 0; s_pass[3] = 0;
            }
            __syncthreads();
            auto ld_i = [](int* q) -> int { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
            auto st_i = [](int* q, int v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
            // WATCHDOG.  Every polling loop below counts its polls; one that exceeds the budget (~0.1 s: thousands of passes)
            // raises s_ring[2], every other loop sees that and leaves, and the instance is flagged SLAM_INST_WATCHDOG and frozen
            // instead of hanging the GPU.  tests/test_ring_protocol_model.py checks the protocol itself exhaustively; this is the
            // net under it (one tuning variant did deadlock in round 2).  p.dbg & 128 (tests only) makes the pass leader lose its
            // `applied` update so that the control wavefront starves.
            constexpr int kSpinBudget = 1 << 21;
            auto spin_over = [&](int& spins) -> bool {
                spins += 1;
                if (spins > kSpinBudget) st_i(&s_ring[2], 1);
                return ld_i(&s_ring[2]) != 0;
            };

            bool is_streamer = true;
            if (tid < 64) {
                is_streamer = false;
                // ------------------------------------------------ CONTROL ------------------------------------------------
                __builtin_amdgcn_s_setprio(3);
                int tt = t;
                int pub = nu;
                int fl_or = 0;
                bool first_it = true;
#pragma unroll 1
                for (;;) {
                    const int pq = tt & 1;
                    const float* const meas_q = s_meas + (tt % SD) * 3 * KCAP;
                    const int* const didx_q = s_didx + pq * KCAP;
                    const int kq = s_next[4 * pq];
                    if (lane == 0) st_i(&s_sim[1], tt);   // ring slots of the timesteps before tt are free for the generator
                    if (!first_it && lane == 0) s_kh[kq < 7 ? kq : 7] += 1;
                    first_it = false;
                    if (lane < 3) s_xp[lane] = s_ps[10 * pq + lane];
                    SLAM_STAMP(16);  // loop overhead
                    int lastu = -1;
                    if constexpr (!kWide) {
                        const bool isupd = lane < kq && didx_q[lane] >= 0;
                        const unsigned long long um = __ballot(isupd);
                        lastu = um ? 63 - __clzll((long long)um) : -1;
                    }
                    // the detections of the timestep in groups of at most KP (one landmark slot pair each); nearly always one group
                    int l0q = 0, l1q, nTq;
#pragma unroll 1
                    do {
                    const int needg = form_known(didx_q, kq, l0q, KP, n, l1q, nTq);
                    SLAM_STAMP(17);  // group formation
                    const bool veh = s_need[0] == 1;   // first step of the launch: the vehicle rows / columns are still in HBM only
                    if (needg || veh) {
                        // A landmark comes into view: its row / column comes from HBM, which holds the updates the streamers
                        // have applied so far (`app`); the ones still pending are in the ring slots, so the gathered copy is
                        // brought up to date here, with the operations the stream will apply to P.  Only a pass in flight
                        // must end first (P is half-updated meanwhile), and no new one may start during the gather.
                        if (lane == 0) st_i(&s_ring[6], 1);                          // hold
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                        for (int sp = 0; ld_i(&s_ring[7]) && !spin_over(sp);) __builtin_amdgcn_s_sleep(1);   // pass in flight
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        const int app = ld_i(&s_ring[1]);
                        // two slots (the row and column pair of one landmark) per trip: their row and column loads are issued together, one HBM round
                        // trip per 64 state indices instead of one per slot and 64 indices
                        int sl = 0;
#pragma unroll 1
                        while (sl < nTq) {
                            constexpr int GB = 2;
                            int ss[GB], ts[GB], nb = 0;
#pragma unroll
                            for (int g = 0; g < GB; ++g) { ss[g] = 0; ts[g] = 0; }
#pragma unroll 1
                            while (sl < nTq && nb < GB) {
                                if (s_need[sl] == 1) {
#pragma unroll
                                    for (int g = 0; g < GB; ++g)
                                        if (g == nb) { ss[g] = sl; ts[g] = s_T[sl]; }
                                    nb += 1;
                                }
                                sl += 1;
                            }
                            if (nb == 0) break;
                            if (lane == 0) count_other(s_cnt, 2 * nb * n);
#pragma unroll 1
                            for (int j = lane; j < LDP; j += 64) {
                                double rv[GB], cv[GB];
                                const int jc = j < n ? j : 0;
#pragma unroll
                                for (int g = 0; g < GB; ++g) {
                                    rv[g] = (double)Pbuf[(size_t)ts[g] * ldn + jc];   // P[t_s][j]
                                    cv[g] = (double)Pbuf[(size_t)jc * ldn + ts[g]];   // P[j][t_s]
                                }
#pragma unroll 1
                                for (int u = app; u < pub; ++u) {
                                    const int us = u % KG;
                                    const double2* Ku = s_K + us * LDP;
                                    const double2* HPu = s_HP + us * HPW;
                                    const double2 kj = Ku[jc], hj = HPu[hpi(jc)];
                                    bool we = false;
                                    if constexpr (!kWide) we = s_wend[us] != 0;
#pragma unroll
                                    for (int g = 0; g < GB; ++g) {
                                        const double2 kt = Ku[ts[g]], ht = HPu[hpi(ts[g])];
                                        rv[g] = rv[g] - (kt.x * hj.x + kt.y * hj.y);
                                        cv[g] = cv[g] - (kj.x * ht.x + kj.y * ht.y);
                                        if constexpr (!kWide) {
                                            if (we) { rv[g] = (double)(ST)rv[g]; cv[g] = (double)(ST)cv[g]; }   // end of a timestep: storage rounding
                                        }
                                    }
                                }
#pragma unroll
                                for (int g = 0; g < GB; ++g)
                                    if (g < nb) {
                                        s_R[ss[g] * LDP + j] = j < n ? rv[g] : 0.0;
                                        s_C[ss[g] * LDP + j] = j < n ? cv[g] : 0.0;
                                    }
                            }
                            // entries against the vehicle states come from the resident vehicle columns / rows (HBM does not
                            // have the predictions of the steps since the loop began)
#pragma unroll
                            for (int g = 0; g < GB; ++g)
                                if (g < nb && ss[g] >= 3 && lane < 3) {
                                    s_R[ss[g] * LDP + lane] = s_C[lane * LDP + ts[g]];   // P[t_s][c], c < 3
                                    s_C[ss[g] * LDP + lane] = s_R[lane * LDP + ts[g]];   // P[r][t_s], r < 3
                                }
                        }
                        if (lane == 0) st_i(&s_ring[6], 0);
                    }
                    if (lane < TS) s_need[lane] = 0;
                    SLAM_STAMP(18);  // flush wait + gather
                    // ---- prediction on the thin copies (ekf.cpp:41-61), one wavefront: see the synchronised path ----
                    if (l0q == 0) {
                        const double* const ps = s_ps + 10 * pq;
                        const double* const r2o = s_R + 2 * LDP;
                        const double* const c2o = s_C + 2 * LDP;
                        const double fa = ps[3], fb = ps[4];
                        const double p22 = r2o[2];
                        auto predicted = [&](double tv, int r, int cc) -> double {
                            const double f_r = r == 0 ? fa : fb;
                            if (r < 2) tv = tv + f_r * r2o[cc];
                            if (cc < 2) {
                                double a2 = c2o[r];
                                if (r < 2) a2 = a2 + f_r * p22;
                                tv = tv + a2 * (cc == 0 ? fa : fb);
                            }
                            if (r < 2 && cc < 2) tv = tv + ps[5 + 2 * r + cc];
                            if (r == 2 && cc == 2) tv = tv + p.V11;
                            return tv;
                        };
                        double n_r0 = 0.0, n_r1 = 0.0, n_c0 = 0.0, n_c1 = 0.0, n_22 = 0.0;
                        const int t_s = (lane >= 2 && lane < nTq) ? s_T[lane] : -1;
                        const bool thin_l = (unsigned)t_s < (unsigned)n;
                        if (thin_l) {   // entries 0, 1 (+ (2,2)) of the other thin rows / cols: computed BEFORE rows / cols 0, 1 change
                            n_r0 = predicted(s_R[lane * LDP + 0], t_s, 0);
                            n_r1 = predicted(s_R[lane * LDP + 1], t_s, 1);
                            n_c0 = predicted(s_C[lane * LDP + 0], 0, t_s);
                            n_c1 = predicted(s_C[lane * LDP + 1], 1, t_s);
                            if (lane == 2) n_22 = predicted(p22, 2, 2);
                        }
                        // rows / cols 0, 1 at state index j >= 2 take one term each (what `predicted` reduces to there):
                        // P[0][j] += F02 P[2][j], P[1][j] += F12 P[2][j], P[j][0] += P[j][2] F02, P[j][1] += P[j][2] F12
#pragma unroll
                        for (int u = 0; u < (LDP + 63) / 64; ++u) {
                            const int j = lane + 64 * u;
                            if (j >= 2 && j < n) {
                                const double r2 = r2o[j], c2 = c2o[j];
                                s_R[j] = s_R[j] + fa * r2;
                                s_R[LDP + j] = s_R[LDP + j] + fb * r2;
                                s_C[j] = s_C[j] + c2 * fa;
                                s_C[LDP + j] = s_C[LDP + j] + c2 * fb;
                            }
                        }
                        if (lane < 2) {   // the 2 x 2 corner (all terms)
                            const int j = lane;
                            const double v00 = predicted(s_R[j], 0, j), v10 = predicted(s_R[LDP + j], 1, j);
                            const double w00 = predicted(s_C[j], j, 0), w10 = predicted(s_C[LDP + j], j, 1);
                            s_R[j] = v00; s_R[LDP + j] = v10; s_C[j] = w00; s_C[LDP + j] = w10;
                        }
                        if (thin_l) {   // late stores: every operand above was read before
                            s_R[lane * LDP + 0] = n_r0; s_R[lane * LDP + 1] = n_r1;
                            s_C[lane * LDP + 0] = n_c0; s_C[lane * LDP + 1] = n_c1;
                            if (lane == 2) { s_R[2 * LDP + 2] = n_22; s_C[2 * LDP + 2] = n_22; }
                        }
                    }
                    SLAM_STAMP(19);  // prediction
                    // ---- detections of the group in message order (all of them updates: the step inserts nothing) ----
#pragma unroll 1
                    for (int l = l0q; l < l1q; ++l) {
                        const int idx = didx_q[l];
                        if (idx < 0) continue;
                        const float r_m = meas_q[3 * l + 1], b_m = meas_q[3 * l + 2];
                        const int ii = 3 + 2 * idx;
                        const int si = s_slot[ii];
                        double H[8], Si[4], nu0, nu1;
                        if (!leader_chain(ii, si, r_m, b_m, H, nu0, nu1, Si)) fl_or |= SLAM_INST_S_SINGULAR;
                        SLAM_STAMP(20);  // scalar chain of the update
                        for (int sp = 0; pub - ld_i(&s_ring[1]) >= KG && !spin_over(sp);) __builtin_amdgcn_s_sleep(SLAM_SLEEP_RING);   // a free slot in the ring
                        if (ld_i(&s_ring[2])) break;   // watchdog
                        SLAM_STAMP(21);  // waiting for a ring slot
                        const int slot = pub % KG;
                        double2* __restrict__ HPu = s_HP + slot * HPW;
                        double2* __restrict__ Ku = s_K + slot * LDP;
                        double2 kreg[NU], hreg[NU];   // this lane's K[j], (H P)[j], j = lane + 64 u: the thin downdate takes them from here
#pragma unroll
                        for (int u = 0; u < NU; ++u) { kreg[u] = make_double2(0.0, 0.0); hreg[u] = make_double2(0.0, 0.0); }
                        if (!SLAM_DBG(p.dbg & 512)) {   // (ablation 512: timing without H P / K / x)
                            const double h00 = H[0], h01 = H[1], h03 = H[2], h04 = H[3], h10 = H[4], h11 = H[5], h12 = -1.0, h13 = H[6], h14 = H[7];
                            const double* Ri = s_R + si * LDP;
                            const double* Rj = s_R + (si + 1) * LDP;
                            const double* Ci = s_C + si * LDP;
                            const double* Cj = s_C + (si + 1) * LDP;
#pragma unroll
                            for (int u = 0; u < (LDP + 63) / 64; ++u) {
                                const int c = lane + 64 * u;
                                double2 hp = make_double2(0.0, 0.0), kk = make_double2(0.0, 0.0);
                                if (c < n) {
                                    const double p0 = s_R[c], p1 = s_R[LDP + c], p2 = s_R[2 * LDP + c], pi = Ri[c], pj = Rj[c];
                                    hp.x = ((h00 * p0 + h01 * p1) + h03 * pi) + h04 * pj;
                                    hp.y = (((h10 * p0 + h11 * p1) + h12 * p2) + h13 * pi) + h14 * pj;
                                    const double q0 = s_C[c], q1 = s_C[LDP + c], q2 = s_C[2 * LDP + c], qi = Ci[c], qj = Cj[c];
                                    const double phx = ((q0 * h00 + q1 * h01) + qi * h03) + qj * h04;
                                    const double phy = (((q0 * h10 + q1 * h11) + q2 * h12) + qi * h13) + qj * h14;
                                    kk.x = phx * Si[0] + phy * Si[2];
                                    kk.y = phx * Si[1] + phy * Si[3];
                                    double xv = s_xp[c] + (kk.x * nu0 + kk.y * nu1);
                                    if (c == 2) xv = rem2pi(xv);
                                    s_xp[c] = xv;
                                }
                                if (c < LDP) { HPu[hpi(c)] = hp; Ku[c] = kk; }
                                kreg[u] = kk; hreg[u] = hp;
                            }
                        }
                        if (!kWide && lane == 0) s_wend[slot] = (l == lastu) ? 1 : 0;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // K / H P of the slot are in LDS before it is published
                        pub += 1;
                        if (lane == 0) st_i(&s_ring[0], pub);
                        SLAM_STAMP(22);  // H P, K, x_pred
                        // thin copies follow the same downdate  P -= K (H P)
                        if (!SLAM_DBG(p.dbg & 256)) {   // (ablation 256: timing without the thin downdates)
#if SLAM_CTRL_ILP
                            thin_downdate_ctl(nTq, n, Ku, HPu, kreg, hreg);
#else
                            thin_downdate(lane, 64, 0, 1, nTq, n, Ku, HPu);
#endif
                        }
                    }
                    l0q = l1q;
                    } while (l0q < kq && !ld_i(&s_ring[2]));
                    if (ld_i(&s_ring[2])) break;   // watchdog fired: the instance is frozen below
                    SLAM_STAMP(23);  // thin downdates (+ loop)
                    // ---- end of the step: error statistic, x_t = x_pred (ekf.cpp:176), storage rounding ----
                    if (p.sim && lane == 0) {   // plotting_node.py:209-212 with the float32 wire format of EKFState.x_v / y_v
                        const double* tru = s_tru + (tt % SD) * 6 + 3;   // true pose after this timestep
                        const double ex = (double)(float)s_xp[0] - tru[0], ey = (double)(float)s_xp[1] - tru[1];
                        s_keep[3] = s_keep[3] + sqrt(ex * ex + ey * ey);
                    }
#pragma unroll 1
                    for (int i = lane; i < n; i += 64) {
                        const ST sv = (ST)s_xp[i];
                        s_xt[i] = (double)sv;
                        s_xp[i] = (double)sv;
                        const unsigned h0 = hi_abs((double)sv);
                        hiacc = hiacc > h0 ? hiacc : h0;
                    }
                    if constexpr (!kWide) {   // resident thin rows / cols carry the storage rounding of every step
                        // four elements of each per trip: the reads of a trip issue together (one at a time this loop was a dozen
                        // dependent LDS round trips per step)
                        const int nel = nTq * LDP;
#pragma unroll 1
                        for (int i0 = lane; i0 < nel; i0 += 256) {
                            double rv[4], cv[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int i = i0 + 64 * u < nel ? i0 + 64 * u : i0;
                                rv[u] = s_R[i]; cv[u] = s_C[i];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int i = i0 + 64 * u;
                                if (i < nel) { s_R[i] = (double)(ST)rv[u]; s_C[i] = (double)(ST)cv[u]; }
                            }
                        }
                    }
                    if ((p.dbg & 32) && p.prof != nullptr && lane == 0 && tt < kEkfProfSlots)
                        p.prof[(size_t)blockIdx.x * kEkfProfSlots + tt] = (wall_clock64() << 4) | (unsigned long long)(kq < 15 ? kq : 15);
                    tt += 1;
                    SLAM_STAMP(24);  // end of step
                    if (tt >= T) break;
                    if constexpr (kGen) {   // the measurements of timestep tt come from the generator wavefront
                        for (int sp = 0; ld_i(&s_sim[0]) <= tt && !spin_over(sp);) __builtin_amdgcn_s_sleep(1);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    SLAM_STAMP(26);  // waiting for the generator wavefront
                    prestep(tt);
                    SLAM_STAMP(25);  // pre-step of the next timestep
                    if (!fastable(tt)) break;   // that step goes through the synchronised path
                }
                if (lane == 0) {
                    s_ring[4] = tt;
                    s_ring[5] = fl_or;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    st_i(&s_ring[3], 1);   // exit: the streamers drain the ring and leave
                    if (ld_i(&s_ring[2])) st_i(&s_ring[6], 0);   // watchdog: never leave a hold behind
                }
                __builtin_amdgcn_s_setprio(0);
            }
            if (is_streamer) {
                // ------------------------------------------------ STREAMERS ------------------------------------------------
                __builtin_amdgcn_s_setprio(0);
                const bool leader = (tid >> 6) == kFirstStreamer;
                // never more than the ring holds: with KG < SLAM_PASS_MIN the control wavefront would wait for a slot and the
                // leader for updates that cannot be published (a KG = 3 sweep variant hung the GPU that way)
#ifdef SLAM_PASS_MIN_FORCE
                constexpr int kPassMinCfg = SLAM_PASS_MIN_FORCE;
#else
                // fp64: a pass starts at SLAM_PASS_MIN = 4 pending updates, and from five ring slots on at KG - 1 (one slot stays free)
#ifdef SLAM_PASS_MIN_F32
                constexpr int kPassMinF32 = SLAM_PASS_MIN_F32;
#else
                constexpr int kPassMinF32 = KG > 4 ? KG - 2 : 3;
#endif
                constexpr int kPassMinCfg = kWide ? (KG > SLAM_PASS_MIN + 1 ? KG - 1 : SLAM_PASS_MIN) : kPassMinF32;
#endif
                constexpr int kPassMin = kPassMinCfg < KG ? kPassMinCfg : KG;
                int seen = 0;   // passes this wavefront has taken part in
                int sp = 0;     // polls since this wavefront last made progress (watchdog)
                // (Tried in round 3 and refused: a read-only sweep of P by the idle streamers at the first entry of a launch, so that the
                // first pass finds the matrix in L2 / the Infinity Cache: 60.6 vs 61.1 M steps/s on the 20-step window.  What a launch
                // pays for its cold matrices is their bytes, not the latency of the first pass.)
#pragma unroll 1
                for (;;) {
                    if (leader) {
                        int app, pend;
                        bool stop = false;
#pragma unroll 1
                        for (;;) {
                            if (spin_over(sp)) { stop = true; break; }   // watchdog: tell the other streamers to leave
                            app = ld_i(&s_ring[1]);
                            pend = ld_i(&s_ring[0]) - app;
                            const int ex = ld_i(&s_ring[3]);
                            if (pend > 0 && (pend >= kPassMin || ex) && !ld_i(&s_ring[6])) break;
                            if (ex && pend == 0) {   // re-read: an update published just before the exit flag
                                if (ld_i(&s_ring[0]) - app == 0) { stop = true; break; }
                                continue;
                            }
                            if constexpr (W == 2) {   // the only streamer: no pass is due, so generate a timestep ahead if the ring has room
                                const int ts = ld_i(&s_sim[0]);
                                if (ts < T && ts < ld_i(&s_sim[1]) + SD && !ex) {
                                    simgen(ts);
                                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                    if (lane == 0) st_i(&s_sim[0], ts + 1);
                                    continue;
                                }
                            }
                            __builtin_amdgcn_s_sleep(SLAM_SLEEP_LEADER);
                        }
                        int cnt = pend < KG ? pend : KG;
                        if constexpr (!kWide) {
                            // fp32 storage rounds P once per timestep: a pass must not end inside a step, or the store would round
                            // an intermediate result.  Take the longest prefix that ends where a step ends (there is one whenever
                            // the control wavefront is waiting for a slot, because a step has at most KG updates).
                            if (!stop) {
                                while (cnt > 0 && !ld_i(&s_wend[(app + cnt - 1) % KG])) cnt -= 1;
                                if (cnt == 0) { __builtin_amdgcn_s_sleep(1); continue; }
                            }
                        }
                        if (!stop) {   // claim the pass; back off if the control wavefront is gathering (it waits for a claimed pass)
                            if (lane == 0) st_i(&s_ring[7], 1);
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                            if (ld_i(&s_ring[6])) {
                                if (lane == 0) st_i(&s_ring[7], 0);
                                __builtin_amdgcn_s_sleep(1);
                                continue;
                            }
                        }
                        if (lane == 0) {
                            s_pass[1] = app;
                            s_pass[2] = stop ? -1 : cnt;
                            s_pass[3] = 0;
                            s_chunk = 0;
                            if (!stop) count_pass(s_cnt, 2 * n * (ldn / VEC), cnt);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            st_i(&s_pass[0], seen + 1);
                        }
                    }
                    while (ld_i(&s_pass[0]) <= seen && !spin_over(sp)) {
                        if constexpr (kGen && W >= 3) {
                            if ((tid >> 6) == W - 1) {   // between passes: run the measurement generator ahead of the filter
                                const int ts = ld_i(&s_sim[0]);
                                if (ts < T && ts < ld_i(&s_sim[1]) + SD && !ld_i(&s_ring[3])) {
                                    simgen(ts);
                                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                                    if (lane == 0) st_i(&s_sim[0], ts + 1);
                                    continue;
                                }
                            }
                        }
                        __builtin_amdgcn_s_sleep(SLAM_SLEEP_PASS);
                    }
                    seen += 1;
                    sp = 0;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const int lo = ld_i(&s_pass[1]), cnt = ld_i(&s_pass[2]);
                    if (cnt < 0 || ld_i(&s_ring[2])) break;
                    PassArgs pa;
                    pa.src = Pbuf; pa.dst = Pbuf; pa.mid = nullptr;
                    pa.nf = n; pa.ldd = ldn; pa.lds = ldn; pa.nsrc = n; pa.nu = cnt; pa.lo = lo;
                    stream_pass(std::integral_constant<int, 2>{}, pa);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wavefront's stores of the pass have landed
                    if (lane == 0) atomicAdd(&s_pass[3], 1);
                    if (leader) {
                        while (ld_i(&s_pass[3]) < NS && !spin_over(sp)) __builtin_amdgcn_s_sleep(1);
                        sp = 0;
                        if (lane == 0) {
                            if (!(p.dbg & 128)) st_i(&s_ring[1], lo + cnt);   // the ring slots are free, P holds these updates
                            st_i(&s_ring[7], 0);                               // (dbg & 128, tests only: lose the update -> the watchdog must fire)
                        }
                    }
                }
            }
            __syncthreads();
            if (s_ring[2]) {   // the watchdog fired: P is half-updated; flag and freeze the instance (later launches skip it)
                wd_fired = true;
                break;
            }
            // back to the synchronised path: everything published is in P; write what the loop left aside
            nu = 0;
            flags |= s_ring[5];
            const int t_next = s_ring[4];
            write_vehicle(Pbuf, n);
            if (tid < KG) s_wend[tid] = 0;
            if (__syncthreads_or(hiacc >= 0x7ff00000u)) flags |= SLAM_INST_NONFINITE;
            t = t_next - 1;
            continue;
        }
    }
