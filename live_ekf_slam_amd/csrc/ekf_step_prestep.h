// ekf_step_prestep.h — part of the BODY of ekf_step_kernel (ekf_kernel_impl.h includes it inside the kernel function; round 6: the 1 840-line kernel split into its
// parts, pure moves - every object file byte-identical).  Write-back of an instance (`finish`), the measurement generator call (`simgen`) and the PRE-STEP of a timestep: motion scalars, association, bookkeeping of the next step (`prestep`).
// Lambdas and statements here capture the kernel's locals (p, tid, lane, the LDS arrays ...): not a stand-alone header.  DESIGN.md 4.1.

    // state -> HBM at the end of the launch (or when the instance freezes): x_t lives in s_xt
    // `pre`: the instance freezes in its PRE-step state (x, P, timestep, error sum and the true pose alike)
    auto finish = [&](int steps_done, int Mf, int fl, bool pre) {
        const int nfin = 3 + 2 * Mf;
        for (int i = tid; i < nfin; i += TPB) xb[i] = (ST)s_xt[i];
        if (Mf != M_init) {
            for (int i = tid; i < Mf; i += TPB) p.ids[(size_t)b * p.L_max + i] = s_ids[i];
        }
        if (tid == 0) {
            p.M[b] = Mf;
            p.flags[b] = fl;
            p.timestep[b] = ts0 + steps_done;
            if (p.sim) p.err_sum[b] = s_keep[3];
        }
        // true pose: before the frozen step, or after the last step of the launch (the generator never runs past it)
        if (p.sim && tid < 3) {
            const int tq = pre ? steps_done : steps_done - 1;
            p.truth[3 * (size_t)b + tid] = steps_done == 0 && !pre ? s_keep[tid] : s_tru[(tq % SD) * 6 + (pre ? 0 : 3) + tid];
        }
        if (p.khist != nullptr && tid < 8 && s_kh[tid] != 0) atomicAdd(&p.khist[tid], (unsigned long long)s_kh[tid]);
        if (p.khist != nullptr && tid >= 8 && tid < 12) {   // slam_traffic_counters: bytes of the P stream, other bytes, passes, updates
            const unsigned long long v = s_cnt[tid - 8] + (tid == 9 ? (unsigned long long)((nfin + 8) * ESZ / 8) : 0ull);
            const unsigned long long unit = tid == 8 ? 16ull : (tid == 9 ? 8ull : 1ull);
            if (v != 0ull) atomicAdd(&p.khist[kEkfTrafficSlot + tid - 8], v * unit);
        }
    };

    int M = M_init;
    int na = n_init;     // active dimension
    int nu = 0;          // updates of the open group: K / (H P) slots 0 .. nu-1 are pending, P in HBM does not have them yet
    unsigned hiacc = 0u; // non-finite detector (max of |hi word|)

    // Everything of timestep tn that does not depend on P, executed by ONE wavefront: the measurement generator
    // (sim_node.py:209-250), the known-id association of the whole message (ekf.cpp:99-108; lane l <-> detection l)
    // and the vehicle part of the prediction (ekf.cpp:41-59).  For tn > first step of the launch it runs inside the
    // bulk stream of step tn-1 (the other wavefronts keep streaming), so its latency chain is off the critical path.
    // Reads x_{tn} from s_xp (final x_pred of step tn-1), the current M / s_ids; writes the parity-tn buffers.
    // simgen(tn): the measurement generator for timestep tn (ONE wavefront) into ring slot tn % SD; advances the true pose.
    auto simgen = [&](int tn) {
        const int sq = tn % SD;
        if (!p.sim) {
            // EXT mode: the message of timestep tn comes from the caller's queue in device memory,
            // meas_in[tn][b][k_stride][3] / meas_count_in[tn][b] (one timestep per launch: tn = 0)
            int kk = p.meas_count_in[(size_t)tn * p.B + b];
            kk = kk < p.k_stride_in ? kk : p.k_stride_in;
            kk = kk < 0 ? 0 : kk;
            const int kc = kk < KCAP ? kk : KCAP;
            const float* src = p.meas_in + ((size_t)tn * p.B + b) * p.k_stride_in * 3;
            for (int i = lane; i < 3 * kc; i += 64) s_meas[sq * 3 * KCAP + i] = src[i];
            if (lane == 0) s_kraw[sq] = kk;
            return;
        }
        const float fwd_n = MULTI ? p.cmds[2 * tn] : p.fwd;
        const float ang_n = MULTI ? p.cmds[2 * tn + 1] : p.ang;
        double tx = s_keep[0], ty = s_keep[1], tth = s_keep[2];
        if (lane == 0) { s_tru[sq * 6 + 0] = tx; s_tru[sq * 6 + 1] = ty; s_tru[sq * 6 + 2] = tth; }
        const double lmx0 = lane < p.L ? p.map[2 * lane] : 0.0, lmy0 = lane < p.L ? p.map[2 * lane + 1] : 0.0;
        const int kr = sim_wave<KCAP, false>(p, b, lane, fwd_n, ang_n, p.step + (uint32_t)tn, tx, ty, tth, lmx0, lmy0,
                                             s_meas + sq * 3 * KCAP);   // the true pose goes to HBM in finish()
        if (lane == 0) {
            s_keep[0] = tx; s_keep[1] = ty; s_keep[2] = tth;
            s_tru[sq * 6 + 3] = tx; s_tru[sq * 6 + 4] = ty; s_tru[sq * 6 + 5] = tth;
            s_kraw[sq] = kr;
        }
    };
    auto prestep = [&](int tn) {
        const double* const xv = s_xp;   // the vehicle's x_pred of the previous step
        const int qb = tn & 1;
        float* meas = s_meas + (tn % SD) * 3 * KCAP;
        int* didx = s_didx + qb * KCAP;
        int* nx = s_next + 4 * qb;
        double* ps = s_ps + 10 * qb;
        const float fwd_n = MULTI ? p.cmds[2 * tn] : p.fwd;
        const float ang_n = MULTI ? p.cmds[2 * tn + 1] : p.ang;
        if (s_sim[0] <= tn) {   // not produced ahead of time (the decoupled loop's generator wavefront does that)
            simgen(tn);
            if (lane == 0) s_sim[0] = tn + 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int kraw = s_kraw[tn % SD];
        {   // x_pred of the vehicle (ekf.cpp:56-59) and the scalars of F_x, F_v V F_v^T (ekf.cpp:41-55)
            const double x0 = (double)(ST)xv[0], x1 = (double)(ST)xv[1], th = (double)(ST)xv[2];
            double sn, cs;
            det_sincos(th, &sn, &cs);
            const float dd = fwd_n + p.v_d;
            const double cv = cs * p.V00, sv = sn * p.V00;
            if (lane == 0) {
                nx[0] = kraw;
                ps[0] = x0 + (double)dd * cs;
                ps[1] = x1 + (double)dd * sn;
                ps[2] = rem2pi((th + (double)ang_n) + (double)p.v_th);
                ps[3] = (double)(-1 * fwd_n) * sn;  // F_x(0,2)
                ps[4] = (double)fwd_n * cs;         // F_x(1,2)
                ps[5] = cv * cs; ps[6] = cv * sn; ps[7] = sv * cs; ps[8] = sv * sn;
            }
        }
#ifdef SLAM_EXP_ASSOC_REP   // timing experiment (round 5): the association SLAM_EXP_ASSOC_REP extra times - what it costs is what moving it to the
        // generator wavefront could save (same results: it rewrites the same values)
#pragma unroll 1
        for (int rep_ = 0; rep_ <= SLAM_EXP_ASSOC_REP; ++rep_)
#endif
        if (p.id_known) {
            const int kn = kraw < KCAP ? kraw : KCAP;
            // lanes scan lm_IDs in parallel for each detection (first match wins, ekf.cpp:102-107); lane l % 64 then keeps the
            // result of detection l.  The message is walked 64 detections at a time (ekf.cpp:73 loops over any number of them).
            // What the reference's loop does with a NEW id (ekf.cpp:99-108,141-173), per detection in message order: the first
            // occurrence is inserted while there is room, else skipped (no capacity there; here SLAM_INST_CAPACITY); a LATER
            // occurrence of an id this message inserted is found among the pushed ids and indexes x_t out of range (ekf.cpp:115 ->
            // eigen_assert -> exception, filter.h:5: the reference dies, we freeze in the pre-step state); a later occurrence of a
            // SKIPPED id is skipped again.  So: the first `room` distinct new ids are inserted in order of first occurrence, the
            // step freezes at the first repeat of one of those, and the capacity flag is raised by a skip BEFORE that point only.
            // (Until round 3 any repeated new id froze the instance and the capacity flag ignored the order: found by
            // tools/gpu_soak_adversarial.py on messages no AprilTag front-end sends.)
            const int room = (p.L_max < LMAX ? p.L_max : LMAX) - M;
            int nins = 0;                   // insertions of this message
            bool frz = false, capf = false;
            if (kn <= 64 && M <= 64) {
                // the common case in registers: lane l holds the id of detection l and lm_IDs[l]; the id of detection l reaches the
                // others by v_readlane, a match is one ballot (one LDS round trip for the whole message instead of two per detection)
                const int myid = lane < kn ? (int)meas[3 * lane] : -1;
                const int sid = lane < M ? s_ids[lane] : 0;
                int idx = -1, firstl = lane;
                bool isnew = false;
#pragma unroll 1
                for (int l = 0; l < kn; ++l) {
                    const int id = __builtin_amdgcn_readlane(myid, l);
                    const unsigned long long m = __ballot(lane < M && sid == id);        // first match wins (ekf.cpp:102-107); any int is an id
                    const unsigned long long e = __ballot(lane < l && myid == id);       // earlier detections of this message with the id (l < kn)
                    if (lane == l) { idx = m ? __ffsll((long long)m) - 1 : -1; isnew = m == 0ull; firstl = e ? __ffsll((long long)e) - 1 : l; }
                }
                const bool isfirst = isnew && firstl == lane;
                const unsigned long long fmask = __ballot(isfirst);
                const int rankf = __popcll(fmask & ((1ull << firstl) - 1ull));           // rank of my id's first occurrence among the new ids
                const bool insf = rankf < room;
                const unsigned long long fz = __ballot(isnew && !isfirst && insf);
                const unsigned long long cm = __ballot(isnew && !insf);
                const unsigned long long before = fz ? ((1ull << (__ffsll((long long)fz) - 1)) - 1ull) : ~0ull;
                frz = fz != 0ull;
                capf = (cm & before) != 0ull;
                if (isnew) idx = (isfirst && insf) ? M + rankf : -1;
                if (lane < kn) didx[lane] = idx;
                nins = __popcll(fmask);
                nins = nins < room ? nins : (room > 0 ? room : 0);
            } else {
                // long messages / large maps: one detection at a time, the wavefront scans lm_IDs and the earlier part of the message
                // 64 entries per ballot; didx of an earlier detection tells what became of its id
                int nfirst = 0;
#pragma unroll 1
                for (int l = 0; l < kn && !frz; ++l) {
                    const int id = (int)meas[3 * l];
                    int found = -1;
#pragma unroll 1
                    for (int j0 = 0; j0 < M && found < 0; j0 += 64) {
                        const int j = j0 + lane;
                        const unsigned long long m = __ballot(j < M && s_ids[j] == id);
                        if (m != 0ull) found = j0 + (__ffsll((long long)m) - 1);
                    }
                    int code = found;
                    if (found < 0) {
                        int first = -1;       // first earlier detection of this message with the same id
#pragma unroll 1
                        for (int q0 = 0; q0 < l && first < 0; q0 += 64) {
                            const int q = q0 + lane;
                            const unsigned long long m = __ballot(q < l && (int)meas[3 * (q < l ? q : 0)] == id);
                            if (m != 0ull) first = q0 + (__ffsll((long long)m) - 1);
                        }
                        if (first < 0) {                          // first occurrence: inserted while there is room
                            code = nfirst < room ? M + nfirst : -1;
                            capf = capf || nfirst >= room;
                            nfirst += 1;
                        } else if (didx[first] >= M) {            // its first occurrence was inserted by this message: out of range
                            frz = true;
                        } else {                                  // its first occurrence was skipped: skipped again
                            code = -1;
                            capf = true;
                        }
                    }
                    if (lane == 0) didx[l] = code;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // didx[l] is read back (uniformly) by later detections
                }
                nins = nfirst < room ? nfirst : (room > 0 ? room : 0);
            }
            if (lane == 0) {
                nx[3] = capf ? 1 : 0;       // capacity overflow (before the freeze point, if any)
                nx[1] = nins;               // insertions
                nx[2] = frz ? 1 : 0;        // freeze
            }
        }
    };
