// ekf_step_control.h — part of the BODY of ekf_step_kernel (ekf_kernel_impl.h includes it inside the kernel function; round 6: the 1 840-line kernel split into its
// parts, pure moves - every object file byte-identical).  The dependent scalar chain of one landmark update on one wavefront (`leader_chain`), group formation for known ids (`form_known`), the downdates of the thin rows / columns in LDS (`thin_downdate`, `thin_downdate_ctl`).
// Lambdas and statements here capture the kernel's locals (p, tid, lane, the LDS arrays ...): not a stand-alone header.  DESIGN.md 4.1.

    // ---- the scalar chain of one landmark update (ekf.cpp:110-135), evaluated by ONE wavefront without a barrier: the eight
    //      quotients of H on eight lanes at once, atan2 beside them (independent chains in one instruction stream), the five
    //      columns of H P that S needs on five lanes, results passed between lanes as wave-uniform values (v_readlane).
    //      H = {H00, H01, H0i, H0i+1, H10, H11, H1i, H1i+1} (H12 = -1), innovation (nu0, nu1), Si = S^-1.  Every lane of the
    //      wavefront returns the same values.  false: zero pivot in the PartialPivLU of S. ----
    auto leader_chain = [&](int ii, int si, float r_m, float b_m, double (&H)[8], double& nu0, double& nu1, double (&Si)[4]) -> bool {
        const double* const xl = p.lm_from_pred ? s_xp : s_xt;   // quirk D-2 (ekf.cpp:115-116): the landmark is read from x_t
        const double dx = xl[ii] - s_xp[0], dy = xl[ii + 1] - s_xp[1];
        const float dist = (float)sqrt(dx * dx + dy * dy);
        const double dd = (double)dist, d2 = (double)(dist * dist);
        // lane j < 8: H entry j = num_j / den_j
        const int hl = lane & 7;
        const bool usey = (hl == 1) || (hl == 3) || (hl == 4) || (hl == 6);
        const bool neg = (hl == 0) || (hl == 1) || (hl == 5) || (hl == 6);
        double num = usey ? dy : dx;
        num = neg ? -num : num;
        const double q = num / (hl < 4 ? dd : d2);
        const float angf = (float)rem2pi(det_atan2(dy, dx) - s_xp[2]);
        const float nu0f = r_m - dist - p.w_r;     // float arithmetic (ekf.cpp:130-131)
        const float nu1f = b_m - angf - p.w_b;
        nu0 = (double)nu0f; nu1 = (double)nu1f;
#pragma unroll
        for (int j = 0; j < 8; ++j) H[j] = rdlane(q, j);
        const double h00 = H[0], h01 = H[1], h03 = H[2], h04 = H[3], h10 = H[4], h11 = H[5], h12 = -1.0, h13 = H[6], h14 = H[7];
        // the columns 0, 1, 2, i, i+1 of H P (lanes 0..4), same expression as the full pass over all columns
        const int cs = lane < 3 ? lane : (lane == 3 ? ii : ii + 1);
        const double p0 = s_R[cs], p1 = s_R[LDP + cs], p2 = s_R[2 * LDP + cs], pi = s_R[si * LDP + cs], pj = s_R[(si + 1) * LDP + cs];
        const double gx = ((h00 * p0 + h01 * p1) + h03 * pi) + h04 * pj;
        const double gy = (((h10 * p0 + h11 * p1) + h12 * p2) + h13 * pi) + h14 * pj;
        const double g0x = rdlane(gx, 0), g1x = rdlane(gx, 1), g2x = rdlane(gx, 2), gix = rdlane(gx, 3), gjx = rdlane(gx, 4);
        const double g0y = rdlane(gy, 0), g1y = rdlane(gy, 1), g2y = rdlane(gy, 2), giy = rdlane(gy, 3), gjy = rdlane(gy, 4);
        double S[4];   // S = (H P) H^T + W (ekf.cpp:133)
        S[0] = ((g0x * h00 + g1x * h01) + gix * h03) + gjx * h04;
        S[1] = (((g0x * h10 + g1x * h11) + g2x * h12) + gix * h13) + gjx * h14;
        S[2] = ((g0y * h00 + g1y * h01) + giy * h03) + gjy * h04;
        S[3] = (((g0y * h10 + g1y * h11) + g2y * h12) + giy * h13) + gjy * h14;
        S[0] = S[0] + p.W00;
        S[3] = S[3] + p.W11;
        return inv2x2_lu(S, Si);
    };

    // ---- group formation for KNOWN ids, lane-parallel in ONE wavefront: lane l <-> detection l0 + l of the group AND thin
    //      slot pair l.  Landmarks that are detected again keep their slot (their LDS copy IS the current P row / column),
    //      the others give theirs up, newly wanted ones take the lowest free pairs in detection order (s_need: 1 = gather
    //      from HBM, 2 = new landmark, starts from zeros).  Returns whether a gather is needed; l1 = end of the group, nT =
    //      high-water mark of the slots in use.  Same assignment as the serial path for unknown ids below. ----
    auto form_known = [&](const int* didx_g, int k, int l0, int lim, int nsrc, int& l1_out, int& nT_out) -> int {
        const int l1 = (k - l0 < lim) ? k : l0 + lim;
        const int ng = l1 - l0;                                   // detections of this group (<= KP)
        const int idx = (lane < ng) ? didx_g[l0 + lane] : -1;
        const int myii = idx >= 0 ? 3 + 2 * idx : -1;             // wanted state index of detection lane
        const int cur = (lane < KP) ? s_T[3 + 2 * lane] : -1;     // landmark in slot pair lane
        bool dupl = false, has = false, keep = false;
#pragma unroll
        for (int w = 0; w < KP; ++w) {
            const int ii_w = __builtin_amdgcn_readlane(myii, w), cur_w = __builtin_amdgcn_readlane(cur, w);   // (v_readlane: no LDS crossbar trip)
            dupl = dupl || (w < lane && ii_w == myii);            // an earlier detection wants the same landmark
            has = has || (cur_w >= 0 && cur_w == myii);           // my landmark already has a slot
            keep = keep || (ii_w >= 0 && ii_w == cur);            // somebody wants the landmark in my slot
        }
        const bool wantv = myii >= 0 && !dupl;
        const bool release = lane < KP && cur >= 0 && !keep;
        if (release) {   // the pending updates (or the last pass) produce this row / column in HBM bit for bit
            s_slot[cur] = (signed char)-1; s_slot[cur + 1] = (signed char)-1;
            s_T[3 + 2 * lane] = -1; s_T[4 + 2 * lane] = -1;
        }
        const bool freeslot = lane < KP && (cur < 0 || !keep);
        const unsigned long long fmask = __ballot(freeslot);
        const bool needs = wantv && !has;
        const unsigned long long nmask = __ballot(needs);
        const int rank = __popcll(nmask & ((1ull << lane) - 1ull));
        int j = 0;                                                 // the rank-th free pair
        {
            unsigned long long fm = fmask;
#pragma unroll
            for (int w = 0; w < KP; ++w) {
                const int lowest = __ffsll((long long)fm) - 1;
                if (w == rank) j = lowest;
                fm &= fm - 1ull;
            }
        }
        bool gath = false;
        if (needs) {
            s_T[3 + 2 * j] = myii; s_T[4 + 2 * j] = myii + 1;
            s_slot[myii] = (signed char)(3 + 2 * j); s_slot[myii + 1] = (signed char)(4 + 2 * j);
            const signed char nd = (signed char)(myii < nsrc ? 1 : 2);   // known landmark: gather, new one: zeros
            s_need[3 + 2 * j] = nd; s_need[4 + 2 * j] = nd;
            gath = nd == 1;
        }
        // occupied pairs after release + assignment: the kept ones and the lowest free ones the needing lanes took
        unsigned long long occ = __ballot(lane < KP && cur >= 0 && keep);
        {
            unsigned long long fm = fmask;
            const int ntake = __popcll(nmask);
#pragma unroll
            for (int w = 0; w < KP; ++w) {
                const int lowest = __ffsll((long long)fm) - 1;
                if (w < ntake && lowest >= 0) occ |= 1ull << lowest;
                fm &= fm - 1ull;
            }
        }
        l1_out = l1;
        nT_out = occ ? 5 + 2 * (63 - __clzll((long long)occ)) : 3;   // slots [3, nT) may contain free pairs (s_T < 0)
        return __ballot(gath) != 0ull ? 1 : 0;
    };

    // ---- the thin copies follow a downdate  P -= K (H P):  R[s][j] -= K[T_s] . (H P)[j],  C[s][j] -= K[j] . (H P)[T_s].
    //      A thread owns state index j (its K[j], (H P)[j] are read once) and walks the slots s0, s0 + sstride, ...; the
    //      slot operands K[T_s], (H P)[T_s] are the same address for all lanes (LDS broadcast).  One downdate per element,
    //      same expression as the bulk stream. ----
    auto thin_downdate = [&](int j0, int jstride, int s0, int sstride, int nTd, int nd, const double2* __restrict__ Ku,
                             const double2* __restrict__ HPu) {
        const int* const Ttab = s_T;
#pragma unroll 1
        for (int j = j0; j < nd; j += jstride) {
            const double2 kj = Ku[j], hj = HPu[hpi(j)];
#pragma unroll 2
            for (int sl = s0; sl < nTd; sl += sstride) {
                const int t_s = Ttab[sl];
                if ((unsigned)t_s < (unsigned)nd) {   // wave-uniform
                    const double2 kt = Ku[t_s], ht = HPu[hpi(t_s)];
                    const int i = sl * LDP + j;
                    s_R[i] = s_R[i] - (kt.x * hj.x + kt.y * hj.y);   // P[T_s][j]
                    s_C[i] = s_C[i] - (kj.x * ht.x + kj.y * ht.y);   // P[j][T_s]
                }
            }
        }
    };

    // ---- the same downdate for the CONTROL wavefront of the decoupled loop (round 4), written for memory-level parallelism: the loop
    //      above makes two DEPENDENT LDS round trips per slot and state index (slot table -> operands -> element), 18 of them per
    //      update at n = 103, and that latency was 15 % of the control wavefront's timeline.  Here the slot table is read once (one
    //      batch), a lane's own K[j] / (H P)[j] arrive in registers from the phase that computed them, and the slots are walked in
    //      groups of SG: the group's operands K[T_s], (H P)[T_s] and its elements of R and C are requested together, then updated and
    //      stored.  Same expression per element, so not a bit changes; it wants registers (W = 3 variants: 168 VGPRs). ----
    constexpr int NU = (LDP + 63) / 64;   // state indices per lane of ONE wavefront
    auto thin_downdate_ctl = [&](int nTd, int nd, const double2* __restrict__ Ku, const double2* __restrict__ HPu, const double2 (&kj)[NU],
                                 const double2 (&hj)[NU]) {
        constexpr int SG = SLAM_CTRL_SG;
        int tsv[TS];
#pragma unroll
        for (int sl = 0; sl < TS; ++sl) tsv[sl] = s_T[sl];
#pragma unroll
        for (int s0 = 0; s0 < TS; s0 += SG) {
            if (s0 >= nTd) break;   // wave-uniform
            double2 kt[SG], ht[SG];
            double rv[SG][NU], cv[SG][NU];
            bool ok[SG];
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                const int sl = s0 + g < TS ? s0 + g : TS - 1;
                const int t_s = tsv[sl];
                ok[g] = s0 + g < nTd && (unsigned)t_s < (unsigned)nd;   // wave-uniform
                const int tc = ok[g] ? t_s : 0;
                kt[g] = Ku[tc]; ht[g] = HPu[hpi(tc)];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int j = lane + 64 * u;
                    const int i = sl * LDP + (j < LDP ? j : 0);
                    rv[g][u] = s_R[i]; cv[g][u] = s_C[i];
                }
            }
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                if (!ok[g]) continue;
                const int sl = s0 + g;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int j = lane + 64 * u;
                    if (j < nd) {
                        const int i = sl * LDP + j;
                        s_R[i] = rv[g][u] - (kt[g].x * hj[u].x + kt[g].y * hj[u].y);   // P[T_s][j]
                        s_C[i] = cv[g][u] - (kj[u].x * ht[g].x + kj[u].y * ht[g].y);   // P[j][T_s]
                    }
                }
            }
        }
    };

    // (Round 3, measured and dropped: a one-wavefront variant that keeps a lane's K / (H P) entries in registers and fetches the slot
    // operands once per slot - thin downdates 5.3 k -> 4.8 k cycles per step, but five more spilled registers moved the same cycles
    // into the prediction and the end of the step: 77.4 vs 77.1 M steps/s.  At 128 VGPRs every added live range is paid elsewhere.)
