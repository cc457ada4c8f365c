// ekf_kernel_impl.h — fused EKF-SLAM step for gfx950 (MI355X): predict + per-detection update / insertion of
// EKF::update (reference ekf_ws/src/localization_pkg/src/ekf.cpp:37-179), optionally preceded by the
// range-bearing measurement generator get_cmd (ekf_ws/src/base_pkg/src/sim_node.py:209-250).
// Included by the per-variant instantiation units ekf_inst_*.hip so the variants compile in parallel.
//
// Mapping: one workgroup of W wavefronts per filter instance, two phases per step.
//  * THIN phase.  Everything EKF::update does besides the rank-2 downdate  P -= K (H P)  touches only the rows and
//    columns T = {0,1,2} U {landmarks detected this step} of P (<= 3+2*KG of them).  They are gathered from HBM
//    into LDS (rows coalesced, columns as strided 8-byte loads), and the predict / H P / S / K / x / insertion
//    algebra runs on those LDS copies in O(k n) work; the few scalar chains (Jacobian entries, 2x2 inverse,
//    sincos/atan2) are evaluated by ONE leader lane and broadcast through LDS.  Each update leaves its
//    K (n x 2) and H P (2 x n) in LDS.
//  * BULK phase.  P is streamed exactly once, IN PLACE: 16-byte-per-lane coalesced loads, the group's rank-2
//    downdates applied in detection order from the LDS-resident K / HP, thin rows/cols patched in from their LDS
//    copies, 16-byte coalesced stores to the same addresses (every element is read and rewritten by the same lane).
//    Only a step that grows the state changes the packed leading dimension and writes into the second buffer.  A step
//    without any update or insertion writes just the three vehicle rows / columns the prediction changed.  HBM
//    traffic per step is at most the algorithmic 2*n^2*8 bytes plus the thin gather of newly visible landmarks.
//    No large register arrays: the kernel runs at high occupancy and has no upper limit on n other than LDS.
//  More than KG detections in one step are processed in groups (a second pass over P; rare: P(k>4) ~ 0.3 %).
//
// Arithmetic: plain IEEE fp64 mul/add/div (-ffp-contract=off), operation order identical to the CPU oracle's
// MODE_FAST (oracle/slam_oracle.cpp) so results are bit-identical; float truncations of the reference
// (ekf.cpp:43-44,57,75-76,115,129-131) are reproduced with real fp32 operations.
#pragma once
#include "ekf_kernel.h"

#include <stdio.h>

#include <type_traits>

#include "../../include/slam_batch.h"
#include "slam_math.h"
#include "slam_rng.h"
#include "sim_device.h"

namespace slam {

template <int NMAX, int W, int KG_, int UNR_, int KP_ = 0>
struct EkfGeom {
    static constexpr int TPB = 64 * W;
    static constexpr int LDP = (NMAX + 2) & ~1;          // LDS row length (> NMAX, even)
    static constexpr int LMAX = (NMAX - 3) / 2;
    static constexpr int KCAP = LMAX > 0 ? LMAX : 1;     // detections held per step = the landmark capacity: every message without
                                                         // repeated ids fits (one wavefront associates them, 64 at a time)
    static constexpr int KG = KG_;                       // slots of the K / H P ring = updates one pass over P can apply
    // landmark slot pairs of the thin rows / cols = DISTINCT landmarks ONE GROUP of updates can touch (a ring slot costs 3.4 KB of
    // LDS at n = 103, a pair 3.3 KB: the ring may be deeper than the pairs; a timestep with more detections than pairs runs as
    // several groups, in the decoupled loop too).  KP_ = 0: the round-3 rule (three pairs under a ring of six, else min(KG, 4));
    // variant codes >= 10000 name it (round 4: two pairs buy the fifth workgroup of a CU).
    static constexpr int KP = KP_ > 0 ? KP_ : ((KG_ > 5 && NMAX > 43 && W < 5) ? 3 : (KG_ > 4 ? 4 : KG_));
    static constexpr int KLOOP = 2 * KP;                 // detections of a timestep the decoupled loop takes (<= two groups)
    static constexpr int TS = 3 + 2 * KP;                // thin rows / cols held in LDS
    static constexpr int UNR = UNR_;                     // register pairs in flight per lane in the bulk stream
};

// (H P) of an update is kept de-interleaved by (column mod VEC): entry c lives at (c % VEC) * HS + c / VEC, so the VEC
// operands a lane of the bulk stream needs for its 16-byte vector of columns are VEC conflict-free reads of consecutive
// 16-byte entries across the lanes.  HS is the smallest stride >= ceil(LDP / VEC) that staggers the VEC sub-arrays
// over the 64 LDS banks (stride-1 accesses in c, as the thin phase makes them, then stay conflict-free too).
template <int VEC>
constexpr int hp_substride(int ldp) {
    int hs = (ldp + VEC - 1) / VEC;   // sub-array e starts e * hs entries = e * hs * 4 banks further: (64 / VEC)-bank steps
    while ((hs % 16) != (16 / VEC) && (hs % 16) != 16 - (16 / VEC)) ++hs;
    return hs;
}

// PartialPivLU inverse of a 2x2 (MatrixXd::inverse(), ekf.cpp:135); same sequence as the oracle's inv2x2_lu.
__device__ __forceinline__ bool inv2x2_lu(const double S[4], double Si[4]) {
    const bool sw = fabs(S[2]) > fabs(S[0]);
    const double a00 = sw ? S[2] : S[0], a01 = sw ? S[3] : S[1];
    const double a10 = sw ? S[0] : S[2], a11 = sw ? S[1] : S[3];
    const double l = a10 / a00;
    const double u11 = a11 - l * a01;
    const bool ok = (a00 != 0.0) && (u11 != 0.0);
    {   // column 0 of the inverse: rhs = P e_0
        const double r0 = sw ? 0.0 : 1.0, r1 = sw ? 1.0 : 0.0;
        const double y1 = r1 - l * r0;
        const double x1 = y1 / u11;
        Si[0] = (r0 - a01 * x1) / a00;
        Si[2] = x1;
    }
    {   // column 1
        const double r0 = sw ? 1.0 : 0.0, r1 = sw ? 0.0 : 1.0;
        const double y1 = r1 - l * r0;
        const double x1 = y1 / u11;
        Si[1] = (r0 - a01 * x1) / a00;
        Si[3] = x1;
    }
    return ok;
}

// Ablation switches (bit 1: no bulk stream, bit 2: no updates / insertions, bit 16: never skip the stream) produce WRONG
// filter state; they exist for timing experiments only and are compiled in by -DSLAM_ABLATE (tools/gpu_ablate.py builds
// its own library).  The release library ignores them; only the timer bits (4, 32) of SLAM_DEBUG_FLAGS stay.
#ifdef SLAM_ABLATE
#define SLAM_DBG(x) (x)
#else
#define SLAM_DBG(x) 0
#endif

#ifndef SLAM_SLEEP_RING
#define SLAM_SLEEP_RING 1     // polling intervals of the decoupled loop (units of 64 cycles): control wavefront waiting for a ring slot,
#endif
#ifndef SLAM_SLEEP_LEADER
#define SLAM_SLEEP_LEADER 2   // pass leader waiting for pending updates,
#endif
#ifndef SLAM_SLEEP_PASS
#define SLAM_SLEEP_PASS 1     // streamers waiting for the next pass
#endif
// fp32 storage: a pass moves half the bytes, so starting passes earlier (the control wavefront keeps free slots) wins: with four ring
// slots passes start at three pending updates (1.07 -> 0.99 ms/step, round 2); from five slots on two stay free (round 5, six slots: passes at
// four - 0.871 ms/step against 0.894 for four slots / three, 0.899 for six slots with four-row strips, 0.875 with passes at five).
// -DSLAM_PASS_MIN_F32=n forces a value for every fp32 variant (tuning builds).
#ifndef SLAM_PASS_MIN
#define SLAM_PASS_MIN 4   // decoupled loop: the streamers start a pass when this many updates are pending (or on request).  With the default
                          // ring of KG = 5 slots that leaves one free for the control wavefront during a pass (KG = 5 with passes at five pending:
                          // 20 % fewer passes and bytes but the stall is back, 72 vs 78 M steps/s; fp32 storage gains nothing from a fifth slot)
#endif
#ifndef SLAM_W1_WAVES
#define SLAM_W1_WAVES 3   // one-wavefront variant: wavefronts per SIMD the register allocation leaves room for
#endif
#ifndef SLAM_SD
#define SLAM_SD 3         // timesteps the measurement generator may run ahead of the filter (ring of messages in LDS)
#endif
#ifndef SLAM_CTRL_ILP
#define SLAM_CTRL_ILP 1   // control wavefront of the decoupled loop: thin downdate with batched LDS requests (thin_downdate_ctl)
#endif
#ifndef SLAM_CTRL_SG
#define SLAM_CTRL_SG 3    // ... thin slots per batch (operands of SG slots and 2 SG NU elements in flight per lane)
#endif
#ifndef SLAM_PRIO_THIN
#define SLAM_PRIO_THIN 2
#endif

// phase timers (debug only): thread 0 stores the shader-clock delta since the previous stamp to prof[block][i]
#define SLAM_STAMP(i)                                                                    \
    do {                                                                                 \
        if (prof_on && tid == 0) {                                                       \
            const unsigned long long now_ = __builtin_readcyclecounter();                \
            p.prof[(size_t)blockIdx.x * kEkfProfSlots + (i)] += now_ - tprev;   /* summed over the steps of the launch */                         \
            tprev = now_;                                                                \
        }                                                                                \
    } while (0)

// The P stream uses PLAIN loads and stores on purpose: in a multi-step launch the matrix a workgroup writes in step t is
// what it reads in step t+1, and the ~90 MB of the resident workgroups stay in the 256 MB Infinity Cache.
// Measured at L=50, batch 65536: plain 40.3 M steps/s, non-temporal loads only 39.2 M, non-temporal loads and stores
// 34.1 M (fp32 storage: 43.3 M plain vs 40.1 M non-temporal).
// Storage type ST of x and P in HBM: double (SLAM_F64) or float (SLAM_F32; arithmetic stays fp64, values are
// rounded to float when they are written back).  One 16-byte vector holds VEC = 2 doubles or 4 floats.
typedef double dbl2_t __attribute__((ext_vector_type(2)));
typedef float flt4_t __attribute__((ext_vector_type(4)));
template <class ST> struct Vec16;
template <> struct Vec16<double> {
    static constexpr int VEC = 2;
    typedef dbl2_t type;
};
template <> struct Vec16<float> {
    static constexpr int VEC = 4;
    typedef flt4_t type;
};

// identity the optimiser cannot see through: index arithmetic derived from the result is recomputed where it is
// used instead of being hoisted to the top of the kernel and kept in registers across every phase
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// value of `v` in lane `l` as a wave-uniform double (two v_readlane_b32)
__device__ __forceinline__ double rdlane(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ unsigned hi_abs(double v) {
    return (unsigned)(__double_as_longlong(v) >> 32) & 0x7fffffffu;
}

// MULTI = false: one timestep per launch (slam_step / slam_step_dev / slam_step_sim); MULTI = true: p.T timesteps per
// launch with the per-instance state resident on chip (slam_run_sim).  Same code, the loop is compiled out for T = 1.
template <int NMAX, int W, int KG_, int UNR_, class ST, int PIPE, bool MULTI, int KP_ = 0>
__global__ __launch_bounds__(64 * W, (W >= 2 ? 4 : SLAM_W1_WAVES)) void ekf_step_kernel(const EkfStepParams p) {
    using G = EkfGeom<NMAX, W, KG_, UNR_, KP_>;
    constexpr int TPB = G::TPB, LDP = G::LDP, KCAP = G::KCAP, LMAX = G::LMAX, KG = G::KG, KP = G::KP, KLOOP = G::KLOOP, TS = G::TS, UNR = G::UNR;

    __shared__ double s_xt[LDP];          // x_t  (posterior of the previous step; landmark positions for H)
    __shared__ double s_xp[LDP];          // x_pred
    __shared__ double s_R[TS * LDP];      // thin rows   R[s][c] = P[T_s][c]
    __shared__ double s_C[TS * LDP];      // thin cols   C[s][r] = P[r][T_s]
    __shared__ double2 s_K[KG * LDP];     // per update of the group: K[r][0..1]
    constexpr int VEC = Vec16<ST>::VEC;   // elements of the storage type per 16-byte vector
    constexpr int HS = hp_substride<VEC>(LDP), HPW = (VEC - 1) * HS + (LDP + VEC - 1) / VEC;   // (the last sub-array is not padded)
    __shared__ double2 s_HP[KG * HPW];    // per update of the group: (H P)[0..1][c] at hpi(c) (de-interleaved by c % VEC)
    __shared__ double s_sc[16];           // scalars computed by the leader lane (H entries, nu, S^-1, G_x ...)
    // The measurement generator does not depend on the filter, so it may run AHEAD of it: a ring of SD timesteps, slot = t % SD.
    constexpr int SD = SLAM_SD;   // (three, to stay within 40 KB of LDS = 4 workgroups per CU; fp64 had four until the ring got its fifth slot)
    __shared__ float s_meas[SD * 3 * KCAP];  // [t % SD][detection][id, range, bearing]
    __shared__ double s_tru[SD * 6];         // [t % SD] true pose before (0..2) and after (3..5) timestep t
    __shared__ int s_kraw[SD];               // [t % SD] detections in the message of timestep t (uncapped)
    __shared__ int s_sim[2];                 // timesteps generated so far (launch-relative), timestep the filter is at
    __shared__ int s_ids[LMAX > 0 ? LMAX : 1];
    __shared__ int s_didx[2 * KCAP];      // [step parity] per detection: landmark number (>= M_old: inserted this
                                          // step), -1 dropped
    __shared__ int s_next[2 * 4];         // [step parity] raw detection count, insertions, freeze, capacity overflow
    __shared__ double s_ps[2 * 10];       // [step parity] x_pred of the vehicle (3), F_x(0,2), F_x(1,2), F_v V F_v^T (4)
    __shared__ int s_chunk;               // next chunk of the bulk stream (waves take chunks as they become free)
    __shared__ int s_T[TS];               // thin slot -> state index, -1 = free.  Slots 0..2 = vehicle rows for good;
                                          // landmarks occupy the pairs (3+2j, 4+2j) and stay resident while detected
    __shared__ signed char s_slot[LDP];   // state index -> thin slot or -1
    __shared__ signed char s_need[TS];    // slot was (re)assigned: 1 = gather from HBM, 2 = new landmark (zero)
    __shared__ int s_misc[8];             // -, -, freeze, capacity (unknown ids), l1, nT, singular-S
    __shared__ double s_keep[4];          // values that live across the timesteps of one launch: true pose, error sum (kept out of
                                          // registers on purpose).  The map entries of ids 0..63 lived here too (1 KB) until round 3;
                                          // the generator reads them from global memory now (L1 / L2 hits, off the filter's critical path)
    __shared__ int s_kh[8];               // instance-steps of this launch by detection count
    __shared__ int s_ring[8];             // decoupled loop: published updates, applied updates, (unused), exit, next timestep,
                                          // flag bits raised by the control wavefront, hold (no new pass), pass in flight
    __shared__ int s_pass[4];             // decoupled loop: pass id, first update, number of updates, streamers done
    __shared__ int s_wend[KG];            // fp32 storage: a timestep ends after this update of the open group (P is rounded there)
    __shared__ unsigned s_cnt[4];         // traffic of this launch: P-stream bytes / 16 (passes read + write), other global bytes / 8 (thin
                                          // gathers, vehicle rows / columns, state vectors), passes, updates applied by passes

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    if (p.long_mode == 1) {   // (workgroup-uniform) a message this size class cannot hold: the streamed kernel's launch takes the instance (ekf_kernel.h)
        const int kk = p.meas_count_in[b];
        if ((kk < p.k_stride_in ? kk : p.k_stride_in) > p.long_cap) return;
    }

    const bool prof_on = (p.dbg & 4) && p.prof != nullptr;
    unsigned long long tprev = prof_on ? __builtin_readcyclecounter() : 0ull;
    if (prof_on && tid < 64) p.prof[(size_t)blockIdx.x * kEkfProfSlots + tid] = 0ull;
    // ---- prologue: every load that does not depend on another load is issued up front (one HBM round trip) ----
    typedef typename Vec16<ST>::type VT;
    constexpr int ESZ = (int)sizeof(ST);
    auto hpi = [](int c) -> int { return (c & (VEC - 1)) * HS + (c / VEC); };
    // T consecutive timesteps per launch: x_t / ids / truth / thin rows stay on chip, P is updated in place
    const int T = MULTI ? p.T : 1;
    ST* const PA = const_cast<ST*>(static_cast<const ST*>(p.P)) + (size_t)b * p.pstride;
    ST* const PB = static_cast<ST*>(p.P_out) + (size_t)b * p.pstride;
    ST* const Pfinal = PA;                    // where the host expects P_t after the launch
    // The stream updates P IN PLACE (every element is read and rewritten by the same lane; the thin rows / columns it
    // depends on were copied to LDS before).  Only a step that inserts landmarks changes the packed leading dimension
    // and therefore writes into the other buffer; Pcur follows the matrix.  Half the ping-pong footprint in the
    // Infinity Cache, and a step without detections touches nothing but its thin rows and columns.
    ST* Pcur = PA;
    ST* __restrict__ xb = static_cast<ST*>(p.x) + (size_t)b * p.xstride;
    constexpr bool kWide = sizeof(ST) == 8;   // fp64 storage: intermediate results can live in P_out itself
    int flags = p.flags[b];
    const int M_init = p.M[b];
    double xpre[(LDP + TPB - 1) / TPB];
#pragma unroll
    for (int u = 0; u < (LDP + TPB - 1) / TPB; ++u) {
        const int i = tid + TPB * u;
        xpre[u] = (i < p.xstride && i < LDP) ? (double)xb[i] : 0.0;   // beyond n_old the slab holds stale values: masked below
    }
    const int idpre = (tid < p.L_max) ? p.ids[(size_t)b * p.L_max + tid] : 0;
    double keep0 = 0.0;
    if (p.sim && tid < 64) {
        if (tid < 3) keep0 = p.truth[3 * (size_t)b + tid];
        if (tid == 3) keep0 = p.err_sum[b];
    }
    const int n_init = 3 + 2 * M_init;
    const int ts0 = p.timestep[b];

    if (flags & SLAM_INST_INDEX_OOR) {
        return;   // frozen instance (the reference node died here, filter.h:5): the state stays as it is
    }

#pragma unroll
    for (int u = 0; u < (LDP + TPB - 1) / TPB; ++u) {
        const int i = tid + TPB * u;
        if (i < LDP) {
            const double v = i < n_init ? xpre[u] : 0.0;
            s_xt[i] = v;
            s_xp[i] = v;
        }
    }
    if (tid < M_init) s_ids[tid] = idpre;
    if (tid < 4) s_keep[tid] = keep0;
#pragma unroll 1
    for (int i = tid; i < LDP; i += TPB) s_slot[i] = (signed char)(i < 3 ? i : -1);
    if (tid < TS) {
        s_T[tid] = tid < 3 ? tid : -1;
        s_need[tid] = (signed char)(tid < 3 ? 1 : 0);
    }
    if (tid < 8) s_kh[tid] = 0;
    if (tid < KG) s_wend[tid] = 0;
    if (tid < 2) s_sim[tid] = 0;
    if (tid < 4) s_cnt[tid] = tid == 1 ? (unsigned)((n_init + M_init / 2 + 8) * ESZ / 8) : 0u;   // x_t, ids, scalars read by the prologue
    // one lane accounts for what a phase moves (wave-uniform arguments; LDS atomics, a handful per timestep)
    // (32-bit arithmetic and no captured state on purpose: the kernel sits at its register limit)
    auto count_pass = [](unsigned* cnt, int vec16, int nupd) {   // vec16: 16-byte vectors read + written
        atomicAdd(&cnt[0], (unsigned)vec16);
        atomicAdd(&cnt[2], 1u);
        atomicAdd(&cnt[3], (unsigned)nupd);
    };
    auto count_other = [](unsigned* cnt, int elems) { atomicAdd(&cnt[1], (unsigned)(elems * ESZ) / 8u); };

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
    if (W == 1) pregather(tid, TPB);
    else if (tid >= 64) pregather(tid - 64, TPB - 64);
    if (tid == TPB - 1) count_other(s_cnt, 6 * n_init);
    if (tid < 64) prestep(0);
    if (tid >= TPB - 3) s_need[tid - (TPB - 3)] = 0;   // slots 0..2 are resident (visible after the barrier at the top of the step)
    SLAM_STAMP(1);   // measurements, association, motion scalars of the first step

    // a freezing instance leaves both loops and writes its PRE-step state below (the reference node died at that step)
    bool wd_fired = false;
    int frz_at = -1, frz_M = 0, frz_n = 0;
    const ST* frz_P = nullptr;
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
    const ST* const Pin = Pcur;
    const int pb = t & 1;
    const float* const meas_t = s_meas + (t % SD) * 3 * KCAP;
    int* const didx_t = s_didx + pb * KCAP;
    const int M_old = M;
    const int n_old = na;
    if (tid < 8) s_misc[tid] = 0;
    __builtin_amdgcn_s_setprio(SLAM_PRIO_THIN);   // the thin phases are dependent chains: let them issue ahead of other workgroups' streams
    __syncthreads();   // the pre-step results of this timestep are visible
    const int kraw = s_next[4 * pb];
    if (kraw > KCAP) flags |= SLAM_INST_CAPACITY;   // more detections in one message than the landmark capacity (only possible with repeated ids)
    const int k = kraw < KCAP ? kraw : KCAP;
    if (tid == 0) s_kh[k < 7 ? k : 7] += 1;
    if (p.sim && p.meas_out != nullptr && t == T - 1) {
        for (int i = tid; i < 3 * k && i < 3 * p.k_stride_out; i += TPB)
            p.meas_out[(size_t)b * p.k_stride_out * 3 + i] = meas_t[i];
        if (tid == 0) p.meas_count_out[b] = k;
    }
    // insertions this step; unknown ids: an upper bound - every detection could be a new landmark - but never more than the
    // capacity has room for (the matrix of this step is laid out for n_old + 2 n_ins: without the clamp a wide message at a full
    // map provisioned rows past the instance's slab - found by tools/gpu_soak_ekf.py, a memory fault with fp32 storage)
    const int room_ins = (p.L_max < LMAX ? p.L_max : LMAX) - M_old;
    const int n_ins = p.id_known ? s_next[4 * pb + 1] : (k < room_ins ? k : (room_ins > 0 ? room_ins : 0));
    const bool frz_top = p.id_known && s_next[4 * pb + 2];  // freeze in the pre-step state (after the pending group is flushed)
    if (p.id_known && s_next[4 * pb + 3]) flags |= SLAM_INST_CAPACITY;
    SLAM_STAMP(2);   // association

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

    // ------------------------------------------------------------------------------------------------------
    // x_t = x_pred (ekf.cpp:176) and bookkeeping.  P_t = P_pred was written by the bulk stream.
    // ------------------------------------------------------------------------------------------------------
    const int te = opaque(tid);
    for (int i = te; i < na; i += TPB) {
        const ST sv = (ST)s_xp[i];   // storage rounding of x_t (identity for fp64)
        s_xt[i] = (double)sv;
        s_xp[i] = (double)sv;
        const unsigned h0 = hi_abs((double)sv);
        hiacc = hiacc > h0 ? hiacc : h0;
    }
    if constexpr (!kWide) {
        // resident thin rows/cols must equal what HBM holds: apply the storage rounding to them as well
        if (t + 1 < T) {
            const int nTl = s_misc[5];
#pragma unroll 1
            for (int i = te; i < nTl * LDP; i += TPB) {
                s_R[i] = (double)(ST)s_R[i];
                s_C[i] = (double)(ST)s_C[i];
            }
        }
    }
    if (s_misc[6]) flags |= SLAM_INST_S_SINGULAR;
    const bool nonfinite = __syncthreads_or(hiacc >= 0x7ff00000u);
    if (nonfinite) flags |= SLAM_INST_NONFINITE;

    if (na != nf) {
        // fewer insertions than provisioned (unknown-id mode over-estimates): re-pack from the leading dimension of nf
        // to the one of na in place.  Rows move towards lower addresses, so go row by row with a barrier in between.
        const int ldf = ekf_ld(nf, ESZ), lda = ekf_ld(na, ESZ);
        if (tid == 0) count_other(s_cnt, 2 * na * na);
#pragma unroll 1
        for (int r = 0; r < na; ++r) {
            ST tmp[(LDP + TPB - 1) / TPB];
#pragma unroll
            for (int u = 0; u < (LDP + TPB - 1) / TPB; ++u) {
                const int c = te + TPB * u;
                tmp[u] = c < na ? Pout[(size_t)r * ldf + c] : (ST)0;   // pad columns: zero
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < (LDP + TPB - 1) / TPB; ++u) {
                const int c = te + TPB * u;
                if (c < lda) Pout[(size_t)r * lda + c] = tmp[u];
            }
            __syncthreads();
        }
    }
    // per-timestep stamp of a multi-step launch: 100 MHz wall clock << 4 | detections of this instance-step
    if ((p.dbg & 32) && p.prof != nullptr && tid == 0 && t < kEkfProfSlots)
        p.prof[(size_t)blockIdx.x * kEkfProfSlots + t] = (wall_clock64() << 4) | (unsigned long long)(k < 15 ? k : 15);
    Pcur = Pout;
    SLAM_STAMP(10);   // end of step: x_t = x_pred, flags, stamps
    }   // timestep loop

    if (wd_fired) {
        finish(0, M, flags | SLAM_INST_WATCHDOG | SLAM_INST_INDEX_OOR, true);
        return;
    }
    if (frz_at >= 0) {   // pre-step state of the frozen instance into the buffer the host reads next
        __syncthreads();
        if (Pfinal != frz_P) {
            const int nn = frz_n * ekf_ld(frz_n, ESZ);
            if (tid == 0) count_other(s_cnt, 2 * nn);
            for (int i = tid; i < nn; i += TPB) Pfinal[i] = frz_P[i];
            __syncthreads();
        }
        finish(frz_at, frz_M, flags | SLAM_INST_INDEX_OOR, true);
        return;
    }
    if (Pcur != Pfinal) {   // an odd number of layout changes in this launch: bring P_t back to the host's buffer
        __syncthreads();
        const int nn = na * ekf_ld(na, ESZ);
        if (tid == 0) count_other(s_cnt, 2 * nn);
        for (int i = tid; i < nn; i += TPB) Pfinal[i] = Pcur[i];
        __syncthreads();
    }

    finish(T, M, flags, false);
    SLAM_STAMP(8);   // epilogue
}

template <int NMAX, int W, int KG_, int UNR_, class ST, int PIPE, int KP_>
hipError_t launch_variant(const EkfStepParams& p, hipStream_t stream) {
    if (p.cmds != nullptr && p.T > 1)
        hipLaunchKernelGGL((ekf_step_kernel<NMAX, W, KG_, UNR_, ST, PIPE, true, KP_>), dim3(p.B), dim3(64 * W), 0, stream, p);
    else   // a single step takes (fwd, ang); the host sets them to the first command of the chunk
        hipLaunchKernelGGL((ekf_step_kernel<NMAX, W, KG_, UNR_, ST, PIPE, false, KP_>), dim3(p.B), dim3(64 * W), 0, stream, p);
    return hipGetLastError();
}

template <int NMAX, int W, int KG_, int UNR_, class ST, int PIPE, int KP_>
hipError_t variant_info(int multi, EkfKernelInfo* out) {
    const void* fn = multi ? reinterpret_cast<const void*>(&ekf_step_kernel<NMAX, W, KG_, UNR_, ST, PIPE, true, KP_>)
                           : reinterpret_cast<const void*>(&ekf_step_kernel<NMAX, W, KG_, UNR_, ST, PIPE, false, KP_>);
    hipFuncAttributes a;
    hipError_t e = hipFuncGetAttributes(&a, fn);
    if (e != hipSuccess) return e;
    int nb = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * W, 0);
    if (e != hipSuccess) return e;
    snprintf(out->name, sizeof(out->name), "ekf_step_kernel<%d,%d,%d,%d,%s,%d,%s,%d>", NMAX, W, KG_, UNR_,
             sizeof(ST) == 8 ? "double" : "float", PIPE, multi ? "true" : "false", KP_);
    out->lds_bytes = (int)a.sharedSizeBytes;
    out->vgprs = a.numRegs;
    out->sgprs = 0;
    out->threads = 64 * W;
    out->wg_per_cu = nb;
    return hipSuccess;
}

}  // namespace slam
