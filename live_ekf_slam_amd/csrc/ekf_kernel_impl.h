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

#include "ekf_step_prestep.h"   // finish (write-back of an instance), simgen (measurement generator), prestep (motion scalars, association, next step's bookkeeping)
#include "ekf_step_control.h"   // leader_chain (scalar chain of one update on one wavefront), form_known, thin_downdate / thin_downdate_ctl (thin rows / columns in LDS)
#include "ekf_step_stream.h"   // PassArgs, stream_pass (in-place bulk stream of P with the deferred rank-2 updates), mid_pass, write_vehicle, pregather
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

#include "ekf_step_decoupled.h"   // the DECOUPLED steady-state loop: control wavefront + ring of updates + streamers (MULTI && W >= 2)
#include "ekf_step_lockstep.h"   // the barrier-synchronised timestep: insertions, unknown ids, freezes, > KG detections, single-step launches
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
