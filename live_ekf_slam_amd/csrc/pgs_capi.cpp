// pgs_capi.cpp — C ABI (include/slam_pgs.h) over the pose-graph kernels.  Host side only: owns the device memory,
// the stream and the lockstep timestep; every numeric operation happens in pgs_kernel.hip.  No CPU fallback.
#include "../../include/slam_pgs.h"

#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <thread>
#include <vector>

#include "capi_internal.h"
#include "pgs_kernel.h"

#define fail slam_internal_fail
// (The runtime's "last error" is sticky and per thread: a launcher that ends in hipGetLastError() would report an error some OTHER library
// of the process left behind - PyTorch creating a stream right before slam_init did exactly that in a test.  It is cleared before every
// call; our own calls are all checked through their return values.)
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        (void)hipGetLastError();                                                                   \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(SLAM_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

struct pgs_handle {
    slam_config cfg;
    int B, N_max, L_max, KP, LD, device;
    int timestep = 0;
    bool inited = false;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint64_t seed = 2025;
    int64_t inst0 = 0;
    slam::PgsParams p;
    std::vector<void*> allocs;
    float* dcmds = nullptr;
    float* dmeas = nullptr; int32_t* dcount = nullptr; double* dsec = nullptr; int k_stride = 0;
    double* dout = nullptr;
    int max_trials = 400;
    int lanes = 4;                             // slots per instance for speculative lambda lanes (SLAM_PGS_LANES, 1 = off)
    int lanes_switch_all = 16;                 // ... from which down ALL lanes are used (SLAM_PGS_LANES_SWITCH_ALL)
    int lanes_switch = 64;                     // active instances (of the whole batch) from which down the lanes are used (SLAM_PGS_LANES_SWITCH)
    struct Slab { void* ptr; size_t bytes; };  // per-slot arrays the clones need a copy of when a solve begins (bytes per slot)
    std::vector<Slab> clone_slabs;
    // solve groups: the batch is split into `groups` contiguous ranges whose LM loops run on their own streams, so
    // the latency-bound phases of one group overlap the bandwidth-bound phases of another (0 = choose from the batch)
    int groups = 0;
    std::vector<hipStream_t> gstreams;
    std::vector<hipEvent_t> gevents;
    int32_t* h_active = nullptr;               // pinned host: per-group active counts
    // streaming (round 6, pgs_kernel.h "streaming"): at most `slots` graphs of the batch are in flight (0 = lockstep: all of them), split over
    // the solve groups; trials are enqueued `stream_depth` ahead of the host's reading of their counters
    int slots = 0, stream_depth = 3;
    static constexpr int kRing = 8, kMaxGroups = 16;
    int32_t* d_cnt = nullptr;                  // device [kMaxGroups][2][8]: the counter blocks, ping-pong by trial parity
    int32_t* d_wait = nullptr;                 // device [kMaxGroups]: the groups' wait cursors
    int32_t* h_ring = nullptr;                 // pinned host [kMaxGroups][kRing][8]
    std::vector<hipEvent_t> ring_events;       // [kMaxGroups][kRing]
    std::vector<std::vector<int32_t>> timeline;   // per group: slots that ran in every trial of the last solve
    // asynchronous ticks (pgs_run_sim_every_iteration; pgs_kernel.h "asynchronous ticks")
    // SLAM_PGS_ITER_ASYNC=1: every graph walks through its ticks at its own pace.  Off by default: measured slower on BASELINE configs[4] (9.0 k
    // against 15.0 k graph-ticks/s) - a run lasts as long as its HARDEST graph's dependent chain of lambda trials (42 per tick for the slowest of
    // 256 graphs, median 3: profiles/r06_pgs/iterative_mode.txt), which the lockstep loop shortens with its speculative lambda lanes
    int iter_async = 0;
    int32_t* d_Nv = nullptr;                   // [slots + 1] poses per graph
    int32_t* d_mono = nullptr;                 // [2]
    hipStream_t tick_stream = nullptr;
    std::vector<hipEvent_t> async_events;      // [2 * kRing + 1]: decide done, tick step done (rings), start
    long long async_trials = 0;
    double* d_tick_flop = nullptr;             // [B][2] algorithmic FLOP (SYRK | Cholesky) of the same
    int32_t* d_tick = nullptr;                 // [B][2] LM iterations / trials summed over the ticks of pgs_run_sim_every_iteration
    double iter_ms[4] = {0, 0, 0, 0};
    bool host_prof = false;                    // SLAM_PGS_HOST_PROF: host clock spent enqueuing trials / waiting for their counters (stderr, every-iteration runs)
    double host_launch_ms = 0.0, host_wait_ms = 0.0;
    long long iter_trials = 0;
    int p_notrim = 0;
    int chol_ll = 2;                           // SLAM_PGS_CHOL_LL=0: the right-looking Cholesky of rounds 1-3; 1: the left-looking kernel on 1024 threads; 2 (default): on 768
    // SLAM_PGS_CHOL_THREADS = 256 | 1024 forces a Cholesky build; else the 256-thread right-looking one while > chol_switch slots run.  Round 6:
    // never by default - since round 5's work on the left-looking kernel it wins at every count (batch 1024: 7.99 k -> 9.45 k solves/s,
    // profiles/r06_pgs/stream_table.txt), and with ONE Cholesky build a graph's result no longer depends on how many others run beside it
    int chol_threads = 0, chol_switch = 1 << 30;
    bool trace = false;                       // SLAM_PGS_TRACE: print the active-instance count after every trial
    double path_ms[3] = {0.0, 0.0, 0.0};      // profiled solve: ms in the separate SYRK launches / in the fused chain + SYRK launches / in the segmented path's SYRK launches
    int seg_len = 32;                         // SLAM_PGS_SEG: poses per segment of the segmented elimination (pgs_seg_impl.h), 0 = the sequential chain of rounds 1-4
    bool seg_ok = false;                      // this solve runs the segmented elimination (decided in pgs_solve from the plan)
    // Round 6: the segment length is chosen PER SOLVE from {seg_len, seg_len / 2, ... >= 8}: the longest whose every segment sees at most
    // kPgsSegMaxLm landmarks (a wide sensor on a dense map overflowed 32-pose segments and the whole solve fell back to the sequential chain,
    // 2.3 x slower).  seg_cur: where the next solve's search starts (only goes down while the graph grows; pgs_init resets it); seg_alloc: the
    // length the segment-count-dependent arrays are sized for (re-made on demand, resize_segments); seg_used: the last solve's.
    int seg_cur = 32, seg_alloc = 32, seg_used = 0;
    bool fused_ok = false;                    // this solve's graphs fit the fused kernel (decided in pgs_solve from max M)
    int cus = 256;                            // compute units of the device
    bool use_list = true;                     // SLAM_PGS_LIST=0: full-size grids, inactive workgroups return (the round-2 launch shape)
    int fused_mode = -1;                      // SLAM_PGS_FUSED: 0 = chain and SYRK as two launches, -1 (default) = fused when the tiles fit
    int syrk_inst_switch = 100;                  // SLAM_PGS_SYRK_INST_SWITCH: active count from which the instance-resident SYRK runs (SLAM_PGS_SYRK_TILE=1 forces it)
    int syrk_tile = 0, syrk_switch = 1 << 30;    // SLAM_PGS_SYRK_TILE = 32 | 64 forces a variant; SLAM_PGS_SYRK_SWITCH = active count from which
                                             // the 64x64-per-wavefront variant is used (default: never — measured slower at every batch size)
    int last_trials = 0;
    bool profiling = false;                  // per-kernel hipEvent timing of pgs_solve (pgs_set_profiling)
    std::vector<hipEvent_t> events;
    std::vector<int> trial_fused;             // profiled solve: the fused choice (0 | 2 | 3 | 4) of every trial
    double kernel_ms[slam::kPgsTrialKernels] = {0, 0, 0, 0, 0, 0};
};

namespace {

template <class T>
int dalloc(pgs_handle* h, T** out, size_t count) {
    void* ptr = nullptr;
    HIP_TRY(hipMalloc(&ptr, sizeof(T) * (count ? count : 1)));
    h->allocs.push_back(ptr);
    *out = (T*)ptr;
    return SLAM_OK;
}

#define TRY(expr)                   \
    do {                            \
        const int rc_ = (expr);     \
        if (rc_ != SLAM_OK) return rc_; \
    } while (0)

int round_up(int v, int m) { return (v + m - 1) / m * m; }

int check(pgs_handle* h) {
    if (!h) return fail(SLAM_ERR_ARG, "NULL handle");
    HIP_TRY(hipSetDevice(h->device));
    return SLAM_OK;
}

// the segments' Gram matrices: [slots][nseg_max][seg_tld^2] doubles, only when a solve runs the segmented order (128 x 128 per
// segment whatever L_max, allocated also for handles whose every solve takes the sequential chain)
int ensure_segT(pgs_handle* h) {
    if (h->p.segT || h->seg_len <= 0) return SLAM_OK;
    const size_t S = (size_t)h->B * h->lanes;
    return dalloc(h, &h->p.segT, S * (size_t)h->p.nseg_max * (size_t)h->p.seg_tld * h->p.seg_tld);
}

template <class T>
void dfree(pgs_handle* h, T*& ptr) {
    if (!ptr) return;
    for (size_t i = 0; i < h->allocs.size(); ++i)
        if (h->allocs[i] == (void*)ptr) { h->allocs.erase(h->allocs.begin() + (long)i); break; }
    hipFree((void*)ptr);
    ptr = nullptr;
}

// The arrays whose size follows the number of segments (the plan, the segments' outputs, their Gram matrices, and Y - whose separator rows
// live behind the pose rows) re-made for segments of SL poses, if they were sized for longer ones.  Nothing in them outlives a solve.
int resize_segments(pgs_handle* h, int SL) {
    if (SL >= h->seg_alloc) return SLAM_OK;
    HIP_TRY(hipDeviceSynchronize());
    slam::PgsParams& p = h->p;
    const size_t S = (size_t)h->B * h->lanes, L = (size_t)h->L_max;
    void* old[6] = {p.seg_ncol, p.seg_lm, p.seg_inv, p.seg_evt, p.seg_blk, p.sep_evt};
    dfree(h, p.seg_ncol); dfree(h, p.seg_lm); dfree(h, p.seg_inv); dfree(h, p.seg_evt); dfree(h, p.seg_blk); dfree(h, p.sep_evt);
    dfree(h, p.segout); dfree(h, p.sepfac); dfree(h, p.segT); dfree(h, p.Y);
    p.nseg_max = (h->N_max - 2) / SL + 1;
    p.yr_sep = p.yr_rc + 6 * (int64_t)p.nseg_max;
    p.y_stride = (int64_t)(p.yr_sep + round_up(3 * p.nseg_max, 4)) * h->LD;
    const size_t G = (size_t)p.nseg_max;
    const size_t per[6] = {G, G * L, G * L, G * L, G * (size_t)slam::seg_nb1(h->L_max), G * L};
    TRY(dalloc(h, &p.seg_ncol, S * per[0])); TRY(dalloc(h, &p.seg_lm, S * per[1])); TRY(dalloc(h, &p.seg_inv, S * per[2]));
    TRY(dalloc(h, &p.seg_evt, S * per[3])); TRY(dalloc(h, &p.seg_blk, S * per[4])); TRY(dalloc(h, &p.sep_evt, S * per[5]));
    void* now[6] = {p.seg_ncol, p.seg_lm, p.seg_inv, p.seg_evt, p.seg_blk, p.sep_evt};
    for (pgs_handle::Slab& sl : h->clone_slabs)     // the lambda lanes get copies of the plan (clone_instances)
        for (int k = 0; k < 6; ++k)
            if (sl.ptr == old[k]) {   // (once per slab: the allocator may hand a new array the address another old one had)
                sl.ptr = now[k]; sl.bytes = per[k] * sizeof(int32_t);
                break;
            }
    TRY(dalloc(h, &p.segout, S * G * 32)); TRY(dalloc(h, &p.sepfac, S * G * 16));
    TRY(dalloc(h, &p.Y, S * (size_t)p.y_stride));
    h->seg_alloc = SL;
    return SLAM_OK;   // (segT: ensure_segT, by the solve that runs the order)
}

int ensure_staging(pgs_handle* h, int k_stride) {
    if (h->dmeas && h->k_stride >= k_stride) return SLAM_OK;
    if (h->dmeas) { hipFree(h->dmeas); h->dmeas = nullptr; }
    HIP_TRY(hipMalloc((void**)&h->dmeas, sizeof(float) * 3 * (size_t)k_stride * h->B));
    h->k_stride = k_stride;
    return SLAM_OK;
}

}  // namespace

extern "C" {

int pgs_create(const slam_config* cfg, int batch, int N_max, int L_max, int k_per_pose, int device, pgs_handle** out) {
    if (!cfg || !out) return fail(SLAM_ERR_ARG, "NULL argument");
    if (batch <= 0 || N_max < 2 || L_max <= 0 || k_per_pose <= 0) return fail(SLAM_ERR_ARG, "batch, N_max (>= 2), L_max and k_per_pose must be positive");
    if (L_max > 255) return fail(SLAM_ERR_UNSUPPORTED, "L_max %d exceeds the pose-graph kernel limit 255", L_max);
    if (!cfg->landmark_id_is_known) return fail(SLAM_ERR_UNSUPPORTED, "PGS with unknown landmark ID is not supported (the reference throws: pose_graph.cpp:137)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SLAM_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SLAM_ERR_ARG, "device %d out of range (%d devices)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    pgs_handle* h = new pgs_handle();
    h->cfg = *cfg; h->B = batch; h->N_max = N_max; h->L_max = L_max; h->KP = k_per_pose; h->device = device;
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0) h->cus = cu; }
    h->LD = round_up(2 * L_max + 1, 64);
    if (const char* e = getenv("SLAM_PGS_MAX_TRIALS")) h->max_trials = atoi(e) > 0 ? atoi(e) : h->max_trials;
    if (const char* e = getenv("SLAM_PGS_LANES")) h->lanes = atoi(e) >= 1 && atoi(e) <= 8 ? atoi(e) : h->lanes;
    if (const char* e = getenv("SLAM_PGS_LANES_SWITCH")) h->lanes_switch = atoi(e);
    if (const char* e = getenv("SLAM_PGS_LANES_SWITCH_ALL")) h->lanes_switch_all = atoi(e);
    if (const char* e = getenv("SLAM_PGS_SYRK_TILE")) h->syrk_tile = atoi(e) == 64 ? 64 : (atoi(e) == 32 ? 32 : (atoi(e) == 1 ? 1 : 0));
    if (const char* e = getenv("SLAM_PGS_SYRK_INST_SWITCH")) h->syrk_inst_switch = atoi(e);
    if (const char* e = getenv("SLAM_PGS_SYRK_SWITCH")) h->syrk_switch = atoi(e);
    h->trace = getenv("SLAM_PGS_TRACE") != nullptr;
    h->host_prof = getenv("SLAM_PGS_HOST_PROF") != nullptr;
    if (const char* e = getenv("SLAM_PGS_FUSED")) h->fused_mode = atoi(e);
    if (const char* e = getenv("SLAM_PGS_SEG")) { const int v = atoi(e); h->seg_len = v <= 0 ? 0 : (v < 2 ? 2 : (v > slam::kPgsSegMaxLen ? slam::kPgsSegMaxLen : v)); }
    h->seg_cur = h->seg_alloc = h->seg_len;
    if (const char* e = getenv("SLAM_PGS_LIST")) h->use_list = atoi(e) != 0;
    if (const char* e = getenv("SLAM_PGS_NOTRIM")) h->p_notrim = atoi(e) ? atoi(e) : 1;
    if (const char* e = getenv("SLAM_PGS_GROUPS")) h->groups = atoi(e);
    if (const char* e = getenv("SLAM_PGS_CHOL_THREADS")) h->chol_threads = atoi(e) == 256 ? 256 : (atoi(e) == 1024 ? 1024 : 0);
    if (const char* e = getenv("SLAM_PGS_CHOL_SWITCH")) h->chol_switch = atoi(e);
    if (const char* e = getenv("SLAM_PGS_CHOL_LL")) h->chol_ll = atoi(e);
    if (const char* e = getenv("SLAM_PGS_SLOTS")) h->slots = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("SLAM_PGS_ITER_ASYNC")) h->iter_async = atoi(e) != 0;
    if (const char* e = getenv("SLAM_PGS_STREAM_DEPTH")) h->stream_depth = atoi(e) >= 1 && atoi(e) < pgs_handle::kRing ? atoi(e) : h->stream_depth;
    hipError_t e = hipStreamCreate(&h->stream);
    if (e != hipSuccess) { delete h; return fail(SLAM_ERR_HIP, "hipStreamCreate -> %s", hipGetErrorString(e)); }
    h->own_stream = true;
    slam::PgsParams& p = h->p;
    memset(&p, 0, sizeof(p));
    p.b_off = 0; p.b_cnt = batch;
    p.B = batch; p.N_max = N_max; p.L_max = L_max; p.KP = k_per_pose; p.LD = h->LD; p.N = 1;
    const size_t B = batch, N = N_max, L = L_max, K = (size_t)N_max * k_per_pose;
    // Every per-instance array has `lanes` slots per instance: slot b is instance b, slot j * B + b its j-th lambda lane (a
    // clone that pgs_solve fills from the instance; PgsParams::lanes_max).  Arrays only the instance itself uses keep B slots.
    {   // the lanes multiply the LM work space (Y alone is 3 N_max x LD doubles per slot): keep them within half of the free memory
        const double nsegx = h->seg_len > 0 ? (double)((N_max - 2) / h->seg_len + 1) : 0.0;
        const double per_slot = 8.0 * (((double)round_up(3 * N_max, 4) + 10.0 * nsegx) * h->LD + (double)h->LD * h->LD + (double)K * 29 + (double)N * 51 + (double)L * 12 + nsegx * (48 + (double)round_up(2 * (L_max < slam::kPgsSegMaxLm ? L_max : slam::kPgsSegMaxLm) + 1, 16) * round_up(2 * (L_max < slam::kPgsSegMaxLm ? L_max : slam::kPgsSegMaxLm) + 1, 16))) +
                                4.0 * ((double)K * 5 + (double)N + (double)L * 6 + nsegx * (4.0 * L + 16));
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            while (h->lanes > 1 && per_slot * (double)B * h->lanes > 0.5 * (double)free_b) h->lanes -= 1;
    }
    const size_t S = B * (size_t)h->lanes;
    p.lanes_max = h->lanes; p.lanes = 1;
    int rc = SLAM_OK;
    auto A = [&](auto** ptr, size_t count) { if (rc == SLAM_OK) rc = dalloc(h, ptr, count); };
    // AC: S slots, and the clones get the instance's content when a solve begins (per-slot element count given)
    auto AC = [&](auto** ptr, size_t per_slot) {
        if (rc == SLAM_OK) rc = dalloc(h, ptr, S * per_slot);
        if (rc == SLAM_OK) h->clone_slabs.push_back({(void*)*ptr, per_slot * sizeof(**ptr)});
    };
    A(&p.pose0, B * N * 3); A(&p.lm0, B * L * 2); A(&p.pose1, B * N * 3); A(&p.lm1, B * L * 2);
    A(&p.ids, B * L); AC(&p.M, 1); A(&p.flags, S);
    AC(&p.cnt, N); AC(&p.mlm, K); AC(&p.mnext, K); AC(&p.lm_head, L); AC(&p.lm_last, L); AC(&p.lm_first, L);
    AC(&p.mb, K); AC(&p.mr, K);
    A(&h->dcmds, N * 2); p.cmds = h->dcmds;
    A(&p.cur, B * 3); A(&p.truth, B * 3); A(&p.truth_hist, B * N * 2);
    AC(&p.pw, N * 3); AC(&p.lw, L * 2); A(&p.pn, S * N * 3); A(&p.ln, S * L * 2);
    A(&p.A, S * N * 9); A(&p.C, S * N * 9); A(&p.gp, S * N * 3); A(&p.E, S * K * 6); A(&p.Wl, S * K * 5);
    AC(&p.evt_start, L + 1); AC(&p.evt_pose, K); AC(&p.slot_pos, K); AC(&p.evt_slot, K); A(&p.Elm, S * K * 6);
    A(&p.PF, S * K * 12); A(&p.lin_ok, S); A(&p.fact_cnt, B); A(&p.seg_umax, B);
    A(&p.D, S * L * 3); A(&p.gl, S * L * 2); A(&p.Linv, S * N * 6); A(&p.G, S * N * 9);
    // segmented elimination: the segments' contributions to their separators' right-hand sides and the separators' rows of Y live
    // behind the pose rows of Y (pgs_kernel.h: yr_rc, yr_sep)
    p.seg_len = h->seg_len; p.seg_on = 0; p.syrk_rows = -1; p.syrk_row0 = 0; p.syrk_first = nullptr;
    p.seg_back_global = getenv("SLAM_PGS_SEG_BACK_GLOBAL") ? atoi(getenv("SLAM_PGS_SEG_BACK_GLOBAL")) : 0;
    p.nseg_max = h->seg_len > 0 ? (N_max - 2) / h->seg_len + 1 : 1;
    p.yr_rc = round_up(3 * N_max, 4);
    p.yr_sep = p.yr_rc + 6 * (int64_t)p.nseg_max;
    p.y_stride = (int64_t)(p.yr_sep + round_up(3 * p.nseg_max, 4)) * h->LD;
    if (h->seg_len > 0) {
        const size_t G = (size_t)p.nseg_max;
        AC(&p.seg_ncol, G); AC(&p.seg_lm, G * L); AC(&p.seg_inv, G * L); AC(&p.seg_evt, G * L); AC(&p.sep_first, L);
        AC(&p.seg_blk, G * (size_t)slam::seg_nb1(L_max)); AC(&p.sep_evt, G * L);
        A(&p.Gs, S * N * 9); A(&p.segout, S * G * 32); A(&p.sepfac, S * G * 16);
        // (segT, the segments' Gram matrices - the largest of these arrays - is allocated by the first solve that runs the segmented order: ensure_segT)
        p.seg_tld = round_up(2 * (L_max < slam::kPgsSegMaxLm ? L_max : slam::kPgsSegMaxLm) + 1, 16);
    }
    A(&p.Y, S * (size_t)p.y_stride); A(&p.S, S * (size_t)h->LD * h->LD);
    A(&p.dl, S * L * 2); A(&p.dp, S * N * 3);
    A(&p.lambda, S); A(&p.error, B); A(&p.cur_error, B); A(&p.err_init, B);
    A(&p.iters, B); A(&p.trials, B); A(&p.state, S + 1); AC(&p.solve_ok, 1); A(&p.n_active, 64); A(&p.alist, S); A(&p.inst_flop, B); A(&p.work, 3);
    A(&p.nl, B); A(&p.nlin, S); A(&p.nerr, S); A(&p.nok, S);
    A(&h->dcount, B); A(&h->dsec, B * 3); A(&h->dout, B);
    A(&h->d_cnt, (size_t)pgs_handle::kMaxGroups * 16); A(&h->d_wait, (size_t)pgs_handle::kMaxGroups);
    if (getenv("SLAM_PGS_PROF")) { A(&p.prof, S * 24); }   // [S][8] chol phase timers, then [S][2][8] per-workgroup stamps of the fused chain
    if (rc != SLAM_OK) { pgs_destroy(h); return rc; }
    hipMemsetAsync(p.truth_hist, 0, sizeof(double) * B * N * 2, h->stream);
    p.dead_slot = (int32_t)S; p.slots_cap = 0; p.n_list_dev = nullptr; p.wait_next = h->d_wait;
    hipMemsetD32Async((hipDeviceptr_t)(p.state + S), 1, 1, h->stream);   // the slot every kernel returns for at entry (pgs_slot)
    hipMemsetAsync(p.cnt, 0, sizeof(int32_t) * B * N, h->stream);
    // effective noise after Filter::readCommonParams (filter.h:105-121)
    double V00, V11, W00, W11;
    if (cfg->replicate_vw_quirk) { V00 = cfg->W_00; V11 = cfg->W_11; W00 = 1.0; W11 = 1.0; }
    else { V00 = cfg->V_00; V11 = cfg->V_11; W00 = cfg->W_00; W11 = cfg->W_11; }
    const double sp[3] = {1.3, 1.3, 1.2};                      // pose_graph.cpp:83
    for (int k = 0; k < 3; ++k) p.w_prior[k] = 1.0 / sp[k];
    p.w_btw[0] = 1.0 / V00; p.w_btw[1] = 1.0 / V00; p.w_btw[2] = 1.0 / V11;   // :52
    p.w_meas[0] = 1.0 / W11; p.w_meas[1] = 1.0 / W00;                          // :54 (bearing, range)
    p.sV00 = cfg->V_00; p.sV11 = cfg->V_11; p.sW00 = cfg->W_00; p.sW11 = cfg->W_11;
    p.d_max = cfg->d_max; p.th_max = cfg->th_max;
    p.range_max = cfg->range_max; p.fov_min = cfg->fov_min; p.fov_max = cfg->fov_max;
    p.seed = h->seed; p.inst0 = 0;
    *out = h;
    return SLAM_OK;
}

int pgs_destroy(pgs_handle* h) {
    if (!h) return SLAM_OK;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (void* ptr : h->allocs) hipFree(ptr);
    for (hipEvent_t e : h->events) hipEventDestroy(e);
    for (hipEvent_t e : h->gevents) hipEventDestroy(e);
    for (hipEvent_t e : h->ring_events) hipEventDestroy(e);
    for (hipEvent_t e : h->async_events) hipEventDestroy(e);
    if (h->tick_stream) hipStreamDestroy(h->tick_stream);
    if (h->h_ring) hipHostFree(h->h_ring);
    for (hipStream_t st : h->gstreams) hipStreamDestroy(st);
    if (h->h_active) hipHostFree(h->h_active);
    if (h->dmeas) hipFree(h->dmeas);
    if (h->p.map) hipFree((void*)h->p.map);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return SLAM_OK;
}

int pgs_set_stream(pgs_handle* h, void* s) {
    TRY(check(h));
    if (h->own_stream && h->stream) { hipStreamSynchronize(h->stream); hipStreamDestroy(h->stream); }
    h->stream = (hipStream_t)s; h->own_stream = false;
    return SLAM_OK;
}
int pgs_set_instance_offset(pgs_handle* h, int64_t first) { TRY(check(h)); h->inst0 = first; h->p.inst0 = first; return SLAM_OK; }
int pgs_set_seed(pgs_handle* h, uint64_t seed) { TRY(check(h)); h->seed = seed; h->p.seed = seed; return SLAM_OK; }

int pgs_set_map(pgs_handle* h, const double* map_xy, int L) {
    TRY(check(h));
    if (!map_xy || L <= 0) return fail(SLAM_ERR_ARG, "bad map");
    if (h->p.map) { hipStreamSynchronize(h->stream); hipFree((void*)h->p.map); h->p.map = nullptr; }
    double* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, sizeof(double) * 2 * (size_t)L));
    HIP_TRY(hipMemcpyAsync(d, map_xy, sizeof(double) * 2 * (size_t)L, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->p.map = d; h->p.L = L;
    return SLAM_OK;
}

int pgs_init(pgs_handle* h, float x0, float y0, float yaw0) {
    TRY(check(h));
    h->timestep = 0; h->p.N = 1;
    h->seg_cur = h->seg_len;
    h->p.prior[0] = x0; h->p.prior[1] = y0; h->p.prior[2] = yaw0;
    HIP_TRY(slam::pgs_launch_init(h->p, x0, y0, yaw0, h->stream));
    h->inited = true;
    return SLAM_OK;
}

int pgs_update_dev(pgs_handle* h, const float cmd[2], const float* d_meas, const int32_t* d_count, int k_stride, const double* d_sec) {
    TRY(check(h));
    if (!h->inited) return fail(SLAM_ERR_STATE, "pgs_init must be called before pgs_update");
    if (!cmd) return fail(SLAM_ERR_ARG, "cmd is NULL");
    if (h->timestep + 1 >= h->N_max) return fail(SLAM_ERR_STATE, "pose capacity N_max = %d reached", h->N_max);
    HIP_TRY(hipMemcpyAsync(h->dcmds + 2 * (size_t)h->timestep, cmd, sizeof(float) * 2, hipMemcpyHostToDevice, h->stream));
    h->p.N = h->timestep + 1;
    HIP_TRY(slam::pgs_launch_append(h->p, d_meas, d_count, k_stride, d_sec, h->stream));
    h->timestep += 1;
    h->p.N = h->timestep + 1;
    return SLAM_OK;
}

int pgs_update(pgs_handle* h, const float cmd[2], const float* meas, const int32_t* count, int k_stride, const double* sec) {
    TRY(check(h));
    if (k_stride < 0 || (k_stride > 0 && (!meas || !count))) return fail(SLAM_ERR_ARG, "bad measurement arguments");
    const int ks = k_stride > 0 ? k_stride : 1;
    TRY(ensure_staging(h, ks));
    if (k_stride > 0) {
        HIP_TRY(hipMemcpyAsync(h->dmeas, meas, sizeof(float) * 3 * (size_t)k_stride * h->B, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->dcount, count, sizeof(int32_t) * (size_t)h->B, hipMemcpyHostToDevice, h->stream));
    } else {
        HIP_TRY(hipMemsetAsync(h->dcount, 0, sizeof(int32_t) * (size_t)h->B, h->stream));
    }
    if (sec) HIP_TRY(hipMemcpyAsync(h->dsec, sec, sizeof(double) * 3 * (size_t)h->B, hipMemcpyHostToDevice, h->stream));
    const int rc = pgs_update_dev(h, cmd, h->dmeas, h->dcount, ks, sec ? h->dsec : nullptr);
    // the staging buffers are reused by the next call and the host arrays are pageable: finish the copies now
    HIP_TRY(hipStreamSynchronize(h->stream));
    return rc;
}

int pgs_run_sim(pgs_handle* h, const float* cmds, int T) {
    TRY(check(h));
    if (!h->inited) return fail(SLAM_ERR_STATE, "pgs_init must be called before pgs_run_sim");
    if (!h->p.map) return fail(SLAM_ERR_STATE, "pgs_set_map must be called before pgs_run_sim");
    if (!cmds || T <= 0) return fail(SLAM_ERR_ARG, "bad command sequence");
    if (h->timestep + T >= h->N_max) return fail(SLAM_ERR_STATE, "timestep %d + %d commands exceed the pose capacity N_max = %d", h->timestep, T, h->N_max);
    HIP_TRY(hipMemcpyAsync(h->dcmds + 2 * (size_t)h->timestep, cmds, sizeof(float) * 2 * (size_t)T, hipMemcpyHostToDevice, h->stream));
    h->p.N = h->timestep + 1;
    HIP_TRY(slam::pgs_launch_run_sim(h->p, T, (uint32_t)h->timestep, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // cmds is a pageable host array
    h->timestep += T;
    h->p.N = h->timestep + 1;
    return SLAM_OK;
}

namespace {

// after pgs_launch_lm_begin on the same stream: the clones (lambda lanes) of the group's instances get the instance's graph,
// event lists, values and scalars - one contiguous copy per array and lane - and start inactive
int clone_instances(pgs_handle* h, const slam::PgsParams& p, hipStream_t stream, bool slabs = true) {
    const size_t B = (size_t)h->B, off = (size_t)p.b_off, cnt = (size_t)p.b_cnt;
    for (int j = 1; j < h->lanes; ++j) {
        if (slabs)
            for (const pgs_handle::Slab& sl : h->clone_slabs)
                HIP_TRY(hipMemcpyAsync((char*)sl.ptr + ((size_t)j * B + off) * sl.bytes, (const char*)sl.ptr + off * sl.bytes, cnt * sl.bytes,
                                       hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(p.state + (size_t)j * B + off), 1, cnt, stream));
    }
    return SLAM_OK;
}

// the slabs of the instances in the group's current list only, in one launch (pgs_clone_kernel): when the first trial with lanes is near
int clone_listed(pgs_handle* h, const slam::PgsParams& p, int n_list, hipStream_t stream) {
    slam::PgsCloneTable t;
    t.n = 0;
    for (const pgs_handle::Slab& sl : h->clone_slabs) {
        if (t.n >= 28 || (sl.bytes & 3)) return fail(SLAM_ERR_STATE, "clone table: %d arrays, %zu bytes per slot", (int)h->clone_slabs.size(), sl.bytes);
        t.ptr[t.n] = sl.ptr; t.words[t.n] = (uint32_t)(sl.bytes / 4); t.n += 1;
    }
    slam::PgsParams q = p;
    q.n_list = n_list;
    HIP_TRY(slam::pgs_launch_clone(q, t, h->lanes, stream));
    return SLAM_OK;
}

// lanes the decide kernel of a trial may hand out, from the number of active instances (of the whole batch) before it
int lanes_for(const pgs_handle* h, int32_t active_hint) {
    return active_hint <= h->lanes_switch_all ? h->lanes : (active_hint <= h->lanes_switch ? (h->lanes < 2 ? h->lanes : 2) : 1);
}

// one tryLambda of the instances [p.b_off, p.b_off + p.b_cnt) on `stream`; `lanes` = the most slots any of them runs in this
// trial (what the previous trial's pgs_decide_kernel reported; 1 for the first)
// The first two operations of a trial - clearing its counters and the linearisation - depend on nothing the host decides
// (kernel variants, lanes), so they are put on the stream BEFORE the host waits for the previous trial's active count: the
// linearize kernel covers the round trip of that wait and of the next launches.  Every slot is covered (inactive ones return).
int prelaunch_trial(pgs_handle* h, slam::PgsParams& p, hipStream_t stream) {
    p.lanes = h->lanes;
    p.use_list = 0;   // the host does not know the list's length yet
    HIP_TRY(hipMemsetAsync(p.n_active, 0, 4 * sizeof(int32_t), stream));
    HIP_TRY(slam::pgs_launch_trial_kernel(p, 0, stream));
    return SLAM_OK;
}

int launch_trial(pgs_handle* h, slam::PgsParams& p, int32_t active_hint, int lanes, int32_t nslots, hipStream_t stream, int trial_index,
                 bool profile, bool prelaunched = false, int force_lanes_next = 0) {
    p.lanes = lanes < 1 ? 1 : (lanes > h->lanes ? h->lanes : lanes);
    p.use_list = h->use_list ? 1 : 0; p.n_list = nslots;   // the slots pgs_decide_kernel (or lm_begin) listed for this trial
    {   // Chain + SYRK fused: NB workgroups per slot, each alone on a CU.  With idle CUs to spare the chain is replicated on up to
        // four of them so that a workgroup's share of the tiles stays in the shadow of the recursion; between one and two rounds of
        // NB = 2 the two-launch path (instance-resident SYRK) is faster; a full batch of 2 x 256 workgroups is two clean rounds.
        const int run = p.use_list ? nslots : p.b_cnt * p.lanes;
        p.fused = 0;
        if (h->fused_ok) {
            if (h->fused_mode >= 2 && h->fused_mode <= 4) p.fused = h->fused_mode;
            else if (4 * run <= h->cus) p.fused = 4;
            else if (3 * run <= h->cus) p.fused = 3;
            else if (2 * run <= h->cus) p.fused = 2;
            else if (run > (h->cus * 2) / 3 && run <= h->cus) p.fused = 2;
        }
    }
    // Few instances left: the per-trial latency counts and spare slots cost little.  Two lanes from `lanes_switch` active
    // instances down (the common streak is one failure, then a success at 10 lambda), all of them from `lanes_switch_all` down.
    p.lanes_next = lanes_for(h, active_hint);
    if (force_lanes_next > 0) p.lanes_next = force_lanes_next;
    active_hint *= p.lanes;   // the kernel variants below are chosen by the number of slots that run (an upper bound), not of instances
    p.syrk_notrim = h->p_notrim;
    p.chol_threads = h->chol_threads ? h->chol_threads : (active_hint > h->chol_switch ? 256 : 1024);
    p.chol_ll = h->chol_ll;
    // 32x32 wavefront tiles by default; the 64x64 variant (more operand reuse, 4x fewer wavefronts) is kept for tuning
    p.syrk_wave_tile = h->syrk_tile ? h->syrk_tile : (active_hint >= h->syrk_switch ? 64 : 32);
    // instance-resident accumulators (tile code 1) from syrk_inst_switch active instances; its staging registers are sized for LD <= 448
    if ((h->syrk_tile == 1 || (!h->syrk_tile && active_hint >= h->syrk_inst_switch)) && p.LD <= 448) p.syrk_wave_tile = 1;
    else if (p.syrk_wave_tile == 1) p.syrk_wave_tile = 32;
    if (profile) { if ((int)h->trial_fused.size() <= trial_index) h->trial_fused.resize(trial_index + 1); h->trial_fused[trial_index] = p.fused; }
    if (!prelaunched) HIP_TRY(hipMemsetAsync(p.n_active, 0, 4 * sizeof(int32_t), stream));
    for (int k = prelaunched ? 1 : 0; k < slam::kPgsTrialKernels; ++k) {
        if (profile) {
            const size_t need = (size_t)(trial_index + 1) * (slam::kPgsTrialKernels + 1);
            while (h->events.size() < need) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); h->events.push_back(e); }
            if (k == 0) HIP_TRY(hipEventRecord(h->events[(size_t)trial_index * (slam::kPgsTrialKernels + 1)], stream));
        }
        HIP_TRY(slam::pgs_launch_trial_kernel(p, k, stream));
        if (profile) HIP_TRY(hipEventRecord(h->events[(size_t)trial_index * (slam::kPgsTrialKernels + 1) + k + 1], stream));
    }
    return SLAM_OK;
}

}  // namespace

int pgs_solve(pgs_handle* h) {
    TRY(check(h));
    if (!h->inited) return fail(SLAM_ERR_STATE, "pgs_init must be called before pgs_solve");
    h->p.N = h->timestep + 1;
    h->p.b_off = 0; h->p.b_cnt = h->B;
    {   // Segmented elimination of the pose chain (pgs_seg_impl.h): the plan kernel lists the landmarks every segment's interior poses
        // see; the path runs when no segment of any instance sees more than kPgsSegMaxLm of them (its columns fit the segment kernels)
        // and the separators fit the separator kernel's staging.  Otherwise - dense visibility on a big map - the sequential chain.
        h->seg_ok = false; h->p.seg_on = 0;
        int SL = h->seg_len > 0 ? h->seg_cur : 0;
        for (;;) {
            // The segment length of this solve: the longest of seg_cur, seg_cur / 2, ... (>= 8 poses) whose every segment sees at most
            // kPgsSegMaxLm landmarks and whose separators fit the separator kernel's staging.
            h->p.seg_len = SL;
            HIP_TRY(slam::pgs_launch_seg_plan(h->p, h->stream));   // (also counts the factors: the grid of the per-factor kernels)
            std::vector<int32_t> U((size_t)h->B), F((size_t)h->B);
            if (SL > 0) HIP_TRY(hipMemcpyAsync(U.data(), h->p.seg_umax, sizeof(int32_t) * (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipMemcpyAsync(F.data(), h->p.fact_cnt, sizeof(int32_t) * (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            int mx = 0, fx = 0;
            for (int32_t f : F) fx = f > fx ? f : fx;
            h->p.nfact_max = fx;
            if (SL <= 0) break;
            for (int32_t u : U) mx = u > mx ? u : mx;
            if (mx <= slam::kPgsSegMaxLm) { h->seg_ok = true; break; }
            const int next = SL / 2;
            // (0x7fffffff: more separators than pgs_sep_kernel stages - shorter segments only add separators)
            if (mx == 0x7fffffff || next < 8 || (h->p.N - 2) / next > slam::kPgsSegMaxSep) break;
            TRY(resize_segments(h, next));
            SL = next;
        }
        if (h->seg_ok) { h->seg_cur = SL; h->seg_used = SL; TRY(ensure_segT(h)); }
        else {   // (a graph only grows: what no segment length holds now, none will hold later - the next solves go straight to the sequential chain)
            h->p.seg_len = h->seg_len > 0 ? h->seg_alloc : 0; h->seg_used = 0; h->seg_cur = 0;
        }
        h->p.seg_on = h->seg_ok ? 1 : 0;
    }
    {   // chain + SYRK in one launch (Y stays in LDS) is possible while the lower triangle of every instance fits the 72 wavefront
        // tiles of pgs_chain_syrk_kernel, a column of Y per lane (2M + 1 <= 448) and its event staging (32 factor slots per pose)
        h->fused_ok = false; h->p.fused = 0;
        if (!h->seg_ok && h->fused_mode != 0 && h->p.LD <= 448 && h->p.KP <= 32) {
            std::vector<int32_t> M((size_t)h->B);
            HIP_TRY(hipMemcpyAsync(M.data(), h->p.M, sizeof(int32_t) * (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            int mx = 0;
            for (int32_t m : M) mx = m > mx ? m : mx;
            const int nt = (2 * mx + 31) / 32;
            h->fused_ok = nt * (nt + 1) / 2 <= 72;
        }
    }
    HIP_TRY(hipMemsetAsync(h->p.work, 0, 3 * sizeof(double), h->stream));
    h->path_ms[0] = h->path_ms[1] = h->path_ms[2] = 0.0;
    // (round 5: two groups from batch 128 on - with the segmented elimination a trial's kernels are short enough for two LM loops to fill
    // each other's gaps: 5.20 -> 5.67 k solves/s at batch 256 (three groups 5.47 k, four 3.57 k), 3.49 -> 3.64 k at batch 128, 7.39 k at
    // 1024 (three: 7.59 k); profiles/r05_pgs/groups_and_lanes.txt)
    int G = h->groups > 0 ? h->groups : (h->B >= 128 ? 2 : 1);
    if (G > pgs_handle::kMaxGroups) G = pgs_handle::kMaxGroups;
    if (G > h->B) G = h->B;
    if (h->profiling) G = 1;   // per-kernel timing wants the kernels of one stream back to back
    const bool profile = h->profiling;
    // The groups' streams must not share a hardware queue: two LM loops whose launches sit in ONE in-order queue wait for each other's
    // queued trials at every host synchronisation (measured inside bench.py's driver command, where the earlier legs' streams shift the
    // runtime's stream -> queue assignment: 3.40 k solves/s with two groups against 5.21 k with one and 5.66 k with two on distinct
    // queues; profiles/r05_pgs/hw_queues.txt).  The runtime keeps separate queues per stream PRIORITY, so the groups alternate between
    // the priority levels the device offers (SLAM_PGS_GROUP_PRIO=0: all at the default priority, the behaviour before).
    while (G > 1 && (int)h->gstreams.size() < G) {
        hipStream_t st;
        int lo = 0, hi = 0;
        static const bool use_prio = !(getenv("SLAM_PGS_GROUP_PRIO") && atoi(getenv("SLAM_PGS_GROUP_PRIO")) == 0);
        if (use_prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
            const int nlev = lo - hi + 1, g = (int)h->gstreams.size();   // lo = least priority (numerically greatest), hi = greatest
            HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi + (g % nlev)));
        } else {
            (void)hipGetLastError();
            HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        }
        h->gstreams.push_back(st);
    }
    while ((int)h->gevents.size() < G + 1) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->gevents.push_back(e); }
    if (!h->h_active) HIP_TRY(hipHostMalloc((void**)&h->h_active, sizeof(int32_t) * 64, hipHostMallocDefault));
    // streaming: `slots` graphs in flight over all groups; a group whose share of the slots covers its graphs runs lockstep as before
    const int per = (h->B + G - 1) / G;
    const int cap_g = (h->slots > 0 && !profile) ? (h->slots + G - 1) / G : 0;
    if (cap_g > 0) {
        if (!h->h_ring) HIP_TRY(hipHostMalloc((void**)&h->h_ring, sizeof(int32_t) * 8 * pgs_handle::kRing * pgs_handle::kMaxGroups, hipHostMallocDefault));
        while ((int)h->ring_events.size() < pgs_handle::kRing * pgs_handle::kMaxGroups) {
            hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->ring_events.push_back(e);
        }
    }
    h->timeline.assign(G, {});
    if (G > 1) HIP_TRY(hipEventRecord(h->gevents[G], h->stream));   // everything queued on the handle's stream so far comes first
    std::vector<slam::PgsParams> gp(G, h->p);
    std::vector<int> gtrials(G, 0), grc(G, SLAM_OK);
    std::vector<char> gdone(G, 0);
    auto gstream = [&](int g) { return G > 1 ? h->gstreams[g] : h->stream; };
    for (int g = 0; g < G; ++g) {
        slam::PgsParams& q = gp[g];
        q.b_off = g * per;
        q.b_cnt = (h->B - g * per) < per ? (h->B - g * per) : per;
        q.n_active = h->p.n_active + 4 * g;
        q.alist = h->p.alist + (size_t)g * per * h->lanes;
        q.wait_next = h->d_wait + g;
        q.slots_cap = (cap_g > 0 && cap_g < q.b_cnt) ? cap_g : 0;
        q.n_list_dev = nullptr;
        if (q.b_cnt <= 0) { gdone[g] = 1; continue; }
        if (G > 1) HIP_TRY(hipStreamWaitEvent(gstream(g), h->gevents[G], 0));
        HIP_TRY(slam::pgs_launch_lm_begin(q, gstream(g)));
        TRY(clone_instances(h, q, gstream(g), /*slabs=*/!h->use_list || profile));   // (with the slot list: copied on demand, for the listed instances)
    }
    // One host thread per group drives its LM loop; the HIP runtime is thread-safe and the groups touch disjoint instance ranges.
    //  * streaming phase (slots_cap > 0): the list of every trial is refilled on the device, its length read on the device, so the host
    //    enqueues trials `stream_depth` ahead and only looks at the counters of the trial that far back - to learn that the queue of
    //    waiting graphs is empty and few are left (then the lockstep loop below takes over, with its lambda lanes), or none at all.
    //  * lockstep phase: launch a trial, wait for its active count, choose the next trial's kernel variants and lanes from it.  The first
    //    two operations of a trial (counters, linearisation) are enqueued before that wait (prelaunch_trial).
    auto drive = [&](int g) -> int {
        HIP_TRY(hipSetDevice(h->device));
        slam::PgsParams& q = gp[g];
        hipStream_t st = gstream(g);
        std::vector<int32_t>& tl = h->timeline[g];
        int32_t act[3] = {q.b_cnt * G, 1, q.b_cnt};   // active instances (scaled to the batch), lanes and active slots of the next trial
        int trials = 0;
        if (q.slots_cap > 0) {
            constexpr int R = pgs_handle::kRing;
            const int cap = q.slots_cap, D = h->stream_depth, wend = q.b_off + q.b_cnt;
            int32_t* dcnt = h->d_cnt + (size_t)g * 16;
            int32_t* ring = h->h_ring + (size_t)g * R * 8;
            hipEvent_t* ev = &h->ring_events[(size_t)g * R];
            int32_t* const n_active0 = q.n_active;
            // "the counters trial -1 left": the first `cap` graphs run (pgs_lm_begin_kernel listed them), the cursor stands behind them
            HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(dcnt + 2), cap, 1, st));
            HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)q.wait_next, q.b_off + cap, 1, st));
            int t = 0, seen = 0;   // trials enqueued / trials whose counters the host has read
            bool leave = false;
            int32_t last[8] = {cap * G, 1, cap, 0, q.b_off + cap, 0, 0, 0};
            tl.push_back(cap);
            auto read_one = [&]() -> int {
                HIP_TRY(hipEventSynchronize(ev[seen % R]));
                memcpy(last, ring + 8 * (seen % R), sizeof(last));
                seen += 1;
                if (h->trace) fprintf(stderr, "pgs group %d trial %d (streaming): active %d, slots %d, next waiting %d of %d\n", g, seen - 1, last[0], last[2], last[4], wend);
                tl.push_back(last[2]);
                // nothing left, or nothing waiting and few enough running that the lanes pay: the lockstep loop finishes the group
                if (last[0] == 0 || (last[4] >= wend && last[0] * G <= h->lanes_switch)) leave = true;
                return SLAM_OK;
            };
            // (the trial cap is per graph - pgs_decide_kernel applies it while graphs stream; the launches only need a bound that cannot bind first)
            q.max_trials = h->max_trials;
            const long long launch_cap = ((long long)q.b_cnt / cap + 2) * h->max_trials;
            while (!leave && t < launch_cap) {
                q.n_list_dev = dcnt + 8 * (t & 1) + 2;
                q.n_active = dcnt + 8 * ((t + 1) & 1);
                TRY(launch_trial(h, q, cap * G, 1, cap, st, t, false, false, 1));
                HIP_TRY(hipMemcpyAsync(ring + 8 * (t % R), q.n_active, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipEventRecord(ev[t % R], st));
                t += 1;
                while (!leave && t - seen >= D) TRY(read_one());
            }
            while (seen < t) TRY(read_one());   // the trials already enqueued are real ones (or return at entry): their counters are the state to go on from
            tl.pop_back();                      // (the last entry is the list of a trial the lockstep loop launches and records itself)
            trials = t;
            q.slots_cap = 0; q.n_list_dev = nullptr; q.n_active = n_active0;
            act[0] = last[0] * G; act[1] = last[1] > 0 ? last[1] : 1; act[2] = last[2];
            // (graphs still waiting at the trial cap keep state 2: pgs_lm_end_kernel flags them NOT_CONVERGED)
        }
        const bool pipe = !profile;   // per-kernel timing wants every kernel of a trial between its own events
        int32_t* hact = h->h_active + 4 * g;
        const int trials_end = trials + h->max_trials;   // (after a streaming phase: that many more launches for the graphs still running)
        if (act[0] > 0 && trials < trials_end) {
            bool pre = false;   // the phase's first trial runs over the list it was handed; later ones have their first kernels enqueued ahead
            bool cloned = !h->use_list || profile || h->lanes <= 1;
            for (;;) {
                // the first trial whose decide step may hand out lanes: the instances it lists (every later list is a subset) get their clones now
                if (!cloned && lanes_for(h, act[0]) > 1) { TRY(clone_listed(h, q, act[2], st)); cloned = true; }
                tl.push_back(act[2]);
                timespec ta, tb, tc;
                if (h->host_prof) clock_gettime(CLOCK_MONOTONIC, &ta);
                TRY(launch_trial(h, q, act[0], act[1], act[2], st, trials, profile, pre));
                HIP_TRY(hipMemcpyAsync(hact, q.n_active, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipEventRecord(h->gevents[g], st));
                if (pipe) { TRY(prelaunch_trial(h, q, st)); pre = true; }   // the next trial's linearisation runs while the host waits below
                if (h->host_prof) clock_gettime(CLOCK_MONOTONIC, &tb);
                HIP_TRY(hipEventSynchronize(h->gevents[g]));
                if (h->host_prof) {
                    clock_gettime(CLOCK_MONOTONIC, &tc);
                    h->host_launch_ms += (tb.tv_sec - ta.tv_sec) * 1e3 + (tb.tv_nsec - ta.tv_nsec) * 1e-6;
                    h->host_wait_ms += (tc.tv_sec - tb.tv_sec) * 1e3 + (tc.tv_nsec - tb.tv_nsec) * 1e-6;
                }
                act[0] = hact[0] * G; act[1] = hact[1]; act[2] = hact[2];
                trials += 1;
                if (h->trace) {
                    static thread_local double t_prev = 0.0;
                    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
                    const double now = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
                    fprintf(stderr, "pgs group %d trial %d: %.2f ms, lanes %d -> active %d, lanes next %d\n", g, trials - 1, trials > 1 ? now - t_prev : 0.0, (int)q.lanes, (int)hact[0], (int)act[1]);
                    t_prev = now;
                }
                if (hact[0] == 0 || trials >= trials_end) break;
            }
        }
        gtrials[g] = trials;
        HIP_TRY(slam::pgs_launch_lm_end(q, st));
        if (G > 1) HIP_TRY(hipEventRecord(h->gevents[g], st));
        return SLAM_OK;
    };
    std::vector<std::thread> workers;
    for (int g = 1; g < G; ++g)
        if (!gdone[g]) workers.emplace_back([&, g]() { grc[g] = drive(g); });
    if (!gdone[0]) grc[0] = drive(0);
    for (auto& w : workers) w.join();
    int max_trials = 0;
    for (int g = 0; g < G; ++g) {
        if (grc[g] != SLAM_OK) return grc[g];
        if (G > 1 && gp[g].b_cnt > 0) HIP_TRY(hipStreamWaitEvent(h->stream, h->gevents[g], 0));   // the handle's stream continues after every group
        max_trials = gtrials[g] > max_trials ? gtrials[g] : max_trials;
    }
    h->last_trials = max_trials;
    if (profile) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int k = 0; k < slam::kPgsTrialKernels; ++k) h->kernel_ms[k] = 0.0;
        for (int t = 0; t < max_trials; ++t)
            for (int k = 0; k < slam::kPgsTrialKernels; ++k) {
                float ms = 0.f;
                const size_t e0 = (size_t)t * (slam::kPgsTrialKernels + 1) + k;
                HIP_TRY(hipEventElapsedTime(&ms, h->events[e0], h->events[e0 + 1]));
                h->kernel_ms[k] += ms;
                const int tf = t < (int)h->trial_fused.size() ? h->trial_fused[t] : 0;
                if (k == 1 && tf) h->path_ms[1] += ms;      // chain + SYRK in one launch
                if (k == 2 && !tf) h->path_ms[h->seg_ok ? 2 : 0] += ms;     // the SYRK launch(es) of the two-launch path / of the segmented path
                if (h->trace) fprintf(stderr, "%s%.3f%s", k == 0 ? "pgs trial kernels (ms): " : " ", ms, k + 1 == slam::kPgsTrialKernels ? "\n" : "");
            }
    }
    return SLAM_OK;
}

// Streaming: at most `slots` graphs of the batch in flight (0 = lockstep, every graph from the first trial on).  Results do not depend on it.
int pgs_set_slots(pgs_handle* h, int slots) { TRY(check(h)); h->slots = slots < 0 ? 0 : slots; return SLAM_OK; }
// The last solve's running slots per trial of solve group `group`: slots[0 .. *n) (at most cap entries written); *groups = number of groups.
int pgs_last_solve_timeline(pgs_handle* h, int group, int32_t* slots, int cap, int32_t* n, int32_t* groups) {
    TRY(check(h));
    if (groups) *groups = (int32_t)h->timeline.size();
    if (group < 0 || group >= (int)h->timeline.size()) { if (n) *n = 0; return SLAM_OK; }
    const std::vector<int32_t>& tl = h->timeline[group];
    if (n) *n = (int32_t)tl.size();
    if (slots) for (int i = 0; i < (int)tl.size() && i < cap; ++i) slots[i] = tl[i];
    return SLAM_OK;
}

// number of solve groups (0 = automatic: 2 from 512 instances)
int pgs_set_groups(pgs_handle* h, int groups) { TRY(check(h)); h->groups = groups < 0 ? 0 : groups; return SLAM_OK; }

int pgs_adopt_result(pgs_handle* h) {
    TRY(check(h));
    HIP_TRY(slam::pgs_launch_adopt(h->p, h->stream));
    return SLAM_OK;
}

namespace {
// solve_graph_every_iteration WITHOUT a batch-wide barrier per tick (pgs_kernel.h "asynchronous ticks"): a graph's tick t + 1 depends on
// nothing but its own tick t, so every graph walks through its ticks at its own pace - one LM trial of ALL unfinished graphs per round of
// launches, whatever tick each is at; the graphs whose solve converged in a round are advanced (result, adopt, simulator tick, append, plan,
// begin) on a second stream beside the next round and rejoin the one after.  The lockstep tick loop launches, per tick, as many trials as
// the tick's slowest graph needs (27.5 at BASELINE configs[4] against a mean of 7.5 consumed); here a graph pays its own trials + 1 per tick.
// The host only sizes grids (from the monotone counters the decide kernel forwards) and enqueues rounds `stream_depth` ahead.
int run_every_iteration_async(pgs_handle* h, int T) {
    constexpr int R = pgs_handle::kRing;
    const int D = h->stream_depth, B = h->B;
    const size_t S = (size_t)B * h->lanes;
    if (!h->d_Nv) TRY(dalloc(h, &h->d_Nv, S + 1));
    if (!h->d_mono) TRY(dalloc(h, &h->d_mono, 2));
    if (!h->h_ring) HIP_TRY(hipHostMalloc((void**)&h->h_ring, sizeof(int32_t) * 8 * pgs_handle::kRing * pgs_handle::kMaxGroups, hipHostMallocDefault));
    while ((int)h->ring_events.size() < pgs_handle::kRing * pgs_handle::kMaxGroups) {
        hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->ring_events.push_back(e);
    }
    while ((int)h->async_events.size() < 2 * R + 1) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->async_events.push_back(e); }
    if (!h->tick_stream) HIP_TRY(hipStreamCreateWithFlags(&h->tick_stream, hipStreamNonBlocking));
    hipStream_t sa = h->stream, sb = h->tick_stream;
    hipEvent_t* evDec = &h->async_events[0];
    hipEvent_t* evAdv = &h->async_events[R];
    hipEvent_t evStart = h->async_events[2 * R];
    const int N0 = h->timestep + 1;
    TRY(ensure_segT(h));
    slam::PgsParams q = h->p;
    q.seg_len = h->seg_cur;
    q.async_ticks = 1; q.seg_on = 1; q.Nv = h->d_Nv; q.T_end = h->timestep + T; q.split_decide = 1; q.max_trials = h->max_trials;
    q.mono = h->d_mono; q.tick_acc = h->d_tick; q.tick_flop = h->d_tick_flop;
    q.b_off = 0; q.b_cnt = B; q.slots_cap = 0; q.N = N0;
    h->seg_ok = true; h->fused_ok = false;
    int32_t* dcnt = h->d_cnt;                        // group 0's counter blocks
    int32_t* ring = h->h_ring;
    hipEvent_t* ev = &h->ring_events[0];
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)h->d_Nv, N0, S + 1, sa));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)q.state, 5, (size_t)B, sa));                      // first tick: nothing to adopt
    if (S > (size_t)B) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(q.state + B), 1, S - (size_t)B, sa));   // the lambda lanes are off
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)h->d_mono, 0, 1, sa));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(h->d_mono + 1), N0, 1, sa));
    HIP_TRY(hipMemsetAsync(dcnt, 0, 16 * sizeof(int32_t), sa));
    HIP_TRY(hipMemsetAsync(h->p.work, 0, 3 * sizeof(double), sa));
    HIP_TRY(hipEventRecord(evStart, sa));
    HIP_TRY(hipStreamWaitEvent(sb, evStart, 0));
    HIP_TRY(slam::pgs_launch_tick(q, sb));                                                       // every graph's first tick
    HIP_TRY(hipEventRecord(evAdv[R - 1], sb));
    int k = 0, seen = 0;
    bool leave = false;
    int32_t last[8] = {B, 1, 0, 0, 0, 0, N0 + 1, 0};
    int n_known = N0 + 1, f_known = 0;
    const long long cap = (long long)T * h->max_trials + 16;
    auto read_one = [&]() -> int {
        HIP_TRY(hipEventSynchronize(ev[seen % R]));
        memcpy(last, ring + 8 * (seen % R), sizeof(last));
        seen += 1;
        if (last[5] > f_known) f_known = last[5];
        if (last[6] > n_known) n_known = last[6];
        if (h->trace) fprintf(stderr, "pgs async round %d: unfinished %d, listed %d, most factors %d, most poses %d\n", seen - 1, last[0], last[2], last[5], last[6]);
        if (last[0] == 0) leave = true;
        return SLAM_OK;
    };
    std::vector<int32_t>& tl = (h->timeline.assign(1, {}), h->timeline[0]);
    while (!leave && k < cap) {
        q.n_list_dev = dcnt + 8 * (k & 1) + 2;
        q.n_active = dcnt + 8 * ((k + 1) & 1);
        // grids: the most poses / factors a graph can have when this round runs - every round ahead of the host's knowledge may have advanced it by a tick
        q.N = n_known + D + 2 < h->N_max ? n_known + D + 2 : h->N_max;
        q.nfact_max = f_known + h->KP * (D + 2);
        TRY(launch_trial(h, q, B, 1, B, sa, k, false, false, 1));
        HIP_TRY(hipStreamWaitEvent(sa, evAdv[(k + R - 1) % R], 0));      // the graphs advanced beside this round are ready to be listed
        HIP_TRY(slam::pgs_launch_trial_kernel(q, 6, sa));                // decide
        HIP_TRY(hipEventRecord(evDec[k % R], sa));
        HIP_TRY(hipStreamWaitEvent(sb, evDec[k % R], 0));
        HIP_TRY(slam::pgs_launch_tick(q, sb));                           // ... and those that converged in it advance beside the next one
        HIP_TRY(hipEventRecord(evAdv[k % R], sb));
        HIP_TRY(hipMemcpyAsync(ring + 8 * (k % R), q.n_active, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, sa));
        HIP_TRY(hipEventRecord(ev[k % R], sa));
        k += 1;
        while (!leave && k - seen >= D) { TRY(read_one()); tl.push_back(last[2]); }
    }
    while (seen < k) { TRY(read_one()); tl.push_back(last[2]); }
    HIP_TRY(hipStreamWaitEvent(sa, evAdv[(k + R - 1) % R], 0));          // the handle's stream continues after the last tick step
    h->async_trials = k;
    h->last_trials = k;
    if (last[0] != 0) return fail(SLAM_ERR_STATE, "asynchronous ticks: %d graphs unfinished after %d rounds of launches", last[0], k);
    return SLAM_OK;
}
}  // namespace

// solve_graph_every_iteration with the simulator on the device: T x { get_cmd + NaiveFilter + graph append (one tick of pgs_run_sim),
// pgs_solve, pgs_adopt_result }.  SLAM_PGS_ITER_PROF=1: the host clock per phase with a stream synchronisation after each (phase table only).
int pgs_run_sim_every_iteration(pgs_handle* h, const float* cmds, int T, int32_t* counts) {
    TRY(check(h));
    if (!h->inited) return fail(SLAM_ERR_STATE, "pgs_init must be called before pgs_run_sim_every_iteration");
    if (!h->p.map) return fail(SLAM_ERR_STATE, "pgs_set_map must be called before pgs_run_sim_every_iteration");
    if (!cmds || T <= 0) return fail(SLAM_ERR_ARG, "bad command sequence");
    if (h->timestep + T >= h->N_max) return fail(SLAM_ERR_STATE, "timestep %d + %d commands exceed the pose capacity N_max = %d", h->timestep, T, h->N_max);
    HIP_TRY(hipMemcpyAsync(h->dcmds + 2 * (size_t)h->timestep, cmds, sizeof(float) * 2 * (size_t)T, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // cmds is a pageable host array
    if (!h->d_tick) TRY(dalloc(h, &h->d_tick, (size_t)h->B * 2));
    if (!h->d_tick_flop) TRY(dalloc(h, &h->d_tick_flop, (size_t)h->B * 2));
    HIP_TRY(hipMemsetAsync(h->d_tick, 0, sizeof(int32_t) * 2 * (size_t)h->B, h->stream));
    HIP_TRY(hipMemsetAsync(h->d_tick_flop, 0, sizeof(double) * 2 * (size_t)h->B, h->stream));
    const bool prof = getenv("SLAM_PGS_ITER_PROF") != nullptr;
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    for (int k = 0; k < 4; ++k) h->iter_ms[k] = 0.0;
    h->iter_trials = 0;
    // asynchronous ticks need the segmented elimination (its plan is checked on the device) and its separators within the staging of
    // pgs_sep_kernel for the whole run; per-kernel profiling and the phase table want the lockstep loop
    if (h->iter_async && !prof && !h->profiling && h->seg_len > 0 && (h->timestep + T - 1) / h->seg_len <= slam::kPgsSegMaxSep && h->use_list) {
        TRY(run_every_iteration_async(h, T));
        h->iter_trials = h->async_trials;
        h->timestep += T;
        h->p.N = h->timestep + 1;
        if (counts) HIP_TRY(hipMemcpyAsync(counts, h->d_tick, sizeof(int32_t) * 2 * (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        std::vector<int32_t> fl((size_t)h->B);
        HIP_TRY(hipMemcpy(fl.data(), h->p.flags, sizeof(int32_t) * (size_t)h->B, hipMemcpyDeviceToHost));
        int nlim = 0;
        for (int32_t f : fl) nlim += (f & slam::PGS_FLAG_SEG_LIMIT) ? 1 : 0;
        if (nlim) return fail(SLAM_ERR_UNSUPPORTED, "%d graphs have a segment that sees more than %d landmarks (PGS_FLAG_SEG_LIMIT): they stopped at their last "
                              "adopted result; re-run with SLAM_PGS_ITER_ASYNC=0 (the lockstep tick loop falls back to the sequential chain)", nlim, slam::kPgsSegMaxLm);
        return SLAM_OK;
    }
    // One solve group per tick unless the caller chose a number: a tick's solve is a handful of latency-bound trials - two groups gain 2 % in
    // a process of their own (15.2 k against 14.9 k graph-ticks/s) and lose 20 % when their streams land on one hardware queue behind other
    // handles' streams (bench.py's secondary leg: 12.3 k; the stream -> queue assignment of the runtime, profiles/r05_pgs/hw_queues.txt).
    struct GroupsGuard { pgs_handle* h; int saved; ~GroupsGuard() { h->groups = saved; } } guard{h, h->groups};
    if (h->groups == 0) h->groups = 1;
    for (int t = 0; t < T; ++t) {
        double t0 = prof ? now_ms() : 0.0;
        h->p.N = h->timestep + 1;
        HIP_TRY(slam::pgs_launch_run_sim(h->p, 1, (uint32_t)h->timestep, h->stream));
        h->timestep += 1;
        h->p.N = h->timestep + 1;
        if (prof) { HIP_TRY(hipStreamSynchronize(h->stream)); const double t1 = now_ms(); h->iter_ms[0] += t1 - t0; t0 = t1; }
        TRY(pgs_solve(h));
        h->iter_trials += h->last_trials;
        if (prof) { HIP_TRY(hipStreamSynchronize(h->stream)); const double t1 = now_ms(); h->iter_ms[1] += t1 - t0; t0 = t1; }
        h->p.tick_acc = h->d_tick; h->p.tick_flop = h->d_tick_flop;
        const hipError_t e = slam::pgs_launch_adopt(h->p, h->stream);
        h->p.tick_acc = nullptr; h->p.tick_flop = nullptr;
        if (e != hipSuccess) return fail(SLAM_ERR_HIP, "pgs_launch_adopt -> %s", hipGetErrorString(e));
        if (prof) { HIP_TRY(hipStreamSynchronize(h->stream)); const double t1 = now_ms(); h->iter_ms[2] += t1 - t0; }
    }
    if (counts) {
        HIP_TRY(hipMemcpyAsync(counts, h->d_tick, sizeof(int32_t) * 2 * (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    if (h->host_prof) fprintf(stderr, "pgs host clock over the run's trials: %.0f ms enqueuing, %.0f ms waiting for the trials' counters\n", h->host_launch_ms, h->host_wait_ms);
    return SLAM_OK;
}
// Host-clock phase times of the last pgs_run_sim_every_iteration under SLAM_PGS_ITER_PROF=1: ms in {simulator + append, solve, adopt}, and the
// LM trials launched over all ticks.
int pgs_last_iter_phases(pgs_handle* h, double ms[6]) {
    TRY(check(h));
    if (!ms) return fail(SLAM_ERR_ARG, "NULL output");
    ms[0] = h->iter_ms[0]; ms[1] = h->iter_ms[1]; ms[2] = h->iter_ms[2]; ms[3] = (double)h->iter_trials;
    ms[4] = ms[5] = 0.0;
    if (h->d_tick_flop) {   // algorithmic FLOP of the call's trials, summed over the batch: SYRK | Cholesky
        HIP_TRY(hipStreamSynchronize(h->stream));
        std::vector<double> f((size_t)h->B * 2);
        HIP_TRY(hipMemcpy(f.data(), h->d_tick_flop, sizeof(double) * f.size(), hipMemcpyDeviceToHost));
        for (int b = 0; b < h->B; ++b) { ms[4] += f[2 * (size_t)b]; ms[5] += f[2 * (size_t)b + 1]; }
    }
    return SLAM_OK;
}

int pgs_get_graph(pgs_handle* h, int inst, int which, double* poses, double* lms, int32_t* timestep, int32_t* M, int32_t* ids) {
    TRY(check(h));
    if (inst < 0 || inst >= h->B) return fail(SLAM_ERR_ARG, "instance %d out of range", inst);
    HIP_TRY(hipStreamSynchronize(h->stream));
    int32_t m = 0;
    HIP_TRY(hipMemcpy(&m, h->p.M + inst, sizeof(int32_t), hipMemcpyDeviceToHost));
    const double* ps = (which ? h->p.pose1 : h->p.pose0) + (size_t)inst * h->N_max * 3;
    const double* ls = (which ? h->p.lm1 : h->p.lm0) + (size_t)inst * h->L_max * 2;
    if (poses) HIP_TRY(hipMemcpy(poses, ps, sizeof(double) * 3 * (size_t)(h->timestep + 1), hipMemcpyDeviceToHost));
    if (lms && m > 0) HIP_TRY(hipMemcpy(lms, ls, sizeof(double) * 2 * (size_t)m, hipMemcpyDeviceToHost));
    if (ids && m > 0) HIP_TRY(hipMemcpy(ids, h->p.ids + (size_t)inst * h->L_max, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost));
    if (timestep) *timestep = h->timestep;
    if (M) *M = m;
    return SLAM_OK;
}

int pgs_get_connections(pgs_handle* h, int inst, int32_t* conn, int cap, int32_t* n) {
    TRY(check(h));
    if (inst < 0 || inst >= h->B) return fail(SLAM_ERR_ARG, "instance %d out of range", inst);
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int N = h->timestep + 1;
    std::vector<int32_t> cnt(N), mlm((size_t)N * h->KP);
    HIP_TRY(hipMemcpy(cnt.data(), h->p.cnt + (size_t)inst * h->N_max, sizeof(int32_t) * N, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mlm.data(), h->p.mlm + (size_t)inst * h->N_max * h->KP, sizeof(int32_t) * (size_t)N * h->KP, hipMemcpyDeviceToHost));
    int nc = 0;
    for (int i = 0; i < N; ++i)
        for (int s = 0; s < cnt[i]; ++s) {
            const int32_t v = mlm[(size_t)i * h->KP + s];
            if (conn && nc < cap) { conn[2 * nc] = i; conn[2 * nc + 1] = (v & slam::kPgsFirstBit) ? -1 : v; }
            nc += 1;
        }
    if (n) *n = nc;
    return SLAM_OK;
}

int pgs_get_stats(pgs_handle* h, int32_t* iterations, int32_t* trials, int32_t* flags, double* err_init, double* err_final, double* lambda) {
    TRY(check(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t B = h->B;
    if (iterations) HIP_TRY(hipMemcpy(iterations, h->p.iters, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
    if (trials) HIP_TRY(hipMemcpy(trials, h->p.trials, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
    if (flags) HIP_TRY(hipMemcpy(flags, h->p.flags, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
    if (err_init) HIP_TRY(hipMemcpy(err_init, h->p.err_init, sizeof(double) * B, hipMemcpyDeviceToHost));
    if (err_final) HIP_TRY(hipMemcpy(err_final, h->p.error, sizeof(double) * B, hipMemcpyDeviceToHost));
    if (lambda) HIP_TRY(hipMemcpy(lambda, h->p.lambda, sizeof(double) * B, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

int pgs_error_stats(pgs_handle* h, int which, double* out) {
    TRY(check(h));
    if (!out) return fail(SLAM_ERR_ARG, "NULL output");
    h->p.N = h->timestep + 1;
    HIP_TRY(slam::pgs_launch_avg_error(h->p, which, h->dout, h->stream));
    HIP_TRY(hipMemcpyAsync(out, h->dout, sizeof(double) * (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAM_OK;
}

int pgs_last_solve_work(pgs_handle* h, double* syrk_flop, int32_t* trials_launched) {
    TRY(check(h));
    // per trial and instance: what pgs_lm_begin_kernel priced one Schur-complement SYRK of the instance at (inst_flop: every stored
    // element of the lower triangle of S_ext over the rows of Y that can be non-zero in it - by the elimination order the solve ran,
    // sequential or segmented; independent of the kernels' tiling)
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t B = h->B;
    std::vector<int32_t> tr(B);
    std::vector<double> fl(B);
    HIP_TRY(hipMemcpy(tr.data(), h->p.trials, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(fl.data(), h->p.inst_flop, sizeof(double) * B, hipMemcpyDeviceToHost));
    double tot = 0.0;
    for (size_t b = 0; b < B; ++b) tot += fl[b] * tr[b];
    if (syrk_flop) *syrk_flop = tot;
    if (trials_launched) *trials_launched = h->last_trials;
    return SLAM_OK;
}

// The last PROFILED solve by path: out = {algorithmic SYRK FLOP of the trials that ran pgs_syrk_*_kernel, of the trials that ran
// pgs_chain_syrk_kernel, ms in those SYRK launches, ms in those fused launches, FLOP of the trials of the segmented elimination, ms in
// its SYRK launches (tile kernel on the separator rows + pgs_seg_syrk_kernel), 1 if the solve ran the segmented elimination, segment length}
int pgs_last_solve_paths_v2(pgs_handle* h, double* out, int n) {
    TRY(check(h));
    if (!out || n < 0 || n > 8) return fail(SLAM_ERR_ARG, "bad output (n = %d, at most 8 entries)", n);
    HIP_TRY(hipStreamSynchronize(h->stream));
    double w[3];
    HIP_TRY(hipMemcpy(w, h->p.work, 3 * sizeof(double), hipMemcpyDeviceToHost));
    const double v[8] = {w[0], w[1], h->path_ms[0], h->path_ms[1], w[2], h->path_ms[2], h->seg_ok ? 1.0 : 0.0, (double)(h->seg_ok ? h->seg_used : h->seg_len)};
    for (int i = 0; i < n; ++i) out[i] = v[i];
    return SLAM_OK;
}
int pgs_last_solve_paths(pgs_handle* h, double out[4]) { return pgs_last_solve_paths_v2(h, out, 4); }
int pgs_set_profiling(pgs_handle* h, int on) { TRY(check(h)); h->profiling = on != 0; return SLAM_OK; }
int pgs_last_solve_kernel_ms(pgs_handle* h, double ms[6]) {
    TRY(check(h));
    if (!ms) return fail(SLAM_ERR_ARG, "NULL output");
    for (int k = 0; k < slam::kPgsTrialKernels; ++k) ms[k] = h->kernel_ms[k];
    return SLAM_OK;
}
// debug only (not part of the ABI header): phase timers of the last chol launch, [batch][8] ticks of the 100 MHz clock
int pgs_debug_prof(pgs_handle* h, unsigned long long* out) {
    TRY(check(h));
    if (!h->p.prof) return fail(SLAM_ERR_STATE, "set SLAM_PGS_PROF before pgs_create");
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out, h->p.prof, sizeof(unsigned long long) * 8 * (size_t)h->B, hipMemcpyDeviceToHost));
    return SLAM_OK;
}
// debug only: begin / end / HW_ID / XCC_ID of the two workgroups of every instance in the last fused chain launch + phase times, [batch][2][8]
int pgs_debug_prof2(pgs_handle* h, unsigned long long* out) {
    TRY(check(h));
    if (!h->p.prof) return fail(SLAM_ERR_STATE, "set SLAM_PGS_PROF before pgs_create");
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out, h->p.prof + (size_t)h->B * h->lanes * 8, sizeof(unsigned long long) * 16 * (size_t)h->B, hipMemcpyDeviceToHost));
    return SLAM_OK;
}
int pgs_sync(pgs_handle* h) { TRY(check(h)); HIP_TRY(hipStreamSynchronize(h->stream)); return SLAM_OK; }
int pgs_timestep(const pgs_handle* h) { return h ? h->timestep : -1; }

}  // extern "C"
