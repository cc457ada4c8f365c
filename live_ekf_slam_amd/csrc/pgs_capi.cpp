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
    int p_notrim = 0;
    int chol_ll = 2;                           // SLAM_PGS_CHOL_LL=0: the right-looking Cholesky of rounds 1-3; 1: the left-looking kernel on 1024 threads; 2 (default): on 768
    int chol_threads = 0, chol_switch = 256;   // SLAM_PGS_CHOL_THREADS = 256 | 1024 forces; else 256 while > chol_switch instances are active
    bool trace = false;                       // SLAM_PGS_TRACE: print the active-instance count after every trial
    double path_ms[3] = {0.0, 0.0, 0.0};      // profiled solve: ms in the separate SYRK launches / in the fused chain + SYRK launches / in the segmented path's SYRK launches
    int seg_len = 32;                         // SLAM_PGS_SEG: poses per segment of the segmented elimination (pgs_seg_impl.h), 0 = the sequential chain of rounds 1-4
    bool seg_ok = false;                      // this solve runs the segmented elimination (decided in pgs_solve from the plan)
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
    if (const char* e = getenv("SLAM_PGS_FUSED")) h->fused_mode = atoi(e);
    if (const char* e = getenv("SLAM_PGS_SEG")) { const int v = atoi(e); h->seg_len = v <= 0 ? 0 : (v < 2 ? 2 : (v > slam::kPgsSegMaxLen ? slam::kPgsSegMaxLen : v)); }
    if (const char* e = getenv("SLAM_PGS_LIST")) h->use_list = atoi(e) != 0;
    if (const char* e = getenv("SLAM_PGS_NOTRIM")) h->p_notrim = atoi(e) ? atoi(e) : 1;
    if (const char* e = getenv("SLAM_PGS_GROUPS")) h->groups = atoi(e);
    if (const char* e = getenv("SLAM_PGS_CHOL_THREADS")) h->chol_threads = atoi(e) == 256 ? 256 : (atoi(e) == 1024 ? 1024 : 0);
    if (const char* e = getenv("SLAM_PGS_CHOL_SWITCH")) h->chol_switch = atoi(e);
    if (const char* e = getenv("SLAM_PGS_CHOL_LL")) h->chol_ll = atoi(e);
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
        const double per_slot = 8.0 * (((double)round_up(3 * N_max, 4) + 10.0 * nsegx) * h->LD + (double)h->LD * h->LD + (double)K * 29 + (double)N * 51 + (double)L * 12 + nsegx * (48 + 128.0 * 128.0)) +
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
        A(&p.Gs, S * N * 9); A(&p.segout, S * G * 32); A(&p.sepfac, S * G * 16); A(&p.segT, S * G * 128 * 128);
    }
    A(&p.Y, S * (size_t)p.y_stride); A(&p.S, S * (size_t)h->LD * h->LD);
    A(&p.dl, S * L * 2); A(&p.dp, S * N * 3);
    A(&p.lambda, S); A(&p.error, B); A(&p.cur_error, B); A(&p.err_init, B);
    A(&p.iters, B); A(&p.trials, B); A(&p.state, S); AC(&p.solve_ok, 1); A(&p.n_active, 64); A(&p.alist, S); A(&p.inst_flop, B); A(&p.work, 3);
    A(&p.nl, B); A(&p.nlin, S); A(&p.nerr, S); A(&p.nok, S);
    A(&h->dcount, B); A(&h->dsec, B * 3); A(&h->dout, B);
    if (getenv("SLAM_PGS_PROF")) { A(&p.prof, S * 24); }   // [S][8] chol phase timers, then [S][2][8] per-workgroup stamps of the fused chain
    if (rc != SLAM_OK) { pgs_destroy(h); return rc; }
    hipMemsetAsync(p.truth_hist, 0, sizeof(double) * B * N * 2, h->stream);
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
int clone_instances(pgs_handle* h, const slam::PgsParams& p, hipStream_t stream) {
    const size_t B = (size_t)h->B, off = (size_t)p.b_off, cnt = (size_t)p.b_cnt;
    for (int j = 1; j < h->lanes; ++j) {
        for (const pgs_handle::Slab& sl : h->clone_slabs)
            HIP_TRY(hipMemcpyAsync((char*)sl.ptr + ((size_t)j * B + off) * sl.bytes, (const char*)sl.ptr + off * sl.bytes, cnt * sl.bytes,
                                   hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(p.state + (size_t)j * B + off), 1, cnt, stream));
    }
    return SLAM_OK;
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
                 bool profile, bool prelaunched = false) {
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
    p.lanes_next = active_hint <= h->lanes_switch_all ? h->lanes : (active_hint <= h->lanes_switch ? (h->lanes < 2 ? h->lanes : 2) : 1);
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
        HIP_TRY(slam::pgs_launch_seg_plan(h->p, h->stream));   // (also counts the factors: the grid of the per-factor kernels)
        std::vector<int32_t> U((size_t)h->B), F((size_t)h->B);
        if (h->seg_len > 0) HIP_TRY(hipMemcpyAsync(U.data(), h->p.seg_umax, sizeof(int32_t) * (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(F.data(), h->p.fact_cnt, sizeof(int32_t) * (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        int mx = 0, fx = 0;
        for (int32_t f : F) fx = f > fx ? f : fx;
        h->p.nfact_max = fx;
        if (h->seg_len > 0) {
            for (int32_t u : U) mx = u > mx ? u : mx;
            h->seg_ok = mx <= slam::kPgsSegMaxLm;
            h->p.seg_on = h->seg_ok ? 1 : 0;
        }
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
    if (G > 16) G = 16;
    if (G > h->B) G = h->B;
    if (h->profiling) G = 1;   // per-kernel timing wants the kernels of one stream back to back
    if (G <= 1) {
        HIP_TRY(slam::pgs_launch_lm_begin(h->p, h->stream));
        TRY(clone_instances(h, h->p, h->stream));
        int trials = 0;
        int32_t act[3] = {h->B, 1, h->B};   // active instances, lanes and active slots of the next trial
        if (!h->h_active) HIP_TRY(hipHostMalloc((void**)&h->h_active, sizeof(int32_t) * 64, hipHostMallocDefault));
        while ((int)h->gevents.size() < 1) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); h->gevents.push_back(e); }
        const bool pipe = !h->profiling;   // per-kernel timing wants every kernel of a trial between its own events
        if (pipe) TRY(prelaunch_trial(h, h->p, h->stream));
        for (; trials < h->max_trials; ++trials) {
            TRY(launch_trial(h, h->p, act[0], act[1], act[2], h->stream, trials, h->profiling, pipe));
            HIP_TRY(hipMemcpyAsync(h->h_active, h->p.n_active, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipEventRecord(h->gevents[0], h->stream));
            if (pipe) TRY(prelaunch_trial(h, h->p, h->stream));   // the next trial's linearisation runs while the host waits below
            HIP_TRY(hipEventSynchronize(h->gevents[0]));
            act[0] = h->h_active[0]; act[1] = h->h_active[1]; act[2] = h->h_active[2];
            if (h->trace) {
                static thread_local double t_prev = 0.0;
                timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
                const double now = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
                fprintf(stderr, "pgs trial %d: %.2f ms, lanes %d -> active %d, lanes next %d\n", trials, trials ? now - t_prev : 0.0, (int)h->p.lanes, (int)act[0], (int)act[1]);
                t_prev = now;
            }
            if (act[0] == 0) { trials += 1; break; }
        }
        h->last_trials = trials;
        HIP_TRY(slam::pgs_launch_lm_end(h->p, h->stream));
        if (h->profiling) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            for (int k = 0; k < slam::kPgsTrialKernels; ++k) h->kernel_ms[k] = 0.0;
            for (int t = 0; t < trials; ++t)
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
    // ---- G groups, each with its own stream and LM loop; the host serves them round-robin ----
    // The groups' streams must not share a hardware queue: two LM loops whose launches sit in ONE in-order queue wait for each other's
    // queued trials at every host synchronisation (measured inside bench.py's driver command, where the earlier legs' streams shift the
    // runtime's stream -> queue assignment: 3.40 k solves/s with two groups against 5.21 k with one and 5.66 k with two on distinct
    // queues; profiles/r05_pgs/hw_queues.txt).  The runtime keeps separate queues per stream PRIORITY, so the groups alternate between
    // the priority levels the device offers (SLAM_PGS_GROUP_PRIO=0: all at the default priority, the behaviour before).
    while ((int)h->gstreams.size() < G) {
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
    HIP_TRY(hipEventRecord(h->gevents[G], h->stream));   // everything queued on the handle's stream so far comes first
    std::vector<slam::PgsParams> gp(G, h->p);
    std::vector<int> gtrials(G, 0);
    std::vector<char> gdone(G, 0);
    const int per = (h->B + G - 1) / G;
    for (int g = 0; g < G; ++g) {
        gp[g].b_off = g * per;
        gp[g].b_cnt = (h->B - g * per) < per ? (h->B - g * per) : per;
        gp[g].n_active = h->p.n_active + 4 * g;
        gp[g].alist = h->p.alist + (size_t)g * per * h->lanes;
        if (gp[g].b_cnt <= 0) { gdone[g] = 1; continue; }
        HIP_TRY(hipStreamWaitEvent(h->gstreams[g], h->gevents[G], 0));
        HIP_TRY(slam::pgs_launch_lm_begin(gp[g], h->gstreams[g]));
        TRY(clone_instances(h, gp[g], h->gstreams[g]));
        TRY(launch_trial(h, gp[g], h->B, 1, gp[g].b_cnt, h->gstreams[g], 0, false));
        HIP_TRY(hipMemcpyAsync(h->h_active + 4 * g, gp[g].n_active, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, h->gstreams[g]));
        HIP_TRY(hipEventRecord(h->gevents[g], h->gstreams[g]));
        TRY(prelaunch_trial(h, gp[g], h->gstreams[g]));
    }
    // one host thread per group drives its LM loop (launch a trial, wait for its active count, decide); the HIP runtime
    // is thread-safe and the groups touch disjoint instance ranges
    std::vector<int> grc(G, SLAM_OK);
    auto drive = [&](int g) -> int {
        HIP_TRY(hipSetDevice(h->device));
        for (;;) {
            HIP_TRY(hipEventSynchronize(h->gevents[g]));
            const int32_t active = h->h_active[4 * g], lanes_next = h->h_active[4 * g + 1], nslots = h->h_active[4 * g + 2];
            gtrials[g] += 1;
            if (h->trace) fprintf(stderr, "pgs group %d trial %d: active %d\n", g, gtrials[g] - 1, (int)active);
            if (active == 0 || gtrials[g] >= h->max_trials) {
                HIP_TRY(slam::pgs_launch_lm_end(gp[g], h->gstreams[g]));
                HIP_TRY(hipEventRecord(h->gevents[g], h->gstreams[g]));
                return SLAM_OK;
            }
            TRY(launch_trial(h, gp[g], active * G, lanes_next, nslots, h->gstreams[g], gtrials[g], false, true));
            HIP_TRY(hipMemcpyAsync(h->h_active + 4 * g, gp[g].n_active, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, h->gstreams[g]));
            HIP_TRY(hipEventRecord(h->gevents[g], h->gstreams[g]));
            TRY(prelaunch_trial(h, gp[g], h->gstreams[g]));
        }
    };
    std::vector<std::thread> workers;
    for (int g = 1; g < G; ++g)
        if (!gdone[g]) workers.emplace_back([&, g]() { grc[g] = drive(g); });
    if (!gdone[0]) grc[0] = drive(0);
    for (auto& w : workers) w.join();
    int max_trials = 0;
    for (int g = 0; g < G; ++g) {
        if (grc[g] != SLAM_OK) return grc[g];
        if (gp[g].b_cnt > 0) HIP_TRY(hipStreamWaitEvent(h->stream, h->gevents[g], 0));   // the handle's stream continues after every group
        max_trials = gtrials[g] > max_trials ? gtrials[g] : max_trials;
    }
    h->last_trials = max_trials;
    return SLAM_OK;
}

// number of solve groups (0 = automatic: 2 from 512 instances)
int pgs_set_groups(pgs_handle* h, int groups) { TRY(check(h)); h->groups = groups < 0 ? 0 : groups; return SLAM_OK; }

int pgs_adopt_result(pgs_handle* h) {
    TRY(check(h));
    HIP_TRY(slam::pgs_launch_adopt(h->p, h->stream));
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
int pgs_last_solve_paths(pgs_handle* h, double out[8]) {
    TRY(check(h));
    if (!out) return fail(SLAM_ERR_ARG, "NULL output");
    HIP_TRY(hipStreamSynchronize(h->stream));
    double w[3];
    HIP_TRY(hipMemcpy(w, h->p.work, 3 * sizeof(double), hipMemcpyDeviceToHost));
    out[0] = w[0]; out[1] = w[1]; out[2] = h->path_ms[0]; out[3] = h->path_ms[1];
    out[4] = w[2]; out[5] = h->path_ms[2]; out[6] = h->seg_ok ? 1.0 : 0.0; out[7] = (double)h->seg_len;
    return SLAM_OK;
}
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
