// slam_capi.cpp — C ABI (include/slam_batch.h) over the HIP kernels.  Host-side only: owns device memory,
// the stream, the step counter and the config; every compute entry point enqueues one fused kernel.
// There is NO CPU fallback: without a HIP device every compute call fails with SLAM_ERR_HIP.
#include "../../include/slam_batch.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "capi_internal.h"
#include "host/config_parse.h"
#include "ekf_kernel.h"
#include "slam_math.h"
#include "ukf_kernel.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// (The runtime's "last error" is sticky and per thread: a launcher that ends in hipGetLastError() would report an error some OTHER library
// of the process left behind - PyTorch creating a stream right before slam_init did exactly that in a test.  It is cleared before every
// call; our own calls are all checked through their return values.)
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        (void)hipGetLastError();                                                                   \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(SLAM_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// fn(begin, end) over [0, n) on a few host threads (packing 65 536 per-instance messages is ~1 ms on one core)
template <class F>
void host_parallel(size_t n, F fn) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = n < 8192 ? 1 : (hw >= 8 ? 4 : (hw >= 2 ? 2 : 1));
    if (nt == 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (size_t i = 1; i < nt; ++i) th.emplace_back(fn, i * per, (i + 1) * per < n ? (i + 1) * per : n);
    fn((size_t)0, per < n ? per : n);
    for (auto& t : th) t.join();
}

}  // namespace

struct slam_handle {
    slam_config cfg;
    int kind, B, L_max, dtype, device;
    int n_max, ld_max, pstride, xstride;
    int waves_per_filter = 0;
    int dbg = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool inited = false;
    uint64_t seed = 2025;
    int64_t inst0 = 0;
    uint32_t step = 0;
    double range_max, fov_min, fov_max;
    // device buffers
    void* dP = nullptr; void* dP2 = nullptr;   // dP = current P_t; dP2 = second buffer (EKF: layout changes, UKF: ping-pong)
    void* dx = nullptr; int esz = 8;           // x_t; element size of P / x storage
    double* dscratch = nullptr;                // fp32 storage: fp64 slab for P between detection groups
    int32_t* dM = nullptr; int32_t* dids = nullptr;
    int32_t* dflags = nullptr; int32_t* dts = nullptr; double* dtruth = nullptr; double* derr = nullptr;
    double* dmap = nullptr; int L = 0;
    float* dmeas = nullptr; int32_t* dcount = nullptr; int k_stride = 0;   // last-measurement dump (slam_get_last_meas)
    // slam_step (host measurements): two pinned staging buffers + two device buffers, filled on a copy stream while the
    // previous step's kernel runs; no stream synchronisation per step
    struct Stage {
        float* hmeas = nullptr; int32_t* hcount = nullptr;     // pinned host
        float* dmeas = nullptr; int32_t* dcount = nullptr;     // device
        size_t cap = 0;                                        // floats in hmeas / dmeas
        hipEvent_t copied = nullptr, used = nullptr;
        bool in_use = false;
    } stage[2];
    hipStream_t copy_stream = nullptr;
    hipEvent_t shadow_ev = nullptr;
    uint32_t stage_next = 0;
    unsigned long long* dkhist = nullptr;                      // [8] instance-steps by detection count
    // slam_step_sim (EKF): the commands of consecutive calls are queued on the host and run as ONE multi-step launch when the
    // queue is full or anything else touches the handle (every other entry point flushes first).  The kernels are asynchronous
    // anyway, and a multi-step launch gives the same bits as single steps, so only the speed changes (one launch per call
    // re-reads x, ids and the thin rows / columns of P and cannot keep update groups open across timesteps).
    std::vector<float> lazy_cmds;
    int eager_init = 2;                                        // first idle-GPU launch size (SLAM_EAGER_FLUSH, 0 = off)
    int eager_target = 2;                                      // queued steps an idle GPU is given at once (doubles per such launch)
    bool lazy_explicit = false;                                // slam_set_lazy_steps / SLAM_LAZY_STEPS asked for queueing: slam_step_dev queues only then
    int lazy_max = 32;                                         // 0 / 1 = off (SLAM_LAZY_STEPS, slam_set_lazy_steps); per launch a workgroup pays
                                                               // ~25 us of start / drain: 16 -> 61 M, 32 -> 66 M, 64 -> 70 M steps/s (one launch: 73 M)
    // slam_step (EKF, HOST measurements): the same queueing.  Each call packs its message (stride kExtQ detections per
    // instance; a message with more goes through the immediate path) and its command into one of two pinned queues of up to
    // lazy_max timesteps; a flush copies the queue on the copy stream and runs ONE multi-step launch that takes the message of
    // timestep t from the device-side queue instead of the generator.
    struct ExtQueue {
        float* hmeas = nullptr; int32_t* hcount = nullptr; float* hcmds = nullptr;   // pinned: [cap][B][kExtQ][3], [cap][B], [cap][2]
        float* dmeas = nullptr; int32_t* dcount = nullptr;                           // device
        int cap = 0, n = 0;
        int ks = 0;                                                                  // detections per instance of this fill (<= kExtQ)
        size_t meas_cap = 0;                                                         // floats in hmeas / dmeas
        hipEvent_t copied = nullptr, used = nullptr;
        bool in_use = false;
    } extq[2];
    int extq_cur = 0;
    int extq_ks = 2;                                                                 // stride of the next fill: largest message seen so far
    // slam_step_dev (EKF, DEVICE measurements): the same queueing.  The message is copied device-to-device into the queue on
    // the compute stream at the call (so the caller may overwrite its buffers in stream order, as with an immediate launch)
    // and up to lazy_max of them run as one multi-step launch.  One buffer suffices: copies and launches share the stream.
    struct DevQueue {
        float* dmeas = nullptr; int32_t* dcount = nullptr;   // [cap][B][ks][3], [cap][B]
        std::vector<float> cmds;
        int cap = 0, ks = 0, n = 0;
    } devq;
    // slam_track_instance: instance `tracked` also runs in a one-instance SHADOW filter (same config, seed, map and GLOBAL
    // instance id, hence the same bits: results do not depend on how a batch is partitioned), stepped at once at every step
    // call on its own stream, so that slam_get_state(h, tracked) - the publishState of every tick, localization_node.cpp:139 -
    // does not have to run the batch's queued timesteps first.
    slam_handle* shadow = nullptr; int tracked = -1;
    std::vector<double> hmap;                  // host copy of the map (the shadow needs it)
    double* dscalar = nullptr;
    unsigned long long* dprof = nullptr;
    double* dsq = nullptr; int32_t* dnsq = nullptr;   // UKF: matrix square root scratch + its dimension
    double* dxprev = nullptr;                         // UKF: x_t the last sigma points were drawn around
    double* dvt = nullptr; int32_t* dvage = nullptr;  // UKF: V^T of the last eigen-decomposition + warm-start age
    double* dbigws = nullptr;                         // UKF beyond the LDS size classes: [B][2 * pstride] scratch (ukf_big_kernel.hip)
    uint4* drot = nullptr;                            // UKF (n <= 44): pass table of the fast sqrt kernel (launch_ukf_quad_table)
    hipStream_t aux_stream[3] = {nullptr, nullptr, nullptr}; hipEvent_t aux_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // UKF run_sim: the other parts of the batch
    int ukf_parts = 2;                                                               // streams the UKF batch is split over (SLAM_UKF_PARTS, 1..4)
    int ukf_split_min = 1024;                                                        // batch size from which it is used
    bool predicted = false; float pred_cmd[2] = {0.f, 0.f};   // UKF: slam_predict done, slam_update_dev pending
    float* dmapf = nullptr;                           // UKF_LOC: the known map as float32 [id, x, y] triplets
    float* dcmds = nullptr; int cmds_cap = 0;         // command sequence of a multi-step launch (slam_run_sim)
    int run_chunk = 0;                                // timesteps per launch in slam_run_sim (0 = all of them)
    int base = 3;                                     // state offset of the first landmark: 3 (EKF) or 4 (UKF)
    bool dump_meas = false;
};

namespace {

void fill_params(slam_handle* h, slam::EkfStepParams& p, const float cmd[2]) {
    memset(&p, 0, sizeof(p));
    p.P = h->dP; p.P_out = h->dP2; p.x = h->dx; p.scratch = h->dscratch; p.M = h->dM; p.ids = h->dids; p.flags = h->dflags; p.timestep = h->dts;
    p.truth = h->dtruth; p.err_sum = h->derr; p.map = h->dmap; p.L = h->L;
    p.fwd = cmd[0]; p.ang = cmd[1];
    const slam_config& c = h->cfg;
    p.v_d = c.v_d; p.v_th = c.v_th; p.w_r = c.w_r; p.w_b = c.w_b;
    if (c.replicate_vw_quirk) {  // filter.h:116-117: W_00/W_11 land in V, W stays I2
        p.V00 = c.W_00; p.V11 = c.W_11; p.W00 = 1.0; p.W11 = 1.0;
    } else {
        p.V00 = c.V_00; p.V11 = c.V_11; p.W00 = c.W_00; p.W11 = c.W_11;
    }
    p.id_known = c.landmark_id_is_known;
    p.min_sep = c.min_landmark_separation;
    p.abs_is_int = c.ekf_abs_is_int ? 1 : 0; p.lm_from_pred = c.ekf_landmark_from_x_pred ? 1 : 0;
    p.sV00 = c.V_00; p.sV11 = c.V_11; p.sW00 = c.W_00; p.sW11 = c.W_11;
    p.d_max = c.d_max; p.th_max = c.th_max;
    p.range_max = h->range_max; p.fov_min = h->fov_min; p.fov_max = h->fov_max;
    p.seed = h->seed; p.inst0 = h->inst0; p.step = h->step;
    p.B = h->B; p.L_max = h->L_max; p.pstride = h->pstride; p.xstride = h->xstride;
    p.dbg = h->dbg;
    p.prof = (h->dbg & (4 | 32)) ? h->dprof : nullptr;
    p.khist = h->dkhist;
}

void fill_ukf_params(slam_handle* h, slam::UkfStepParams& p, const float cmd[2]) {
    memset(&p, 0, sizeof(p));
    p.P = (const double*)h->dP; p.P_out = (double*)h->dP2; p.x = (double*)h->dx; p.sqtP = h->dsq; p.n_sq = h->dnsq; p.x_prev = h->dxprev; p.Vt_store = h->dvt; p.v_age = h->dvage;
    p.M = h->dM; p.ids = h->dids; p.flags = h->dflags; p.timestep = h->dts;
    p.truth = h->dtruth; p.err_sum = h->derr; p.map = h->dmap; p.L = h->L;
    p.fwd = cmd[0]; p.ang = cmd[1];
    const slam_config& c = h->cfg;
    p.v_d = c.v_d; p.v_th = c.v_th; p.w_r = c.w_r; p.w_b = c.w_b;
    if (c.replicate_vw_quirk) { p.V00 = c.W_00; p.V11 = c.W_11; p.W00 = 1.0; p.W11 = 1.0; }
    else { p.V00 = c.V_00; p.V11 = c.V_11; p.W00 = c.W_00; p.W11 = c.W_11; }
    p.float_trig = c.ukf_float_trig;
    p.acc_zest1 = c.ukf_accumulate_zest1 ? 1 : 0; p.yaw_sigma = c.ukf_sensing_yaw_from_sigma ? 1 : 0;
    p.sV00 = c.V_00; p.sV11 = c.V_11; p.sW00 = c.W_00; p.sW11 = c.W_11;
    p.d_max = c.d_max; p.th_max = c.th_max;
    p.range_max = h->range_max; p.fov_min = h->fov_min; p.fov_max = h->fov_max;
    p.seed = h->seed; p.inst0 = h->inst0; p.step = h->step;
    p.B = h->B; p.L_max = h->L_max; p.pstride = h->pstride; p.xstride = h->xstride;
    p.b_off = 0; p.b_cnt = h->B;
    p.loc = h->kind == SLAM_UKF_LOC; p.mapf = h->dmapf;
    p.quad_tab = h->drot;
    p.khist = h->dkhist;
    p.big_ws = h->dbigws;
    p.prof = (h->dbg & 4) ? h->dprof : nullptr;
}

// ekf.cpp:65,73 and ukf.cpp:249-287 walk a message of any length.  The LDS size classes hold as many detections of one message as the class
// holds landmarks (every message without repeated ids about a map the state can hold fits); a launch whose messages MAY be longer - external
// measurements: the caller's bound k_stride is (slam_step passes the largest count it saw); SIM mode: the map is - runs the long-message path
// of launch_ekf_step / launch_ukf_step: the instances concerned go through the HBM-streamed kernels, which read a message where it lies
// (same state layout, same arithmetic: both are bit-identical to the oracle; about a pass over P per detection slower).  Returns the class
// capacity if that path is needed, 0 if not (the streamed classes have no limit in the first place).  fp32-storage EKF handles too: the
// streamed kernel reads and writes floats there and runs the timestep in the handle's fp64 slab.
int long_message_cap(slam_handle* h, int sim, int k_stride, int* cap_out) {
    *cap_out = 0;
    int cap;
    if (h->kind == SLAM_EKF_SLAM) {
        if (h->L_max > (h->esz == 4 ? slam::kEkfLdsMaxLandmarksF32 : slam::kEkfLdsMaxLandmarks)) return SLAM_OK;   // a streamed class: no limit
        cap = slam::ekf_class_message_capacity(h->L_max);
    } else {
        if (h->kind == SLAM_UKF_SLAM && h->L_max > slam::kUkfLdsMaxLandmarks) return SLAM_OK;
        cap = slam::ukf_class_message_capacity(h->L_max, h->kind == SLAM_UKF_LOC, h->L);
    }
    if (!(sim ? h->L > cap : k_stride > cap)) return SLAM_OK;
    if (h->kind != SLAM_EKF_SLAM && !h->dbigws)   // scratch of the streamed step kernel (P_pred), on first use
        HIP_TRY(hipMalloc(&h->dbigws, sizeof(double) * (size_t)h->B * 2 * h->pstride));
    *cap_out = cap;
    return SLAM_OK;
}

// one timestep, either filter kind; `sim` = device-side measurement generator
int launch_step(slam_handle* h, const float cmd[2], int sim, const float* d_meas, const int32_t* d_count, int k_stride) {
    if (h->predicted) return fail(SLAM_ERR_STATE, "a prediction stage is pending: call slam_update_dev before the next step");
    int long_cap = 0;
    if (const int rc = long_message_cap(h, sim, k_stride, &long_cap)) return rc;
    if (h->kind == SLAM_EKF_SLAM) {
        slam::EkfStepParams p;
        fill_params(h, p, cmd);
        p.sim = sim;
        p.long_mode = long_cap > 0 ? 1 : 0; p.long_cap = long_cap;
        p.meas_in = d_meas; p.meas_count_in = d_count; p.k_stride_in = k_stride;
        if (sim && h->dump_meas) { p.meas_out = h->dmeas; p.meas_count_out = h->dcount; p.k_stride_out = h->k_stride; }
        HIP_TRY(slam::launch_ekf_step(p, h->waves_per_filter, h->esz == 4, h->stream));
    } else {
        slam::UkfStepParams p;
        fill_ukf_params(h, p, cmd);
        p.sim = sim;
        p.long_mode = long_cap > 0 ? 1 : 0; p.long_cap = long_cap;
        p.meas_in = d_meas; p.meas_count_in = d_count; p.k_stride_in = k_stride;
        if (sim && h->dump_meas) { p.meas_out = h->dmeas; p.meas_count_out = h->dcount; p.k_stride_out = h->k_stride; }
        HIP_TRY(slam::launch_ukf_sqrt(p, h->stream));   // nearestSPD + sqrt (ukf.cpp:106-123,208)
        HIP_TRY(slam::launch_ukf_step(p, h->stream));   // predictionStage + updateStage (ukf.cpp:197-372)
    }
    if (h->kind != SLAM_EKF_SLAM) std::swap(h->dP, h->dP2);   // UKF: the kernel wrote the next P_t into the other buffer
    h->step += 1;                                              // (the EKF kernel updates dP in place)
    return SLAM_OK;
}

int ensure_meas_buffers(slam_handle* h, int k_stride) {
    if (h->dmeas && h->k_stride >= k_stride) return SLAM_OK;
    if (h->dmeas) { hipFree(h->dmeas); hipFree(h->dcount); h->dmeas = nullptr; h->dcount = nullptr; }
    HIP_TRY(hipMalloc(&h->dmeas, sizeof(float) * 3 * (size_t)k_stride * h->B));
    HIP_TRY(hipMalloc(&h->dcount, sizeof(int32_t) * (size_t)h->B));
    HIP_TRY(hipMemsetAsync(h->dcount, 0, sizeof(int32_t) * (size_t)h->B, h->stream));
    h->k_stride = k_stride;
    return SLAM_OK;
}

}  // namespace

static int run_sim_now(slam_handle* h, const float* cmds, int T);
static int flush_lazy(slam_handle* h);
static int flush_ext(slam_handle* h);
static int flush_dev(slam_handle* h);
// The queues exist to give the GPU long multi-step launches while the caller keeps calling once per tick; when the GPU has
// nothing to do, waiting for the queue to fill only delays the work (a run of K calls paid one whole queue of packing with
// the GPU idle before the first launch).  So a queued step is launched at once if the compute stream is idle.
// Launching every single queued step would waste the multi-step kernel (2.4 vs 1.5 ms per step), so the idle-GPU launch
// needs `eager_target` queued steps, and the target doubles with every such launch (2, 4, 8, ... up to the queue length):
// the pipeline fills geometrically instead of after one whole queue.  SLAM_EAGER_FLUSH=0 switches it off.  Only slam_step
// does this (its caller spends ~1 ms per call delivering a message); slam_step_sim calls cost microseconds, so its queue is
// full long before the first launch would have finished.
static bool eager_flush(slam_handle* h, int queued) {
    if (h->eager_target <= 0 || queued < h->eager_target) return false;
    const hipError_t e = hipStreamQuery(h->stream);
    (void)hipGetLastError();   // hipErrorNotReady is an answer, not an error
    if (e != hipSuccess) return false;
    h->eager_target = 2 * queued < h->lazy_max ? 2 * queued : h->lazy_max;
    return true;
}
static constexpr int kExtQ = 4;   // detections per instance a queued slam_step message can hold
#define FLUSH(h)                        \
    do {                                \
        const int frc_ = flush_lazy(h); \
        if (frc_) return frc_;          \
    } while (0)

extern "C" {

int slam_internal_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

const char* slam_last_error(void) { return g_err; }
const char* slam_version(void) { return "live_ekf_slam_amd 0.1 (gfx950)"; }

int slam_config_default(slam_config* c) {
    if (!c) return fail(SLAM_ERR_ARG, "cfg is NULL");
    memset(c, 0, sizeof(*c));
    c->v_d = 0.0f; c->v_th = 0.0f; c->V_00 = 0.01; c->V_11 = 0.001;      // params.yaml:39-45
    c->w_r = 0.0f; c->w_b = 0.0f; c->W_00 = 0.01; c->W_11 = 0.01;        // params.yaml:46-52
    c->landmark_id_is_known = 1; c->min_landmark_separation = 0.1f;        // params.yaml:35-36
    c->d_max = 0.1; c->th_max = 0.0546;                                    // params.yaml:27-28
    c->range_max = 3.0; c->fov_min = -1.57; c->fov_max = 1.57;             // params.yaml:30-32
    c->init_x = 0.0; c->init_y = 0.0; c->init_yaw = 0.0;                   // params.yaml:19-22
    c->replicate_vw_quirk = 1;
    c->ukf_float_trig = 1;
    return SLAM_OK;
}

// The reference reads one YAML file with nested maps; the keys the hot path needs are unique leaf names
// except min_landmark_separation (constraints.measurements vs map), disambiguated by section tracking.  The reader
// itself is host/config_parse.h (host-only, also compiled under ASan + UBSan by `make -C oracle asan`).
int slam_config_load(slam_config* c, const char* path) {
    if (!c || !path) return fail(SLAM_ERR_ARG, "NULL argument");
    std::string err;
    const int rc = slam_host::config_parse_file(c, path, &err);
    if (rc) return fail(SLAM_ERR_IO, "%s", err.c_str());
    return SLAM_OK;
}

int slam_create(const slam_config* cfg, int kind, int batch, int L_max, int dtype, int device, slam_handle** out) {
    if (!cfg || !out) return fail(SLAM_ERR_ARG, "NULL argument");
    if (batch <= 0 || L_max <= 0) return fail(SLAM_ERR_ARG, "batch and L_max must be positive");
    if (kind != SLAM_EKF_SLAM && kind != SLAM_UKF_SLAM && kind != SLAM_UKF_LOC)
        return fail(SLAM_ERR_ARG, "unknown filter kind %d", kind);
    // The four quirk switches took over fields that were `reserved` until round 4: a caller compiled against that header may
    // have left anything there, and any non-zero value would silently switch a reference quirk off.  Only 0 and 1 are configurations.
    for (const int q : {cfg->ekf_abs_is_int, cfg->ekf_landmark_from_x_pred, cfg->ukf_accumulate_zest1, cfg->ukf_sensing_yaw_from_sigma})
        if (q != 0 && q != 1) return fail(SLAM_ERR_ARG, "slam_config quirk switch = %d: the switches (ekf_abs_is_int, ekf_landmark_from_x_pred, ukf_accumulate_zest1, "
                                          "ukf_sensing_yaw_from_sigma) are 0 or 1 - zero-initialise the struct (slam_config_default)", q);
    if (kind == SLAM_UKF_LOC) L_max = 1;   // localisation only: the state never holds landmarks
    if (dtype != SLAM_F64 && !(dtype == SLAM_F32 && kind == SLAM_EKF_SLAM))
        return fail(SLAM_ERR_UNSUPPORTED, "fp32 state storage is implemented for EKF_SLAM only");
    if (L_max > (kind != SLAM_EKF_SLAM ? slam::kUkfMaxLandmarks : slam::kEkfMaxLandmarks))
        return fail(SLAM_ERR_UNSUPPORTED, "L_max %d exceeds the limit of this filter kind (EKF %d, UKF %d): the EKF beyond %d landmarks (fp32 storage: beyond %d) and the UKF beyond 50 run the HBM-streamed size classes, whose working set is 2 x n x n doubles per instance in HBM", L_max, slam::kEkfMaxLandmarks, slam::kUkfMaxLandmarks, slam::kEkfLdsMaxLandmarks, slam::kEkfLdsMaxLandmarksF32);
    HIP_TRY(hipSetDevice(device));
    slam_handle* h = new slam_handle();
    h->cfg = *cfg; h->kind = kind; h->B = batch; h->L_max = L_max; h->dtype = dtype; h->device = device;
    h->esz = dtype == SLAM_F32 ? 4 : 8;
    h->base = (kind == SLAM_EKF_SLAM) ? 3 : 4;
    h->n_max = h->base + 2 * L_max;
    // per-filter slab: n_max rows of the padded leading dimension (EKF: slam::ekf_ld, rows start on 16-byte boundaries;
    // the UKF packs n x n with n even), 256/512-byte aligned
    h->ld_max = kind == SLAM_EKF_SLAM ? slam::ekf_ld(h->n_max, h->esz) : h->n_max;
    h->pstride = round_up(h->n_max * h->ld_max + 4, 64);
    h->xstride = round_up(h->n_max + 1, 2);
    h->range_max = cfg->range_max; h->fov_min = cfg->fov_min; h->fov_max = cfg->fov_max;
    const char* env = getenv("SLAM_WAVES_PER_FILTER");
    if (env) h->waves_per_filter = atoi(env);
    env = getenv("SLAM_DEBUG_FLAGS");   // 4 / 32: phase and per-step timers.  The ablation bits (1, 2, 16: WRONG results) are
    if (env) h->dbg = atoi(env);        // compiled out of the release kernels (-DSLAM_ABLATE builds only)
#ifndef SLAM_ABLATE
    h->dbg &= (4 | 32);   // (the fault-injection bit 128 is reachable through slam_set_debug_flags only: no environment variable can plant it)
#endif
    env = getenv("SLAM_UKF_SPLIT_MIN");   // batch size from which UKF run_sim splits the batch over two streams
    if (env) h->ukf_split_min = atoi(env);
    env = getenv("SLAM_UKF_PARTS");   // streams the UKF batch is split over in run_sim (2..4)
    if (env) h->ukf_parts = atoi(env);
    env = getenv("SLAM_LAZY_STEPS");   // slam_step_sim calls queued per multi-step launch (0 = one launch per call)
    if (env) { h->lazy_max = atoi(env); h->lazy_explicit = true; }
    env = getenv("SLAM_EAGER_FLUSH");   // queued steps from which an idle GPU is given work before the queue is full (0 = never)
    if (env) h->eager_init = h->eager_target = atoi(env);
    env = getenv("SLAM_RUN_CHUNK");   // timesteps per launch of slam_run_sim (1 = one launch per step)
    if (env) h->run_chunk = atoi(env);
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(SLAM_ERR_HIP, "hipStreamCreate -> %s", hipGetErrorString(e)); }
    h->own_stream = true;
    const size_t B = (size_t)batch;
    hipError_t errs[] = {
        hipMalloc(&h->dP, (size_t)h->esz * B * h->pstride),
        hipMalloc(&h->dP2, (size_t)h->esz * B * h->pstride),
        hipMalloc(&h->dx, (size_t)h->esz * B * h->xstride),
        hipMalloc(&h->dM, sizeof(int32_t) * B),
        hipMalloc(&h->dids, sizeof(int32_t) * B * L_max),
        hipMalloc(&h->dflags, sizeof(int32_t) * B),
        hipMalloc(&h->dts, sizeof(int32_t) * B),
        hipMalloc(&h->dtruth, sizeof(double) * B * 3),
        hipMalloc(&h->derr, sizeof(double) * B),
        hipMalloc(&h->dscalar, sizeof(double) * 4),
        (h->dbg & (4 | 32)) ? hipMalloc(&h->dprof, sizeof(unsigned long long) * slam::kEkfProfSlots * B) : hipSuccess,
        hipMalloc(&h->dkhist, sizeof(unsigned long long) * 16),
        kind != SLAM_EKF_SLAM ? hipMalloc(&h->dsq, sizeof(double) * B * h->pstride) : hipSuccess,
        h->esz == 4 ? hipMalloc(&h->dscratch, sizeof(double) * B * h->pstride) : hipSuccess,
        kind != SLAM_EKF_SLAM ? hipMalloc(&h->dnsq, sizeof(int32_t) * B) : hipSuccess,
        kind != SLAM_EKF_SLAM ? hipMalloc(&h->dxprev, sizeof(double) * B * h->xstride) : hipSuccess,
        kind != SLAM_EKF_SLAM ? hipMalloc(&h->dvt, sizeof(double) * B * h->pstride) : hipSuccess,
        (kind == SLAM_UKF_SLAM && L_max > slam::kUkfLdsMaxLandmarks) ? hipMalloc(&h->dbigws, sizeof(double) * B * 2 * h->pstride) : hipSuccess,
        kind != SLAM_EKF_SLAM ? hipMalloc(&h->dvage, sizeof(int32_t) * B) : hipSuccess,
        (kind != SLAM_EKF_SLAM && h->n_max <= 44) ? hipMalloc(&h->drot, sizeof(uint4) * slam::kUkfQuadTabEntries) : hipSuccess,
    };
    for (hipError_t ee : errs)
        if (ee != hipSuccess) {
            slam_destroy(h);
            return fail(SLAM_ERR_HIP, "hipMalloc -> %s", hipGetErrorString(ee));
        }
    // every buffer a getter can read before slam_init is zeroed (M = 0, flags = 0, timestep = 0, ...)
    hipError_t zs[] = {
        hipMemsetAsync(h->dP, 0, (size_t)h->esz * B * h->pstride, h->stream),
        hipMemsetAsync(h->dP2, 0, (size_t)h->esz * B * h->pstride, h->stream),
        hipMemsetAsync(h->dx, 0, (size_t)h->esz * B * h->xstride, h->stream),
        hipMemsetAsync(h->dids, 0, sizeof(int32_t) * B * L_max, h->stream),
        hipMemsetAsync(h->dM, 0, sizeof(int32_t) * B, h->stream),
        hipMemsetAsync(h->dflags, 0, sizeof(int32_t) * B, h->stream),
        hipMemsetAsync(h->dts, 0, sizeof(int32_t) * B, h->stream),
        hipMemsetAsync(h->dtruth, 0, sizeof(double) * B * 3, h->stream),
        hipMemsetAsync(h->derr, 0, sizeof(double) * B, h->stream),
        hipMemsetAsync(h->dkhist, 0, sizeof(unsigned long long) * 16, h->stream),
        h->dnsq ? hipMemsetAsync(h->dnsq, 0, sizeof(int32_t) * B, h->stream) : hipSuccess,
        h->dprof ? hipMemsetAsync(h->dprof, 0, sizeof(unsigned long long) * slam::kEkfProfSlots * B, h->stream) : hipSuccess,
        h->drot ? slam::launch_ukf_quad_table(h->drot, h->stream) : hipSuccess,
    };
    for (hipError_t ee : zs)
        if (ee != hipSuccess) {
            slam_destroy(h);
            return fail(SLAM_ERR_HIP, "hipMemsetAsync -> %s", hipGetErrorString(ee));
        }
    *out = h;
    return SLAM_OK;
}

int slam_destroy(slam_handle* h) {
    if (!h) return SLAM_OK;
    if (h->shadow) { slam_destroy(h->shadow); h->shadow = nullptr; }
    flush_lazy(h);
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (auto& st : h->aux_stream) if (st) { hipStreamSynchronize(st); hipStreamDestroy(st); }
    for (auto& ev : h->aux_ev) if (ev) hipEventDestroy(ev);
    if (h->copy_stream) { hipStreamSynchronize(h->copy_stream); hipStreamDestroy(h->copy_stream); }
    if (h->shadow_ev) hipEventDestroy(h->shadow_ev);
    if (h->devq.dmeas) { hipFree(h->devq.dmeas); hipFree(h->devq.dcount); }
    for (auto& q : h->extq) {
        if (q.hmeas) { hipHostFree(q.hmeas); hipHostFree(q.hcount); hipHostFree(q.hcmds); hipFree(q.dmeas); hipFree(q.dcount); }
        if (q.copied) { hipEventDestroy(q.copied); hipEventDestroy(q.used); }
    }
    for (auto& s : h->stage) {
        if (s.hmeas) hipHostFree(s.hmeas);
        if (s.hcount) hipHostFree(s.hcount);
        if (s.dmeas) hipFree(s.dmeas);
        if (s.dcount) hipFree(s.dcount);
        if (s.copied) hipEventDestroy(s.copied);
        if (s.used) hipEventDestroy(s.used);
    }
    void* bufs[] = {h->dP, h->dP2, h->dx, h->dM, h->dids, h->dflags, h->dts, h->dtruth, h->derr, h->dmap, h->dmeas, h->dcount, h->dscalar, h->dprof, h->dsq, h->dnsq, h->dscratch, h->dmapf, h->dcmds, h->dxprev, h->dvt, h->dvage, h->dkhist, h->drot, h->dbigws};
    for (void* q : bufs)
        if (q) hipFree(q);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return SLAM_OK;
}

int slam_set_stream(slam_handle* h, void* s) {
    if (!h) return fail(SLAM_ERR_ARG, "NULL handle");
    FLUSH(h);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)s;
    h->own_stream = false;
    return SLAM_OK;
}
int slam_set_instance_offset(slam_handle* h, int64_t v) {
    if (!h) return fail(SLAM_ERR_ARG, "NULL handle");
    FLUSH(h);
    h->inst0 = v;
    return h->shadow ? slam_set_instance_offset(h->shadow, v + h->tracked) : SLAM_OK;
}
int slam_set_seed(slam_handle* h, uint64_t s) {
    if (!h) return fail(SLAM_ERR_ARG, "NULL handle");
    FLUSH(h);
    h->seed = s;
    return h->shadow ? slam_set_seed(h->shadow, s) : SLAM_OK;
}
int slam_set_vision(slam_handle* h, double range_max, double fov_min, double fov_max) {
    if (!h) return fail(SLAM_ERR_ARG, "NULL handle");
    FLUSH(h);
    h->range_max = range_max; h->fov_min = fov_min; h->fov_max = fov_max;
    return h->shadow ? slam_set_vision(h->shadow, range_max, fov_min, fov_max) : SLAM_OK;
}

int slam_init(slam_handle* h, float x0, float y0, float yaw0) {
    if (!h) return fail(SLAM_ERR_ARG, "NULL handle");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    if (h->kind == SLAM_EKF_SLAM) {
        slam::EkfInitParams p;
        p.P = h->dP; p.x = h->dx; p.M = h->dM; p.flags = h->dflags; p.timestep = h->dts; p.truth = h->dtruth; p.err_sum = h->derr;
        p.B = h->B; p.pstride = h->pstride; p.xstride = h->xstride; p.f32_storage = h->esz == 4;
        p.x0 = x0; p.y0 = y0; p.yaw0 = yaw0;
        // the simulator starts from the un-rounded YAML pose (sim_node.py:361); the filter gets float args
        p.tx = h->cfg.init_x; p.ty = h->cfg.init_y; p.tyaw = h->cfg.init_yaw;
        HIP_TRY(slam::launch_ekf_init(p, h->stream));
    } else {
        slam::UkfInitParams p;
        p.P = (double*)h->dP; p.x = (double*)h->dx; p.n_sq = h->dnsq; p.v_age = h->dvage; p.M = h->dM; p.flags = h->dflags; p.timestep = h->dts; p.truth = h->dtruth; p.err_sum = h->derr;
        p.B = h->B; p.pstride = h->pstride; p.xstride = h->xstride;
        double s, c;   // x_t << x_0, y_0, cos(yaw_0), sin(yaw_0) with a float argument (ukf.cpp:33)
        slam::det_sincos((double)yaw0, &s, &c);
        p.x0 = x0; p.y0 = y0;
        p.c0 = h->cfg.ukf_float_trig ? (double)(float)c : c;
        p.s0 = h->cfg.ukf_float_trig ? (double)(float)s : s;
        p.tx = h->cfg.init_x; p.ty = h->cfg.init_y; p.tyaw = h->cfg.init_yaw;
        HIP_TRY(slam::launch_ukf_init(p, h->stream));
    }
    h->step = 0;
    h->inited = true;
    if (h->shadow) return slam_init(h->shadow, x0, y0, yaw0);
    return SLAM_OK;
}

int slam_set_map(slam_handle* h, const double* map_xy, int L) {
    if (!h || !map_xy || L <= 0) return fail(SLAM_ERR_ARG, "bad map");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    if (h->dmap) { HIP_TRY(hipStreamSynchronize(h->stream)); hipFree(h->dmap); h->dmap = nullptr; }
    HIP_TRY(hipMalloc(&h->dmap, sizeof(double) * 2 * (size_t)L));
    HIP_TRY(hipMemcpyAsync(h->dmap, map_xy, sizeof(double) * 2 * (size_t)L, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->kind == SLAM_UKF_LOC) {   // trueMapCallback: filter->map = [id, x, y] float32 (localization_node.cpp:152-156)
        std::vector<float> trip((size_t)3 * L);
        for (int i = 0; i < L; ++i) { trip[3 * i] = (float)i; trip[3 * i + 1] = (float)map_xy[2 * i]; trip[3 * i + 2] = (float)map_xy[2 * i + 1]; }
        if (h->dmapf) { hipFree(h->dmapf); h->dmapf = nullptr; }
        HIP_TRY(hipMalloc(&h->dmapf, sizeof(float) * trip.size()));
        HIP_TRY(hipMemcpy(h->dmapf, trip.data(), sizeof(float) * trip.size(), hipMemcpyHostToDevice));
    }
    h->L = L;
    h->hmap.assign(map_xy, map_xy + 2 * (size_t)L);
    if (h->shadow) { const int rs = slam_set_map(h->shadow, map_xy, L); if (rs) return rs; }
    return SLAM_OK;
}

int slam_step_dev(slam_handle* h, const float cmd[2], const float* d_meas, const int32_t* d_count, int k_stride) {
    if (!h || !cmd || !d_meas || !d_count || k_stride <= 0) return fail(SLAM_ERR_ARG, "bad argument");
    if (!h->inited) return fail(SLAM_ERR_STATE, "slam_init has not been called");
    HIP_TRY(hipSetDevice(h->device));
    if (h->shadow) {   // the tracked instance's slice of the message, ordered after what the caller enqueued on the handle's stream
        if (!h->shadow_ev) HIP_TRY(hipEventCreateWithFlags(&h->shadow_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(h->shadow_ev, h->stream));
        HIP_TRY(hipStreamWaitEvent(h->shadow->stream, h->shadow_ev, 0));
        const int rs = slam_step_dev(h->shadow, cmd, d_meas + (size_t)h->tracked * k_stride * 3, d_count + h->tracked, k_stride);
        if (rs) return rs;
        // ... and the other way round: the shadow reads the CALLER's buffers on its own stream, and the header promises that they
        // may be overwritten by work enqueued on the handle's stream after this call returns
        HIP_TRY(hipEventRecord(h->shadow_ev, h->shadow->stream));
        HIP_TRY(hipStreamWaitEvent(h->stream, h->shadow_ev, 0));
    }
    // Device buffers on the caller's stream: queueing is OPT-IN here (slam_set_lazy_steps / SLAM_LAZY_STEPS), because a queued
    // call enqueues only its device-to-device copy on the stream, not the step itself
    if (h->kind == SLAM_EKF_SLAM && h->lazy_explicit && h->lazy_max > 1 && h->run_chunk != 1 && !h->dump_meas &&
        k_stride <= slam::ekf_class_message_capacity(h->L_max)) {   // (messages that may be longer: one launch pair per step, launch_step)
        slam_handle::DevQueue& q = h->devq;
        if (!h->lazy_cmds.empty() || h->extq[h->extq_cur].n > 0 || (q.n > 0 && q.ks != k_stride)) FLUSH(h);   // earlier steps first
        const size_t B = (size_t)h->B;
        if (q.cap < h->lazy_max || q.ks != k_stride) {
            if (q.dmeas) { HIP_TRY(hipStreamSynchronize(h->stream)); hipFree(q.dmeas); hipFree(q.dcount); q.dmeas = nullptr; q.dcount = nullptr; }
            q.cap = h->lazy_max; q.ks = k_stride;
            HIP_TRY(hipMalloc(&q.dmeas, sizeof(float) * 3 * (size_t)k_stride * B * q.cap));
            HIP_TRY(hipMalloc(&q.dcount, sizeof(int32_t) * B * q.cap));
        }
        HIP_TRY(hipMemcpyAsync(q.dmeas + (size_t)q.n * B * k_stride * 3, d_meas, sizeof(float) * 3 * (size_t)k_stride * B, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(q.dcount + (size_t)q.n * B, d_count, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, h->stream));
        q.cmds.push_back(cmd[0]); q.cmds.push_back(cmd[1]);
        q.n += 1;
        return q.n >= h->lazy_max ? flush_dev(h) : SLAM_OK;
    }
    FLUSH(h);
    if (h->kind == SLAM_UKF_LOC && !h->dmapf) return fail(SLAM_ERR_STATE, "UKF_LOC needs the known map: call slam_set_map first (localization_node.cpp:113-116)");
    return launch_step(h, cmd, 0, d_meas, d_count, k_stride);
}

int slam_step(slam_handle* h, const float cmd[2], const float* meas, const int32_t* count, int k_stride) {
    if (!h || !cmd || !meas || !count || k_stride <= 0) return fail(SLAM_ERR_ARG, "bad argument");
    if (!h->inited) return fail(SLAM_ERR_STATE, "slam_init has not been called");
    HIP_TRY(hipSetDevice(h->device));
    if (h->shadow) {   // the tracked instance's slice of the message runs in the shadow at once
        const int rs = slam_step(h->shadow, cmd, meas + (size_t)h->tracked * k_stride * 3, count + h->tracked, k_stride);
        if (rs) return rs;
    }
    if (h->kind == SLAM_EKF_SLAM && h->lazy_max > 1 && h->run_chunk != 1) {
        const size_t B = (size_t)h->B;
        int kmax = 0;
        for (size_t b = 0; b < B; ++b) kmax = count[b] > kmax ? count[b] : kmax;   // 65 536 ints: ~20 us
        kmax = kmax < k_stride ? kmax : k_stride;
        if (kmax <= kExtQ) {
            if (!h->lazy_cmds.empty() || h->devq.n > 0) FLUSH(h);   // steps queued through the other entry points run first
            // a fill of the queue has ONE stride (detections per instance): the largest message this handle has seen so far, at
            // least 2 (a high-water mark: a stride per fill taken from its first message ended fills early whenever a larger
            // message followed, 46 instead of 65 M steps/s); a message above the mark ends the fill and raises it
            if (kmax > h->extq_ks) {
                if (h->extq[h->extq_cur].n > 0) FLUSH(h);
                h->extq_ks = kmax;
            }
            slam_handle::ExtQueue& q = h->extq[h->extq_cur];
            if (q.n == 0) q.ks = h->extq_ks;
            // capacity: the full stride kExtQ at once while that stays below 128 MB per queue (re-pinning 100 MB of host memory
            // costs ~40 ms each time the stride grows: 0.5 ms per step over a 300-step run); above that, the stride actually seen
            const size_t full = (size_t)3 * kExtQ * B * h->lazy_max;
            const size_t need = sizeof(float) * full <= ((size_t)128 << 20) ? full : (size_t)3 * q.ks * B * h->lazy_max;
            if (q.cap < h->lazy_max || q.meas_cap < need) {
                if (q.n > 0) FLUSH(h);
                if (q.in_use) { HIP_TRY(hipEventSynchronize(q.used)); q.in_use = false; }
                if (q.hmeas) { hipHostFree(q.hmeas); hipHostFree(q.hcount); hipHostFree(q.hcmds); hipFree(q.dmeas); hipFree(q.dcount); }
                q.hmeas = nullptr; q.hcount = nullptr; q.hcmds = nullptr; q.dmeas = nullptr; q.dcount = nullptr;
                q.cap = 0; q.meas_cap = 0;
                HIP_TRY(hipHostMalloc((void**)&q.hmeas, sizeof(float) * need, hipHostMallocNonCoherent));
                HIP_TRY(hipHostMalloc((void**)&q.hcount, sizeof(int32_t) * B * h->lazy_max, hipHostMallocNonCoherent));
                HIP_TRY(hipHostMalloc((void**)&q.hcmds, sizeof(float) * 2 * h->lazy_max, hipHostMallocNonCoherent));
                HIP_TRY(hipMalloc(&q.dmeas, sizeof(float) * need));
                HIP_TRY(hipMalloc(&q.dcount, sizeof(int32_t) * B * h->lazy_max));
                q.cap = h->lazy_max; q.meas_cap = need;
                if (!q.copied) {
                    HIP_TRY(hipEventCreateWithFlags(&q.copied, hipEventDisableTiming));
                    HIP_TRY(hipEventCreateWithFlags(&q.used, hipEventDisableTiming));
                }
            }
            if (q.n == 0 && q.in_use) { HIP_TRY(hipEventSynchronize(q.used)); q.in_use = false; }   // the launch that read this queue is done
            const int ks = q.ks;
            float* dst = q.hmeas + (size_t)q.n * B * ks * 3;
            int32_t* dcnt = q.hcount + (size_t)q.n * B;
            const int kc = k_stride < ks ? k_stride : ks;   // detections per instance that are packed
            // The count is clamped to what is packed, like the immediate path clamps to k_stride, so the kernel never
            // reads a slot this call did not fill.  The copy itself has a compile-time size (kc detections, all within the caller's
            // k_stride): a per-instance variable-length memcpy made the call six times slower (1.28 vs 0.2 ms at batch 65 536).
            auto pack = [&](auto kc_tag) {
                constexpr int KC = decltype(kc_tag)::value;
                host_parallel(B, [&](size_t b0, size_t b1) {
                    for (size_t b = b0; b < b1; ++b) {
                        const int cb = count[b];
                        dcnt[b] = cb < 0 ? 0 : (cb < KC ? cb : KC);
                        memcpy(dst + b * ks * 3, meas + b * (size_t)k_stride * 3, sizeof(float) * 3 * KC);
                    }
                });
            };
            switch (kc) {
                case 1: pack(std::integral_constant<int, 1>{}); break;
                case 2: pack(std::integral_constant<int, 2>{}); break;
                case 3: pack(std::integral_constant<int, 3>{}); break;
                default: pack(std::integral_constant<int, 4>{}); break;   // kc <= ks <= kExtQ = 4
            }
            q.hcmds[2 * q.n] = cmd[0]; q.hcmds[2 * q.n + 1] = cmd[1];
            q.n += 1;
            return (q.n >= h->lazy_max || eager_flush(h, q.n)) ? flush_ext(h) : SLAM_OK;
        }
    }
    FLUSH(h);
    if (h->kind == SLAM_UKF_LOC && !h->dmapf) return fail(SLAM_ERR_STATE, "UKF_LOC needs the known map: call slam_set_map first (localization_node.cpp:113-116)");
    // The caller's buffers may be reused right after return (ekf.cpp:64 copies the message), so the message is packed
    // into one of two PINNED staging buffers (only max_b count[b] detections per instance travel), copied on a separate
    // stream while the previous step's kernel is still running, and the kernel waits for the copy by event.  The host
    // blocks only when it gets two steps ahead of the GPU.
    slam_handle::Stage& s = h->stage[h->stage_next & 1];
    h->stage_next += 1;
    const size_t B = (size_t)h->B;
    int kmax = 1;
    for (size_t b = 0; b < B; ++b) kmax = count[b] > kmax ? count[b] : kmax;
    kmax = kmax < k_stride ? kmax : k_stride;
    if (!h->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    if (!s.copied) {
        HIP_TRY(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.used, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc((void**)&s.hcount, sizeof(int32_t) * B, hipHostMallocNonCoherent));   // CPU-cached pinned memory: fast to fill
        HIP_TRY(hipMalloc(&s.dcount, sizeof(int32_t) * B));
    }
    if (s.in_use) HIP_TRY(hipEventSynchronize(s.used));   // the kernel of two steps ago has consumed this buffer
    const size_t need = (size_t)3 * kmax * B;
    if (s.cap < need) {
        if (s.hmeas) hipHostFree(s.hmeas);
        if (s.dmeas) hipFree(s.dmeas);
        s.hmeas = nullptr; s.dmeas = nullptr; s.cap = 0;
        const size_t cap = (size_t)3 * (kmax < 8 && k_stride >= 8 ? 8 : kmax) * B;
        HIP_TRY(hipHostMalloc((void**)&s.hmeas, sizeof(float) * cap, hipHostMallocNonCoherent));
        HIP_TRY(hipMalloc(&s.dmeas, sizeof(float) * cap));
        s.cap = cap;
    }
    memcpy(s.hcount, count, sizeof(int32_t) * B);
    if (kmax == k_stride) {
        memcpy(s.hmeas, meas, sizeof(float) * need);
    } else {
        for (size_t b = 0; b < B; ++b)
            memcpy(s.hmeas + (size_t)3 * kmax * b, meas + (size_t)3 * k_stride * b, sizeof(float) * 3 * kmax);
    }
    HIP_TRY(hipMemcpyAsync(s.dcount, s.hcount, sizeof(int32_t) * B, hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(hipMemcpyAsync(s.dmeas, s.hmeas, sizeof(float) * need, hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(hipEventRecord(s.copied, h->copy_stream));
    HIP_TRY(hipStreamWaitEvent(h->stream, s.copied, 0));
    // (kmax = the longest message of this call: beyond the size class's capacity the instances concerned take the streamed kernel, launch_step)
    int rc = launch_step(h, cmd, 0, s.dmeas, s.dcount, kmax);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s.used, h->stream));
    s.in_use = true;
    return SLAM_OK;
}

int slam_step_sim(slam_handle* h, const float cmd[2]) {
    if (!h || !cmd) return fail(SLAM_ERR_ARG, "bad argument");
    if (!h->inited) return fail(SLAM_ERR_STATE, "slam_init has not been called");
    if (!h->dmap) return fail(SLAM_ERR_STATE, "slam_set_map has not been called");
    if (h->shadow) { const int rs = slam_step_sim(h->shadow, cmd); if (rs) return rs; }
    if (h->kind == SLAM_EKF_SLAM && h->lazy_max > 1 && !h->dump_meas && h->run_chunk != 1) {
        if (h->extq[h->extq_cur].n > 0 || h->devq.n > 0) FLUSH(h);   // measurement-driven steps queued before this one run first
        h->lazy_cmds.push_back(cmd[0]); h->lazy_cmds.push_back(cmd[1]);
        return (int)(h->lazy_cmds.size() / 2) >= h->lazy_max ? flush_lazy(h) : SLAM_OK;
    }
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    return launch_step(h, cmd, 1, nullptr, nullptr, 0);
}

int slam_queued_steps(const slam_handle* h) {
    if (!h) return 0;
    return (int)(h->lazy_cmds.size() / 2) + h->extq[h->extq_cur].n + h->devq.n;
}

int slam_set_lazy_steps(slam_handle* h, int n) {
    if (!h || n < 0) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    h->lazy_max = n;
    h->lazy_explicit = true;
    return SLAM_OK;
}

int slam_run_sim(slam_handle* h, const float* cmds, int T) {
    if (!h || !cmds || T < 0) return fail(SLAM_ERR_ARG, "bad argument");
    if (!h->inited) return fail(SLAM_ERR_STATE, "slam_init has not been called");
    if (!h->dmap) return fail(SLAM_ERR_STATE, "slam_set_map has not been called");
    if (h->shadow) { const int rs = slam_run_sim(h->shadow, cmds, T); if (rs) return rs; }
    FLUSH(h);
    return run_sim_now(h, cmds, T);
}

}  // extern "C"

static int run_sim_now(slam_handle* h, const float* cmds, int T) {
    if (T == 0) return SLAM_OK;
    if (h->predicted) return fail(SLAM_ERR_STATE, "a prediction stage is pending: call slam_update_dev before the next step");
    if (h->kind != SLAM_EKF_SLAM && !h->dump_meas && h->B >= h->ukf_split_min) {
        // UKF: two launches per timestep (LDS-bound eigen-sqrt, then the latency-heavier sigma-point kernel).  The two
        // halves of the batch run on two streams and drift apart, so one half's sqrt overlaps the other's step kernel.
        HIP_TRY(hipSetDevice(h->device));
        const int NP = h->ukf_parts < 2 ? 2 : (h->ukf_parts > 4 ? 4 : h->ukf_parts);
        for (int a = 0; a < NP - 1; ++a)
            if (!h->aux_stream[a]) {
                HIP_TRY(hipStreamCreateWithFlags(&h->aux_stream[a], hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&h->aux_ev[a + 1], hipEventDisableTiming));
            }
        if (!h->aux_ev[0]) HIP_TRY(hipEventCreateWithFlags(&h->aux_ev[0], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(h->aux_ev[0], h->stream));
        for (int a = 0; a < NP - 1; ++a) HIP_TRY(hipStreamWaitEvent(h->aux_stream[a], h->aux_ev[0], 0));
        const int per = (h->B + NP - 1) / NP;
        int long_cap = 0;
        if (const int rc = long_message_cap(h, 1, 0, &long_cap)) return rc;
        for (int t = 0; t < T; ++t) {
            slam::UkfStepParams p;
            fill_ukf_params(h, p, cmds + 2 * (size_t)t);
            p.sim = 1;
            p.long_mode = long_cap > 0 ? 1 : 0; p.long_cap = long_cap;
            for (int part = 0; part < NP; ++part) {
                p.b_off = part * per;
                p.b_cnt = h->B - p.b_off < per ? h->B - p.b_off : per;
                if (p.b_cnt <= 0) continue;
                hipStream_t st = part ? h->aux_stream[part - 1] : h->stream;
                HIP_TRY(slam::launch_ukf_sqrt(p, st));
                HIP_TRY(slam::launch_ukf_step(p, st));
            }
            std::swap(h->dP, h->dP2);
            h->step += 1;
        }
        for (int a = 0; a < NP - 1; ++a) {
            HIP_TRY(hipEventRecord(h->aux_ev[a + 1], h->aux_stream[a]));
            HIP_TRY(hipStreamWaitEvent(h->stream, h->aux_ev[a + 1], 0));
        }
        return SLAM_OK;
    }
    if (h->kind != SLAM_EKF_SLAM || h->dump_meas || h->run_chunk == 1) {
        // one launch (pair) per timestep
        HIP_TRY(hipSetDevice(h->device));
        for (int t = 0; t < T; ++t) {
            int rc = launch_step(h, cmds + 2 * (size_t)t, 1, nullptr, nullptr, 0);
            if (rc) return rc;
        }
        return SLAM_OK;
    }
    // EKF: every workgroup carries its instance through a whole chunk of timesteps, keeping x_t, the landmark ids,
    // the true pose and the thin rows/cols of P on chip; only the P stream touches HBM each step.
    HIP_TRY(hipSetDevice(h->device));
    if (h->cmds_cap < T) {
        if (h->dcmds) { HIP_TRY(hipStreamSynchronize(h->stream)); hipFree(h->dcmds); h->dcmds = nullptr; }
        HIP_TRY(hipMalloc(&h->dcmds, sizeof(float) * 2 * (size_t)T));
        h->cmds_cap = T;
    }
    HIP_TRY(hipMemcpyAsync(h->dcmds, cmds, sizeof(float) * 2 * (size_t)T, hipMemcpyHostToDevice, h->stream));
    const int chunk = h->run_chunk > 0 ? h->run_chunk : T;
    int long_cap = 0;   // a map with more landmarks than a message of this size class holds: the streamed kernel, a launch per timestep
    if (const int rc = long_message_cap(h, 1, 0, &long_cap)) return rc;
    for (int t0 = 0; t0 < T; t0 += chunk) {
        const int tc = T - t0 < chunk ? T - t0 : chunk;
        slam::EkfStepParams p;
        fill_params(h, p, cmds + 2 * (size_t)t0);
        p.sim = 1;
        p.long_mode = long_cap > 0 ? 1 : 0; p.long_cap = long_cap;
        p.cmds = h->dcmds + 2 * (size_t)t0;
        p.T = tc;
        HIP_TRY(slam::launch_ekf_step(p, h->waves_per_filter, h->esz == 4, h->stream));
        h->step += (uint32_t)tc;
    }
    return SLAM_OK;
}

static int flush_ext(slam_handle* h) {
    slam_handle::ExtQueue& q = h->extq[h->extq_cur];
    if (q.n == 0) return SLAM_OK;
    const int T = q.n;
    const size_t B = (size_t)h->B;
    // q.n is reset only after the launch has been enqueued: a failing flush keeps the queued timesteps (and fails again on the
    // next call) instead of dropping them silently
    HIP_TRY(hipSetDevice(h->device));
    if (!h->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    HIP_TRY(hipMemcpyAsync(q.dcount, q.hcount, sizeof(int32_t) * B * T, hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(hipMemcpyAsync(q.dmeas, q.hmeas, sizeof(float) * 3 * q.ks * B * T, hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(hipEventRecord(q.copied, h->copy_stream));
    if (h->cmds_cap < T) {
        if (h->dcmds) { HIP_TRY(hipStreamSynchronize(h->stream)); hipFree(h->dcmds); h->dcmds = nullptr; }
        HIP_TRY(hipMalloc(&h->dcmds, sizeof(float) * 2 * (size_t)T));
        h->cmds_cap = T;
    }
    HIP_TRY(hipMemcpyAsync(h->dcmds, q.hcmds, sizeof(float) * 2 * (size_t)T, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->stream, q.copied, 0));
    slam::EkfStepParams p;
    fill_params(h, p, q.hcmds);
    p.sim = 0;
    p.meas_in = q.dmeas; p.meas_count_in = q.dcount; p.k_stride_in = q.ks;
    p.cmds = h->dcmds;
    p.T = T;
    HIP_TRY(slam::launch_ekf_step(p, h->waves_per_filter, h->esz == 4, h->stream));
    q.n = 0;
    h->step += (uint32_t)T;
    HIP_TRY(hipEventRecord(q.used, h->stream));
    q.in_use = true;
    h->extq_cur ^= 1;
    return SLAM_OK;
}

static int flush_dev(slam_handle* h) {
    slam_handle::DevQueue& q = h->devq;
    if (q.n == 0) return SLAM_OK;
    const int T = q.n;
    HIP_TRY(hipSetDevice(h->device));
    if (h->cmds_cap < T) {
        if (h->dcmds) { HIP_TRY(hipStreamSynchronize(h->stream)); hipFree(h->dcmds); h->dcmds = nullptr; }
        HIP_TRY(hipMalloc(&h->dcmds, sizeof(float) * 2 * (size_t)T));
        h->cmds_cap = T;
    }
    // pageable source: the copy is staged by the runtime before the call returns, so q.cmds may be reused at once
    HIP_TRY(hipMemcpyAsync(h->dcmds, q.cmds.data(), sizeof(float) * 2 * (size_t)T, hipMemcpyHostToDevice, h->stream));
    slam::EkfStepParams p;
    fill_params(h, p, q.cmds.data());
    p.sim = 0;
    p.meas_in = q.dmeas; p.meas_count_in = q.dcount; p.k_stride_in = q.ks;
    p.cmds = h->dcmds;
    p.T = T;
    HIP_TRY(slam::launch_ekf_step(p, h->waves_per_filter, h->esz == 4, h->stream));
    q.n = 0;               // only now: a failed flush keeps its timesteps
    q.cmds.clear();
    h->step += (uint32_t)T;
    return SLAM_OK;
}

static int flush_lazy(slam_handle* h) {
    {
        const int rc = flush_dev(h);
        if (rc) return rc;
    }
    if (!h->lazy_cmds.empty()) {
        std::vector<float> c;
        c.swap(h->lazy_cmds);
        const int rc = run_sim_now(h, c.data(), (int)(c.size() / 2));
        if (rc) { if (h->lazy_cmds.empty()) h->lazy_cmds.swap(c); return rc; }   // a failed launch keeps the queued commands
    }
    return flush_ext(h);
}

extern "C" {

// UKF::predictionStage / UKF::updateStage (filter.h:187-188, ukf.cpp:197-291) as two calls.  predictionStage only
// writes members that updateStage consumes (sqtP, X, X_pred, x_pred, P_pred); x_t / P_t change when updateStage
// finishes (ukf.cpp:289-290).  So the split is: predict = nearestSPD + sqrt (the kernel that dominates the step) with
// the command remembered; update = the fused sigma-point / update / insertion kernel with that command.  The pair is
// bit-identical to slam_step_dev.
int slam_predict(slam_handle* h, const float cmd[2]) {
    if (!h || !cmd) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    if (h->kind == SLAM_EKF_SLAM) return fail(SLAM_ERR_UNSUPPORTED, "EKF has no separate prediction stage: EKF::update does both (ekf.cpp:37-179); use slam_step");
    if (!h->inited) return fail(SLAM_ERR_STATE, "slam_init has not been called");
    if (h->predicted) return fail(SLAM_ERR_STATE, "slam_predict called twice without slam_update_dev");
    HIP_TRY(hipSetDevice(h->device));
    slam::UkfStepParams p;
    fill_ukf_params(h, p, cmd);
    HIP_TRY(slam::launch_ukf_sqrt(p, h->stream));
    h->pred_cmd[0] = cmd[0]; h->pred_cmd[1] = cmd[1];
    h->predicted = true;
    return SLAM_OK;
}
int slam_update_dev(slam_handle* h, const float* d_meas, const int32_t* d_count, int k_stride) {
    if (!h || k_stride < 0 || (k_stride > 0 && (!d_meas || !d_count))) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    if (h->kind == SLAM_EKF_SLAM) return fail(SLAM_ERR_UNSUPPORTED, "EKF has no separate update stage (ekf.cpp:37-179); use slam_step");
    if (!h->predicted) return fail(SLAM_ERR_STATE, "slam_update_dev needs a preceding slam_predict");
    HIP_TRY(hipSetDevice(h->device));
    if (k_stride == 0) {   // empty message for every instance
        int rc = ensure_meas_buffers(h, 1);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(h->dcount, 0, sizeof(int32_t) * (size_t)h->B, h->stream));
        d_meas = h->dmeas; d_count = h->dcount; k_stride = 1;
    }
    int long_cap = 0;
    if (const int rc = long_message_cap(h, 0, k_stride, &long_cap)) return rc;
    slam::UkfStepParams p;
    fill_ukf_params(h, p, h->pred_cmd);
    p.sim = 0;
    p.long_mode = long_cap > 0 ? 1 : 0; p.long_cap = long_cap;
    p.meas_in = d_meas; p.meas_count_in = d_count; p.k_stride_in = k_stride;
    HIP_TRY(slam::launch_ukf_step(p, h->stream));
    std::swap(h->dP, h->dP2);
    h->step += 1;
    h->predicted = false;
    return SLAM_OK;
}

// UKFState.X (ukf.cpp:92-101): the sigma points of the last predictionStage, column-major n x (2n+1):
// X = [x, x + sqtP(:,i), x - sqtP(:,i)] (ukf.cpp:214-219) around the x_t that step started from.
int slam_get_sigma_points(slam_handle* h, int inst, double* X, int32_t* rows, int32_t* cols) {
    if (!h || inst < 0 || inst >= h->B) return fail(SLAM_ERR_ARG, "bad instance");
    FLUSH(h);
    if (h->kind == SLAM_EKF_SLAM) return fail(SLAM_ERR_UNSUPPORTED, "sigma points exist for the UKF kinds only");
    if (!h->inited) return fail(SLAM_ERR_STATE, "slam_init has not been called");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int32_t n = 0;
    HIP_TRY(hipMemcpy(&n, h->dnsq + inst, sizeof(int32_t), hipMemcpyDeviceToHost));
    n = n < 0 ? 0 : (n > h->n_max ? h->n_max : n);
    if (rows) *rows = n;
    if (cols) *cols = n > 0 ? 2 * n + 1 : 0;
    if (!X || n <= 0) return SLAM_OK;
    std::vector<double> x(n), S((size_t)n * n);
    HIP_TRY(hipMemcpy(x.data(), h->dxprev + (size_t)inst * h->xstride, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(S.data(), h->dsq + (size_t)inst * h->pstride, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost));
    for (int r = 0; r < n; ++r) X[r] = x[r];
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < n; ++r) {
            X[(size_t)(1 + i) * n + r] = x[r] + S[(size_t)r * n + i];
            X[(size_t)(1 + n + i) * n + r] = x[r] - S[(size_t)r * n + i];
        }
    return SLAM_OK;
}

// device -> host copy of `count` stored elements, widened to double when the storage type is fp32
static int fetch_elems(slam_handle* h, double* dst, const void* dbase, size_t elem_offset, size_t count) {
    if (h->esz == 8) {
        HIP_TRY(hipMemcpy(dst, (const char*)dbase + elem_offset * 8, count * 8, hipMemcpyDeviceToHost));
    } else {
        std::vector<float> tmp(count);
        HIP_TRY(hipMemcpy(tmp.data(), (const char*)dbase + elem_offset * 4, count * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < count; ++i) dst[i] = (double)tmp[i];
    }
    return SLAM_OK;
}

int slam_get_state(slam_handle* h, int inst, double* x, double* P, int32_t* M, int32_t* ids, int32_t* ts) {
    if (!h || inst < 0 || inst >= h->B) return fail(SLAM_ERR_ARG, "bad instance");
    if (h->shadow && inst == h->tracked) return slam_get_state(h->shadow, 0, x, P, M, ids, ts);   // the batch's queue stays queued
    FLUSH(h);
    if (!h->inited) return fail(SLAM_ERR_STATE, "slam_init has not been called");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int32_t m = 0;
    HIP_TRY(hipMemcpy(&m, h->dM + inst, sizeof(int32_t), hipMemcpyDeviceToHost));
    m = m < 0 ? 0 : (m > h->L_max ? h->L_max : m);   // never index past the caller's n_max-sized buffers
    const int n = h->base + 2 * m;
    if (M) *M = m;
    int rc;
    if (x && (rc = fetch_elems(h, x, h->dx, (size_t)inst * h->xstride, n))) return rc;
    if (P) {   // rows of the device matrix are ld elements apart (pad columns beyond n); the caller gets packed n x n
        const int ld = h->kind == SLAM_EKF_SLAM ? slam::ekf_ld(n, h->esz) : n;
        std::vector<double> tmp((size_t)n * ld);
        if ((rc = fetch_elems(h, tmp.data(), h->dP, (size_t)inst * h->pstride, (size_t)n * ld))) return rc;
        for (int r = 0; r < n; ++r) memcpy(P + (size_t)r * n, tmp.data() + (size_t)r * ld, sizeof(double) * n);
    }
    if (ids && m > 0) HIP_TRY(hipMemcpy(ids, h->dids + (size_t)inst * h->L_max, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
    if (ts) HIP_TRY(hipMemcpy(ts, h->dts + inst, sizeof(int32_t), hipMemcpyDeviceToHost));
    return SLAM_OK;
}

int slam_get_poses(slam_handle* h, double* poses) {
    if (!h || !poses) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->esz == 8) {
        HIP_TRY(hipMemcpy2D(poses, sizeof(double) * 3, h->dx, sizeof(double) * h->xstride, sizeof(double) * 3, h->B, hipMemcpyDeviceToHost));
    } else {
        std::vector<float> tmp((size_t)3 * h->B);
        HIP_TRY(hipMemcpy2D(tmp.data(), sizeof(float) * 3, h->dx, sizeof(float) * h->xstride, sizeof(float) * 3, h->B, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); ++i) poses[i] = (double)tmp[i];
    }
    return SLAM_OK;
}

static int copy_out(slam_handle* h, void* dst, const void* src, size_t bytes) {
    if (!h || !dst) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return SLAM_OK;
}
int slam_get_landmark_counts(slam_handle* h, int32_t* M) { return copy_out(h, M, h ? h->dM : nullptr, h ? sizeof(int32_t) * (size_t)h->B : 0); }
int slam_get_truth(slam_handle* h, double* t) { return copy_out(h, t, h ? h->dtruth : nullptr, h ? sizeof(double) * 3 * (size_t)h->B : 0); }
int slam_status(slam_handle* h, int32_t* f) { return copy_out(h, f, h ? h->dflags : nullptr, h ? sizeof(int32_t) * (size_t)h->B : 0); }

int slam_get_last_meas(slam_handle* h, float* meas, int32_t* count, int k_stride) {
    if (!h || !meas || !count || k_stride <= 0) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    if (!h->dump_meas || h->k_stride < k_stride) {
        // enable the dump for subsequent slam_step_sim calls; nothing recorded yet for past steps
        int rc = ensure_meas_buffers(h, k_stride);
        if (rc) return rc;
        if (!h->dump_meas) {
            h->dump_meas = true;
            memset(count, 0, sizeof(int32_t) * (size_t)h->B);
            return SLAM_OK;
        }
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy2D(meas, sizeof(float) * 3 * k_stride, h->dmeas, sizeof(float) * 3 * h->k_stride,
                        sizeof(float) * 3 * k_stride, h->B, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(count, h->dcount, sizeof(int32_t) * (size_t)h->B, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

int slam_error_stats(slam_handle* h, double* avg) {
    if (!h || !avg) return fail(SLAM_ERR_ARG, "bad argument");
    if (!h->inited) return fail(SLAM_ERR_STATE, "slam_init has not been called");
    std::vector<int32_t> ts(h->B);
    int rc = copy_out(h, avg, h->derr, sizeof(double) * (size_t)h->B);
    if (rc) return rc;
    rc = copy_out(h, ts.data(), h->dts, sizeof(int32_t) * (size_t)h->B);
    if (rc) return rc;
    for (int i = 0; i < h->B; ++i) avg[i] = ts[i] > 0 ? avg[i] / ts[i] : 0.0;  // sum(errors) / num_iters
    return SLAM_OK;
}

extern "C" int slam_internal_error_stats_dev(slam_handle* h, double* d_out, long long pad) {
    if (!h || !d_out || pad < h->B) return fail(SLAM_ERR_ARG, "bad argument");
    if (!h->inited) return fail(SLAM_ERR_STATE, "slam_init has not been called");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(slam::launch_avg_error(h->derr, h->dts, h->B, (int)pad, d_out, h->stream));
    HIP_TRY(h->kind == SLAM_EKF_SLAM ? hipStreamSynchronize(h->stream) : hipDeviceSynchronize());
    return SLAM_OK;
}

// ---- checkpoint / resume (the reference keeps the filter only in memory) ------------------------------
namespace {
struct CkptHeader {
    char magic[8];            // "SLAMCKP1"
    int32_t kind, B, L_max, dtype, n_max, pstride, xstride, esz;
    uint32_t step;
    int32_t inited, reserved0;
    uint64_t seed;
    int64_t inst0;
};
struct CkptItem { void* dev; size_t bytes; };

std::vector<CkptItem> ckpt_items(slam_handle* h) {
    const size_t B = (size_t)h->B, e = (size_t)h->esz;
    std::vector<CkptItem> it = {
        {h->dP, e * B * h->pstride}, {h->dx, e * B * h->xstride}, {h->dM, sizeof(int32_t) * B}, {h->dids, sizeof(int32_t) * B * h->L_max},
        {h->dflags, sizeof(int32_t) * B}, {h->dts, sizeof(int32_t) * B}, {h->dtruth, sizeof(double) * B * 3}, {h->derr, sizeof(double) * B},
    };
    if (h->kind != SLAM_EKF_SLAM) {   // UKF: the square root of the last prediction stage and the warm-start eigenvectors
        it.push_back({h->dsq, sizeof(double) * B * h->pstride}); it.push_back({h->dnsq, sizeof(int32_t) * B});
        it.push_back({h->dxprev, sizeof(double) * B * h->xstride}); it.push_back({h->dvt, sizeof(double) * B * h->pstride});
        it.push_back({h->dvage, sizeof(int32_t) * B});
    }
    return it;
}
}  // namespace

int slam_save_state(slam_handle* h, const char* path) {
    if (!h || !path) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    if (h->predicted) return fail(SLAM_ERR_STATE, "a prediction stage is pending: call slam_update_dev first");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(h->kind == SLAM_EKF_SLAM ? hipStreamSynchronize(h->stream) : hipDeviceSynchronize());
    FILE* f = fopen(path, "wb");
    if (!f) return fail(SLAM_ERR_IO, "cannot write %s", path);
    CkptHeader hd;
    memset(&hd, 0, sizeof(hd));
    memcpy(hd.magic, "SLAMCKP1", 8);
    hd.kind = h->kind; hd.B = h->B; hd.L_max = h->L_max; hd.dtype = h->dtype; hd.n_max = h->n_max; hd.pstride = h->pstride; hd.xstride = h->xstride;
    hd.esz = h->esz; hd.step = h->step; hd.inited = h->inited ? 1 : 0; hd.seed = h->seed; hd.inst0 = h->inst0;
    bool ok = fwrite(&hd, sizeof(hd), 1, f) == 1;
    std::vector<char> buf((size_t)64 << 20);
    for (const CkptItem& it : ckpt_items(h))
        for (size_t off = 0; ok && off < it.bytes; off += buf.size()) {
            const size_t nb = it.bytes - off < buf.size() ? it.bytes - off : buf.size();
            if (hipMemcpy(buf.data(), (const char*)it.dev + off, nb, hipMemcpyDeviceToHost) != hipSuccess) { fclose(f); return fail(SLAM_ERR_HIP, "copying the state to the host failed"); }
            ok = fwrite(buf.data(), 1, nb, f) == nb;
        }
    ok = (fclose(f) == 0) && ok;
    return ok ? SLAM_OK : fail(SLAM_ERR_IO, "short write to %s", path);
}

int slam_load_state(slam_handle* h, const char* path) {
    if (!h || !path) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(h->kind == SLAM_EKF_SLAM ? hipStreamSynchronize(h->stream) : hipDeviceSynchronize());
    FILE* f = fopen(path, "rb");
    if (!f) return fail(SLAM_ERR_IO, "cannot open %s", path);
    CkptHeader hd;
    if (fread(&hd, sizeof(hd), 1, f) != 1 || memcmp(hd.magic, "SLAMCKP1", 8) != 0) { fclose(f); return fail(SLAM_ERR_IO, "%s is not a state file of this library", path); }
    if (hd.kind != h->kind || hd.B != h->B || hd.L_max != h->L_max || hd.dtype != h->dtype || hd.pstride != h->pstride || hd.xstride != h->xstride || hd.esz != h->esz) {
        fclose(f);
        return fail(SLAM_ERR_ARG, "%s holds kind %d, batch %d, L_max %d, dtype %d; the handle is kind %d, batch %d, L_max %d, dtype %d", path, hd.kind, hd.B, hd.L_max,
                    hd.dtype, h->kind, h->B, h->L_max, h->dtype);
    }
    if (hd.n_max != h->n_max) { fclose(f); return fail(SLAM_ERR_ARG, "%s was written for a state capacity of %d, the handle has %d", path, hd.n_max, h->n_max); }
    // The step kernels trust the per-instance counters (n = 3 + 2 M sizes every loop over LDS and the instance's slab), so a
    // corrupted or hand-made file must be refused BEFORE anything reaches the device: landmark counts, the size of the UKF's
    // stored square root and the timesteps are checked on the host first.
    {
        const std::vector<CkptItem> items = ckpt_items(h);
        std::vector<size_t> off(items.size() + 1, sizeof(hd));
        for (size_t i = 0; i < items.size(); ++i) off[i + 1] = off[i] + items[i].bytes;
        std::vector<int32_t> col((size_t)h->B);
        auto column = [&](size_t item, const char* what, int lo, int hi) -> int {
            if (fseek(f, (long)off[item], SEEK_SET) != 0 || fread(col.data(), sizeof(int32_t), col.size(), f) != col.size()) return fail(SLAM_ERR_IO, "%s is truncated", path);
            for (size_t b = 0; b < col.size(); ++b)
                if (col[b] < lo || col[b] > hi) return fail(SLAM_ERR_ARG, "%s: %s of instance %zu is %d, outside [%d, %d]", path, what, b, col[b], lo, hi);
            return SLAM_OK;
        };
        int rc = column(2, "the landmark count", 0, h->L_max);
        if (!rc) rc = column(5, "the timestep", 0, INT32_MAX);
        if (!rc && h->kind != SLAM_EKF_SLAM) rc = column(9, "the size of the stored square root", 0, h->n_max);
        // -1 is the cold-start marker (ukf_init_kernel, and both sqrt kernels after SLAM_INST_SQRT_FAILED); the kernels count the
        // age up to kWarmMaxAge = 100 and start cold from there, so nothing a run can produce lies outside [-1, 100].
        if (!rc && h->kind != SLAM_EKF_SLAM) rc = column(12, "the age of the warm-start eigenvectors", -1, 100);
        if (rc) { fclose(f); return rc; }
        if (fseek(f, (long)sizeof(hd), SEEK_SET) != 0) { fclose(f); return fail(SLAM_ERR_IO, "cannot rewind %s", path); }
    }
    std::vector<char> buf((size_t)64 << 20);
    for (const CkptItem& it : ckpt_items(h))
        for (size_t off = 0; off < it.bytes; off += buf.size()) {
            const size_t nb = it.bytes - off < buf.size() ? it.bytes - off : buf.size();
            if (fread(buf.data(), 1, nb, f) != nb) { fclose(f); return fail(SLAM_ERR_IO, "%s is truncated", path); }
            if (hipMemcpy((char*)it.dev + off, buf.data(), nb, hipMemcpyHostToDevice) != hipSuccess) { fclose(f); return fail(SLAM_ERR_HIP, "copying the state to the device failed"); }
        }
    fclose(f);
    h->step = hd.step; h->inited = hd.inited != 0; h->seed = hd.seed; h->inst0 = hd.inst0;
    h->predicted = false;
    if (h->shadow) {   // the tracked instance restarts from the loaded state
        const int tr = h->tracked;
        const int rc = slam_track_instance(h, tr);
        if (rc) return rc;
    }
    return SLAM_OK;
}

int slam_track_instance(slam_handle* h, int inst) {
    if (!h || inst >= h->B) return fail(SLAM_ERR_ARG, "bad instance");
    FLUSH(h);
    if (h->predicted) return fail(SLAM_ERR_STATE, "a prediction stage is pending: call slam_update_dev first");
    if (h->shadow) { slam_destroy(h->shadow); h->shadow = nullptr; h->tracked = -1; }
    if (inst < 0) return SLAM_OK;
    if (h->kind != SLAM_EKF_SLAM) return fail(SLAM_ERR_UNSUPPORTED, "EKF handles only: the UKF launches every step at the call, there is no queue a getter would have to run");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    slam_handle* s = nullptr;
    int rc = slam_create(&h->cfg, h->kind, 1, h->L_max, h->dtype, h->device, &s);
    if (rc) return rc;
    s->lazy_max = 0;                     // every step of the shadow is launched at the call
    s->seed = h->seed; s->inst0 = h->inst0 + inst;
    s->range_max = h->range_max; s->fov_min = h->fov_min; s->fov_max = h->fov_max;
    s->waves_per_filter = h->waves_per_filter;
    if (!h->hmap.empty()) rc = slam_set_map(s, h->hmap.data(), h->L);
    if (!rc && h->inited) {
        // the instance's state as it is now: same slab layout (pstride / xstride depend on L_max and dtype only)
        const size_t b = (size_t)inst, e = (size_t)h->esz;
        hipError_t errs[] = {
            hipMemcpy(s->dP, (const char*)h->dP + b * h->pstride * e, (size_t)h->pstride * e, hipMemcpyDeviceToDevice),
            hipMemcpy(s->dx, (const char*)h->dx + b * h->xstride * e, (size_t)h->xstride * e, hipMemcpyDeviceToDevice),
            hipMemcpy(s->dM, h->dM + b, sizeof(int32_t), hipMemcpyDeviceToDevice),
            hipMemcpy(s->dids, h->dids + b * h->L_max, sizeof(int32_t) * h->L_max, hipMemcpyDeviceToDevice),
            hipMemcpy(s->dflags, h->dflags + b, sizeof(int32_t), hipMemcpyDeviceToDevice),
            hipMemcpy(s->dts, h->dts + b, sizeof(int32_t), hipMemcpyDeviceToDevice),
            hipMemcpy(s->dtruth, h->dtruth + 3 * b, sizeof(double) * 3, hipMemcpyDeviceToDevice),
            hipMemcpy(s->derr, h->derr + b, sizeof(double), hipMemcpyDeviceToDevice),
            h->dsq ? hipMemcpy(s->dsq, h->dsq + b * h->pstride, sizeof(double) * h->pstride, hipMemcpyDeviceToDevice) : hipSuccess,
            h->dnsq ? hipMemcpy(s->dnsq, h->dnsq + b, sizeof(int32_t), hipMemcpyDeviceToDevice) : hipSuccess,
            h->dxprev ? hipMemcpy(s->dxprev, h->dxprev + b * h->xstride, sizeof(double) * h->xstride, hipMemcpyDeviceToDevice) : hipSuccess,
            h->dvt ? hipMemcpy(s->dvt, h->dvt + b * h->pstride, sizeof(double) * h->pstride, hipMemcpyDeviceToDevice) : hipSuccess,
            h->dvage ? hipMemcpy(s->dvage, h->dvage + b, sizeof(int32_t), hipMemcpyDeviceToDevice) : hipSuccess,
        };
        for (hipError_t ee : errs)
            if (ee != hipSuccess) { slam_destroy(s); return fail(SLAM_ERR_HIP, "copying the tracked instance -> %s", hipGetErrorString(ee)); }
        s->step = h->step; s->inited = true;
    }
    if (rc) { slam_destroy(s); return rc; }
    h->shadow = s; h->tracked = inst;
    return SLAM_OK;
}

int slam_sync(slam_handle* h) {
    if (!h) return fail(SLAM_ERR_ARG, "NULL handle");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->eager_target = h->eager_init;   // the pipeline is drained: the next burst of calls fills it from small launches again
    return SLAM_OK;
}
int slam_batch(const slam_handle* h) { return h ? h->B : 0; }
int slam_state_dim_max(const slam_handle* h) { return h ? h->n_max : 0; }

int slam_set_run_chunk(slam_handle* h, int steps_per_launch) {
    if (!h || steps_per_launch < 0) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    h->run_chunk = steps_per_launch;
    return SLAM_OK;
}

// Per-timestep stamps of the EKF multi-step kernel (flag 32) / phase cycle counters (flag 4) for subsequent launches; 0 = off.
// The ablation bits are honoured by -DSLAM_ABLATE builds only.
int slam_set_debug_flags(slam_handle* h, int flags) {
    if (!h) return fail(SLAM_ERR_ARG, "NULL handle");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
#ifndef SLAM_ABLATE
    flags &= (4 | 32 | 128);   // 128: tests only, provokes the watchdog of the EKF step kernel
#endif
    if ((flags & (4 | 32)) && !h->dprof) {
        const size_t bytes = sizeof(unsigned long long) * slam::kEkfProfSlots * (size_t)h->B;
        HIP_TRY(hipMalloc(&h->dprof, bytes));
        HIP_TRY(hipMemset(h->dprof, 0, bytes));
    }
    h->dbg = flags;
    return SLAM_OK;
}

int slam_variant_available(int L_max, int dtype, int variant) { return slam::ekf_variant_available(L_max, dtype == SLAM_F32, variant); }

int slam_k_histogram(slam_handle* h, uint64_t out[8], int reset) {
    if (!h || !out) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(h->kind == SLAM_EKF_SLAM ? hipStreamSynchronize(h->stream) : hipDeviceSynchronize());
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "histogram element size");
    HIP_TRY(hipMemcpy(out, h->dkhist, sizeof(uint64_t) * 8, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(h->dkhist, 0, sizeof(uint64_t) * 8));
    return SLAM_OK;
}

int slam_traffic_counters(slam_handle* h, uint64_t out[4], int reset) {
    if (!h || !out) return fail(SLAM_ERR_ARG, "bad argument");
    if (h->kind != SLAM_EKF_SLAM) return fail(SLAM_ERR_UNSUPPORTED, "the traffic counters are kept by the EKF step kernels");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out, h->dkhist + slam::kEkfTrafficSlot, sizeof(uint64_t) * 4, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(h->dkhist + slam::kEkfTrafficSlot, 0, sizeof(uint64_t) * 4));
    return SLAM_OK;
}

// Both counter sets zeroed IN STREAM ORDER, without a host round trip (round 6): a measurement that resets the counters between its
// warm-up and its timed launches no longer leaves the device idle for the reads and resets - profiles/r06a/launch_edges.txt: an idle
// device before a 20-step launch costs it 9 % (clocks).
int slam_reset_counters_async(slam_handle* h) {
    if (!h) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemsetAsync(h->dkhist, 0, sizeof(unsigned long long) * 16, h->stream));
    return SLAM_OK;
}

int slam_kernel_info(slam_handle* h, int multi_step, char* name, int name_cap, int32_t out[5]) {
    if (!h || (!name && !out)) return fail(SLAM_ERR_ARG, "bad argument");
    if (h->kind != SLAM_EKF_SLAM) return fail(SLAM_ERR_UNSUPPORTED, "EKF handles only");
    HIP_TRY(hipSetDevice(h->device));
    slam::EkfKernelInfo ki;
    HIP_TRY(slam::ekf_kernel_info(h->L_max, h->B, h->waves_per_filter, h->esz == 4, multi_step ? 1 : 0, &ki));
    if (name && name_cap > 0) { strncpy(name, ki.name, (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    if (out) {
        hipDeviceProp_t pr;
        HIP_TRY(hipGetDeviceProperties(&pr, h->device));
        out[0] = ki.lds_bytes; out[1] = ki.vgprs; out[2] = ki.threads; out[3] = ki.wg_per_cu; out[4] = pr.multiProcessorCount;
    }
    return SLAM_OK;
}

int slam_ukf_sweep_stats(slam_handle* h, uint64_t out[2], int reset) {
    if (!h || !out) return fail(SLAM_ERR_ARG, "bad argument");
    if (h->kind == SLAM_EKF_SLAM) return fail(SLAM_ERR_STATE, "UKF handles only");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());   // the UKF splits large batches over two streams
    HIP_TRY(hipMemcpy(out, h->dkhist + 8, sizeof(uint64_t) * 2, hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(h->dkhist + 8, 0, sizeof(uint64_t) * 2));
    return SLAM_OK;
}

int slam_algorithmic_bytes(slam_handle* h, double* bytes) {
    if (!h || !bytes) return fail(SLAM_ERR_ARG, "bad argument");
    if (!h->inited) return fail(SLAM_ERR_STATE, "slam_init has not been called");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemsetAsync(h->dscalar, 0, sizeof(double), h->stream));
    HIP_TRY(slam::launch_algorithmic_bytes(h->dM, h->B, h->base, h->esz, h->dscalar, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(bytes, h->dscalar, sizeof(double), hipMemcpyDeviceToHost));
    return SLAM_OK;
}

// debug only (not declared in slam_batch.h): per-block phase cycles of the LAST launch, summed over blocks
// (SLAM_DEBUG_FLAGS & 4)
int slam_debug_read_prof(slam_handle* h, unsigned long long* out) {
    if (!h || !out) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!h->dprof) return fail(SLAM_ERR_STATE, "SLAM_DEBUG_FLAGS has no timer bit (4 / 32) set");
    const size_t S = h->kind == SLAM_EKF_SLAM ? slam::kEkfProfSlots : 16;   // the UKF kernels use 16 slots per block
    std::vector<unsigned long long> buf(S * h->B);
    HIP_TRY(hipMemcpy(buf.data(), h->dprof, sizeof(unsigned long long) * buf.size(), hipMemcpyDeviceToHost));
    const int nslot = h->kind == SLAM_EKF_SLAM ? 64 : 16;   // out[64] for the EKF (slots 16.. = decoupled loop, 40.. = split control), out[16] for the UKF
    for (int i = 0; i < nslot; ++i) out[i] = 0;
    for (int b = 0; b < h->B; ++b)
        for (int i = 0; i < nslot; ++i) out[i] += buf[S * b + i];
    return SLAM_OK;
}

// debug only: the raw [B][16] buffer
int slam_debug_read_prof_raw(slam_handle* h, unsigned long long* out) {
    if (!h || !out) return fail(SLAM_ERR_ARG, "bad argument");
    FLUSH(h);
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (!h->dprof) return fail(SLAM_ERR_STATE, "SLAM_DEBUG_FLAGS has no timer bit (4 / 32) set");
    const size_t S = h->kind == SLAM_EKF_SLAM ? slam::kEkfProfSlots : 16;   // out: [B][S]
    HIP_TRY(hipMemcpy(out, h->dprof, sizeof(unsigned long long) * S * (size_t)h->B, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

// device math probe for the bit-exactness tests (a, b, out are HOST arrays; out has 8*n doubles)
int slam_math_probe(const double* a, const double* b, double* out, int n, int device) {
    if (!a || !b || !out || n <= 0) return fail(SLAM_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device));
    double *da, *db, *dout;
    HIP_TRY(hipMalloc(&da, sizeof(double) * n));
    HIP_TRY(hipMalloc(&db, sizeof(double) * n));
    HIP_TRY(hipMalloc(&dout, sizeof(double) * 8 * (size_t)n));
    HIP_TRY(hipMemcpy(da, a, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b, sizeof(double) * n, hipMemcpyHostToDevice));
    HIP_TRY(slam::launch_math_probe(da, db, dout, n, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout, sizeof(double) * 8 * (size_t)n, hipMemcpyDeviceToHost));
    hipFree(da); hipFree(db); hipFree(dout);
    return SLAM_OK;
}

}  // extern "C"
