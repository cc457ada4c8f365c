// multi_capi.cpp — include/slam_multi.h: the global batch over several GPUs from ONE host process / thread.
// One slam_handle (its own stream, its own device memory) per device; every call loops over the shards and only enqueues,
// so the devices work concurrently.  The one collective of a run is the gather of the error statistics.
#include "../../include/slam_multi.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <vector>

#include "capi_internal.h"

struct slam_multi {
    std::vector<slam_handle*> h;
    std::vector<int> dev;
    std::vector<int64_t> first, count;
    int64_t B = 0;
    // RCCL (mode 1 of slam_multi_error_stats), loaded lazily
    void* rccl = nullptr;
    std::vector<void*> comms;
    std::vector<hipStream_t> cstream;
    std::vector<double*> dsend, drecv;
    int64_t pad = 0;
};

namespace {
#define MULTI_ALL(call)                          \
    do {                                         \
        if (!m) return slam_internal_fail(SLAM_ERR_ARG, "NULL handle"); \
        for (size_t s_ = 0; s_ < m->h.size(); ++s_) { \
            slam_handle* hs = m->h[s_];          \
            const int rc_ = (call);              \
            if (rc_) return rc_;                 \
        }                                        \
        return SLAM_OK;                          \
    } while (0)

// minimal RCCL surface (rccl.h is not needed at build time; the ABI of these four calls is NCCL's)
typedef int (*nccl_comm_init_all_t)(void** comms, int ndev, const int* devlist);
typedef int (*nccl_all_gather_t)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream);
typedef int (*nccl_group_t)(void);
typedef int (*nccl_comm_destroy_t)(void* comm);
constexpr int kNcclFloat64 = 8;   // ncclDouble
}  // namespace

extern "C" {

int slam_shard_range(int64_t global_batch, int shard, int world, int64_t* first, int64_t* count) {
    if (global_batch < 0 || world <= 0 || shard < 0 || shard >= world || !first || !count) return slam_internal_fail(SLAM_ERR_ARG, "bad shard arguments");
    const int64_t base = global_batch / world, extra = global_batch % world;
    *first = shard * base + (shard < extra ? shard : extra);
    *count = base + (shard < extra ? 1 : 0);
    return SLAM_OK;
}

int slam_multi_create(const slam_config* cfg, int kind, int64_t global_batch, int L_max, int dtype, const int* devices, int n_devices,
                      slam_multi** out) {
    if (!cfg || !devices || !out || n_devices <= 0 || global_batch < n_devices) return slam_internal_fail(SLAM_ERR_ARG, "bad argument (every device needs at least one instance)");
    for (int a = 0; a < n_devices; ++a)
        for (int b = a + 1; b < n_devices; ++b)
            if (devices[a] == devices[b]) return slam_internal_fail(SLAM_ERR_ARG, "device %d listed twice", devices[a]);
    slam_multi* m = new slam_multi();
    m->B = global_batch;
    for (int s = 0; s < n_devices; ++s) {
        int64_t f = 0, c = 0;
        slam_shard_range(global_batch, s, n_devices, &f, &c);
        slam_handle* h = nullptr;
        int rc = slam_create(cfg, kind, (int)c, L_max, dtype, devices[s], &h);
        if (!rc) rc = slam_set_instance_offset(h, f);
        if (rc) {
            if (h) slam_destroy(h);
            slam_multi_destroy(m);
            return rc;
        }
        m->h.push_back(h); m->dev.push_back(devices[s]); m->first.push_back(f); m->count.push_back(c);
    }
    *out = m;
    return SLAM_OK;
}

int slam_multi_destroy(slam_multi* m) {
    if (!m) return SLAM_OK;
    for (size_t s = 0; s < m->dsend.size(); ++s) {
        hipSetDevice(m->dev[s]);
        if (m->dsend[s]) hipFree(m->dsend[s]);
        if (m->drecv[s]) hipFree(m->drecv[s]);
        if (m->cstream[s]) hipStreamDestroy(m->cstream[s]);
    }
    if (m->rccl) {
        nccl_comm_destroy_t destroy = (nccl_comm_destroy_t)dlsym(m->rccl, "ncclCommDestroy");
        for (void* c : m->comms) if (c && destroy) destroy(c);
        dlclose(m->rccl);
    }
    for (slam_handle* h : m->h) slam_destroy(h);
    delete m;
    return SLAM_OK;
}

int slam_multi_devices(const slam_multi* m) { return m ? (int)m->h.size() : 0; }
int64_t slam_multi_batch(const slam_multi* m) { return m ? m->B : 0; }
slam_handle* slam_multi_handle(slam_multi* m, int s) { return (m && s >= 0 && s < (int)m->h.size()) ? m->h[s] : nullptr; }
int slam_multi_shard(const slam_multi* m, int s, int64_t* first, int64_t* count) {
    if (!m || s < 0 || s >= (int)m->h.size() || !first || !count) return slam_internal_fail(SLAM_ERR_ARG, "bad shard");
    *first = m->first[s]; *count = m->count[s];
    return SLAM_OK;
}

int slam_multi_set_seed(slam_multi* m, uint64_t seed) { MULTI_ALL(slam_set_seed(hs, seed)); }
int slam_multi_set_vision(slam_multi* m, double r, double f0, double f1) { MULTI_ALL(slam_set_vision(hs, r, f0, f1)); }
int slam_multi_set_map(slam_multi* m, const double* map_xy, int L) { MULTI_ALL(slam_set_map(hs, map_xy, L)); }
int slam_multi_init(slam_multi* m, float x0, float y0, float yaw0) { MULTI_ALL(slam_init(hs, x0, y0, yaw0)); }
int slam_multi_step_sim(slam_multi* m, const float cmd[2]) { MULTI_ALL(slam_step_sim(hs, cmd)); }
int slam_multi_run_sim(slam_multi* m, const float* cmds, int T) { MULTI_ALL(slam_run_sim(hs, cmds, T)); }
int slam_multi_sync(slam_multi* m) { MULTI_ALL(slam_sync(hs)); }

int slam_multi_status(slam_multi* m, int32_t* flags) {
    if (!m || !flags) return slam_internal_fail(SLAM_ERR_ARG, "bad argument");
    for (size_t s = 0; s < m->h.size(); ++s) {
        const int rc = slam_status(m->h[s], flags + m->first[s]);
        if (rc) return rc;
    }
    return SLAM_OK;
}

int slam_multi_get_state(slam_multi* m, int64_t g, double* x, double* P, int32_t* M, int32_t* ids, int32_t* ts) {
    if (!m || g < 0 || g >= m->B) return slam_internal_fail(SLAM_ERR_ARG, "bad global instance");
    for (size_t s = 0; s < m->h.size(); ++s)
        if (g < m->first[s] + m->count[s]) return slam_get_state(m->h[s], (int)(g - m->first[s]), x, P, M, ids, ts);
    return slam_internal_fail(SLAM_ERR_ARG, "bad global instance");
}

int slam_multi_error_stats(slam_multi* m, double* out, int mode) {
    if (!m || !out) return slam_internal_fail(SLAM_ERR_ARG, "bad argument");
    const int n = (int)m->h.size();
    if (mode == 0) {   // host-side concatenation: what a single process needs
        for (int s = 0; s < n; ++s) {
            const int rc = slam_error_stats(m->h[s], out + m->first[s]);
            if (rc) return rc;
        }
        return SLAM_OK;
    }
    // ---- mode 1: device-to-device all-gather with RCCL (one communicator per device in this process) ----
    if (!m->rccl) {
        // transactional: the communicators, streams and buffers are built in locals and committed together with the library handle
        // only when every step succeeded; a failure tears down what exists, so a later call starts from scratch instead of indexing
        // half-filled vectors
        void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return slam_internal_fail(SLAM_ERR_UNSUPPORTED, "librccl.so cannot be loaded: %s", dlerror());
        nccl_comm_init_all_t init_all = (nccl_comm_init_all_t)dlsym(lib, "ncclCommInitAll");
        nccl_comm_destroy_t comm_destroy = (nccl_comm_destroy_t)dlsym(lib, "ncclCommDestroy");
        if (!init_all) { dlclose(lib); return slam_internal_fail(SLAM_ERR_UNSUPPORTED, "librccl.so has no ncclCommInitAll"); }
        std::vector<void*> comms(n, nullptr);
        if (init_all(comms.data(), n, m->dev.data()) != 0) { dlclose(lib); return slam_internal_fail(SLAM_ERR_HIP, "ncclCommInitAll failed"); }
        int64_t pad = 0;
        for (int s = 0; s < n; ++s) pad = m->count[s] > pad ? m->count[s] : pad;   // equal-sized contributions (ragged shards are padded)
        std::vector<hipStream_t> cstream(n, nullptr);
        std::vector<double*> dsend(n, nullptr), drecv(n, nullptr);
        bool ok = true;
        for (int s = 0; s < n && ok; ++s)
            ok = hipSetDevice(m->dev[s]) == hipSuccess && hipStreamCreateWithFlags(&cstream[s], hipStreamNonBlocking) == hipSuccess &&
                 hipMalloc(&dsend[s], sizeof(double) * pad) == hipSuccess && hipMalloc(&drecv[s], sizeof(double) * pad * n) == hipSuccess;
        if (!ok) {
            for (int s = 0; s < n; ++s) {
                hipSetDevice(m->dev[s]);
                if (dsend[s]) hipFree(dsend[s]);
                if (drecv[s]) hipFree(drecv[s]);
                if (cstream[s]) hipStreamDestroy(cstream[s]);
                if (comm_destroy && comms[s]) comm_destroy(comms[s]);
            }
            (void)hipGetLastError();
            dlclose(lib);
            return slam_internal_fail(SLAM_ERR_HIP, "allocating the gather buffers failed");
        }
        m->comms.swap(comms); m->cstream.swap(cstream); m->dsend.swap(dsend); m->drecv.swap(drecv);
        m->pad = pad;
        m->rccl = lib;
    }
    nccl_all_gather_t all_gather = (nccl_all_gather_t)dlsym(m->rccl, "ncclAllGather");
    nccl_group_t gstart = (nccl_group_t)dlsym(m->rccl, "ncclGroupStart"), gend = (nccl_group_t)dlsym(m->rccl, "ncclGroupEnd");
    if (!all_gather || !gstart || !gend) return slam_internal_fail(SLAM_ERR_UNSUPPORTED, "librccl.so lacks ncclAllGather / ncclGroupStart / ncclGroupEnd");
    for (int s = 0; s < n; ++s) {   // every shard forms its statistic in its own send buffer, on its device (no host staging: VERDICT r04)
        const int rc = slam_internal_error_stats_dev(m->h[s], m->dsend[s], (long long)m->pad);
        if (rc) return rc;
    }
    gstart();
    for (int s = 0; s < n; ++s) {
        hipSetDevice(m->dev[s]);
        if (all_gather(m->dsend[s], m->drecv[s], (size_t)m->pad, kNcclFloat64, m->comms[s], m->cstream[s]) != 0) { gend(); return slam_internal_fail(SLAM_ERR_HIP, "ncclAllGather failed"); }
    }
    if (gend() != 0) return slam_internal_fail(SLAM_ERR_HIP, "ncclGroupEnd failed");
    std::vector<double> all((size_t)m->pad * n);
    if (hipSetDevice(m->dev[0]) != hipSuccess || hipStreamSynchronize(m->cstream[0]) != hipSuccess ||
        hipMemcpy(all.data(), m->drecv[0], sizeof(double) * all.size(), hipMemcpyDeviceToHost) != hipSuccess)
        return slam_internal_fail(SLAM_ERR_HIP, "reading the gathered statistics failed");
    for (int s = 0; s < n; ++s) {
        hipSetDevice(m->dev[s]);
        hipStreamSynchronize(m->cstream[s]);
        memcpy(out + m->first[s], all.data() + (size_t)s * m->pad, sizeof(double) * m->count[s]);
    }
    return SLAM_OK;
}

}  // extern "C"
