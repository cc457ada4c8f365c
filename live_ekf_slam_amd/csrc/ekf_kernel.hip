// ekf_kernel.hip — variant dispatch of the fused EKF-SLAM step kernel + small auxiliary kernels.
// The kernel template lives in ekf_kernel_impl.h; each (NMAX, W) variant is instantiated in its own
// translation unit (ekf_inst_*.hip) so they compile in parallel.
#include "ekf_kernel.h"

#include "../../include/slam_batch.h"
#include "slam_math.h"
#include "slam_rng.h"

// default variant codes (PIPE*1000 + W*100 + KG*10 + UNR); build.py instantiates exactly these in the release library
#ifndef SLAM_DEF_43
#define SLAM_DEF_43 1254
#endif
#ifndef SLAM_DEF_43_F32
#define SLAM_DEF_43_F32 1244
#endif
#ifndef SLAM_DEF_43_LARGE
#define SLAM_DEF_43_LARGE 1124
#endif
#ifndef SLAM_DEF_103
#define SLAM_DEF_103 1464   // round 3: SIX ring slots, passes at five pending updates - one slot stays free, so the control wavefront does not
                            // stall behind a pass, and a pass carries 4.75 updates instead of 3.83 (20 % fewer bytes per timestep).  The LDS
                            // for it (4 workgroups per CU = 40 960 B each): three landmark pairs of thin rows / cols instead of four (a
                            // timestep with more distinct detections runs as two groups inside the decoupled loop), the map copy out of
                            // LDS, a measurement ring of three.  Same-box A/B (profiles/r03h): 1444 57.0 / 71.1 M (20-step window / steady
                            // state), 1454 61.5 / 80.0, 1464 64.9 / 82.9, 1464 with passes at four 62.4 / 79.4
#endif
#ifndef SLAM_DEF_203
#define SLAM_DEF_203 1454
#endif
#ifndef SLAM_DEF_103_F32
#define SLAM_DEF_103_F32 1462   // fp32 storage: strips of two rows (0.88 vs 0.99 ms/step with four; fp64 prefers four: 0.92 vs 0.97).  Round 5: SIX
                                // ring slots with passes at four pending updates (1442 with passes at three until then; same-box table
                                // profiles/r05a/f32_variants.txt: 0.871 vs 0.894 ms/step on the bench window, 1.03 vs 1.24 at 2.4 detections per step)
#endif

namespace slam {

// ---- variant registry (filled by the static initialisers of the ekf_inst.hip instantiation units) ----
namespace {
EkfVariant* g_variants = nullptr;
const EkfVariant* find_variant(int nmax_class, int f32, int code) {
    for (const EkfVariant* v = g_variants; v; v = v->next)
        if (v->nmax == nmax_class && v->f32 == f32 && v->code == code) return v;
    return nullptr;
}
}  // namespace
void register_ekf_variant(EkfVariant* v) {
    v->next = g_variants;
    g_variants = v;
}

// Default variant code per size class (chosen by sweeps on the GPU, tools/gpu_sweep.py).  NMAX only sizes the LDS arrays
// (L_max <= 20 -> n <= 43, L_max <= 50 -> n <= 103, L_max <= 100 -> n <= 203, L_max <= 200 -> n <= 403).
static int default_code(int nmax_class, int f32, int B) {
    if (nmax_class == 43) {
        // two wavefronts per filter at every batch size: control + one streamer that also generates the measurements ahead
        // while no pass is due (L = 20: 267 M steps/s at batch 65 536, 200 M at 4096).  One wavefront per filter
        // (SLAM_DEF_43_LARGE, no decoupled loop) was the large-batch default until the streamer took the generator over:
        // 207 M at batch 65 536; it stays in the library as the lockstep-only variant the tests force.
        (void)B;
        return f32 ? SLAM_DEF_43_F32 : SLAM_DEF_43;   // (fp32 storage gains nothing from a fifth ring slot: its passes start at three pending updates)
    }
    if (nmax_class == 103 && f32) return SLAM_DEF_103_F32;
    if (nmax_class == 403) return 1444;   // 145 KB of LDS with four ring slots; one workgroup per CU either way
    if (nmax_class == 203) return SLAM_DEF_203;
    return SLAM_DEF_103;
}

static const EkfVariant* pick_variant(int L_max, int B, int variant, int f32_storage) {
    const int nmax = 3 + 2 * L_max;
    const int cls = nmax <= 43 ? 43 : (nmax <= 103 ? 103 : (nmax <= 203 ? 203 : (nmax <= 403 ? 403 : -1)));
    if (cls < 0) return nullptr;
    if (variant <= 0) return find_variant(cls, f32_storage, default_code(cls, f32_storage, B));
    const EkfVariant* v = find_variant(cls, f32_storage, variant);
    if (!v && variant < 10) {   // just W: any registered variant of that width
        for (const EkfVariant* q = g_variants; q && !v; q = q->next)
            if (q->nmax == cls && q->f32 == f32_storage && (q->code / 100) % 10 == variant) v = q;
    }
    return v;   // NULL: asked for a variant this build does not contain -> the caller fails loudly
}

int ekf_variant_available(int L_max, int f32_storage, int variant) {
    if (L_max > (f32_storage ? kEkfLdsMaxLandmarksF32 : kEkfLdsMaxLandmarks)) return variant <= 0 && L_max <= kEkfMaxLandmarks;   // the HBM-streamed class has one kernel
    return pick_variant(L_max, 1, variant, f32_storage) != nullptr;
}

hipError_t launch_ekf_step(const EkfStepParams& p, int variant, int f32_storage, hipStream_t stream) {
    if (p.L_max > (f32_storage ? kEkfLdsMaxLandmarksF32 : kEkfLdsMaxLandmarks)) return launch_ekf_big_step(p, stream, f32_storage);
    const EkfVariant* v = pick_variant(p.L_max, p.B, variant, f32_storage);
    if (!v) return hipErrorInvalidConfiguration;
    if (p.long_mode) {   // a message may exceed what the size class holds (ekf_kernel.h)
        if (p.sim || p.cmds != nullptr) return launch_ekf_big_step(p, stream, f32_storage);
        EkfStepParams q = p;
        q.long_mode = 1;                 // the LDS kernel: every instance whose message fits ...
        if (const hipError_t e = v->launch(q, stream); e != hipSuccess) return e;
        q.long_mode = 2;                 // ... and the streamed kernel: the others
        return launch_ekf_big_step(q, stream, f32_storage);
    }
    return v->launch(p, stream);
}

hipError_t ekf_kernel_info(int L_max, int B, int variant, int f32_storage, int multi, EkfKernelInfo* out) {
    if (L_max > (f32_storage ? kEkfLdsMaxLandmarksF32 : kEkfLdsMaxLandmarks)) return ekf_big_kernel_info(out);
    const EkfVariant* v = pick_variant(L_max, B, variant, f32_storage);
    if (!v) return hipErrorInvalidConfiguration;
    return v->info(multi, out);
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void alg_bytes_kernel(const int32_t* M, int B, int base, int elem_bytes, double* out) {
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += gridDim.x * blockDim.x) {
        const double n = (double)base + 2.0 * M[i];
        acc += 2.0 * (n * n + n) * (double)elem_bytes;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
hipError_t launch_algorithmic_bytes(const int32_t* M, int B, int base, int elem_bytes, double* out, hipStream_t stream) {
    hipLaunchKernelGGL(alg_bytes_kernel, dim3(64), dim3(256), 0, stream, M, B, base, elem_bytes, out);
    return hipGetLastError();
}

// compute_average_error (plotting_node.py:195-218) on the device: sum of the position errors / timesteps per instance, zeros in the
// padding of a ragged shard - what the RCCL gather of include/slam_multi.h sends, without a detour over the host
__global__ void avg_err_kernel(const double* err_sum, const int32_t* timestep, int B, int pad, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pad) return;
    double v = 0.0;
    if (i < B) { const int ts = timestep[i]; v = ts > 0 ? err_sum[i] / (double)ts : 0.0; }
    out[i] = v;
}
hipError_t launch_avg_error(const double* err_sum, const int32_t* timestep, int B, int pad, double* out, hipStream_t stream) {
    hipLaunchKernelGGL(avg_err_kernel, dim3((pad + 255) / 256), dim3(256), 0, stream, err_sum, timestep, B, pad, out);
    return hipGetLastError();
}

template <class ST>
__global__ void ekf_init_kernel(const EkfInitParams p) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    ST* P = static_cast<ST*>(p.P) + (size_t)b * p.pstride;
    ST* x = static_cast<ST*>(p.x) + (size_t)b * p.xstride;
    constexpr int ld = ekf_ld(3, (int)sizeof(ST));   // rows start on 16-byte boundaries (ekf_kernel.h)
    for (int i = 0; i < 3 * ld; ++i) P[i] = (ST)0;
    P[0] = (ST)(0.01 * 0.01); P[ld + 1] = (ST)(0.01 * 0.01); P[2 * ld + 2] = (ST)(0.005 * 0.005);   // ekf.cpp:11-14
    x[0] = (ST)p.x0; x[1] = (ST)p.y0; x[2] = (ST)p.yaw0;                             // ekf.cpp:31 (float arguments)
    p.M[b] = 0; p.flags[b] = 0; p.timestep[b] = 0;
    p.truth[3 * (size_t)b] = p.tx; p.truth[3 * (size_t)b + 1] = p.ty; p.truth[3 * (size_t)b + 2] = p.tyaw;
    p.err_sum[b] = 0.0;
}
hipError_t launch_ekf_init(const EkfInitParams& p, hipStream_t stream) {
    if (p.f32_storage) hipLaunchKernelGGL(ekf_init_kernel<float>, dim3((p.B + 255) / 256), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(ekf_init_kernel<double>, dim3((p.B + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// out layout per i: [sin, cos, atan2(a,b), remainder(a,2pi), sqrt(|a|), a/b, (double)(float)a, u53-noise]
__global__ void math_probe_kernel(const double* a, const double* b, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s, c;
    det_sincos(a[i], &s, &c);
    out[8 * (size_t)i + 0] = s;
    out[8 * (size_t)i + 1] = c;
    out[8 * (size_t)i + 2] = det_atan2(a[i], b[i]);
    out[8 * (size_t)i + 3] = rem2pi(a[i]);   // the device's short-cut for remainder(x, 2 pi); the host compares with libm
    out[8 * (size_t)i + 4] = sqrt(fabs(a[i]));
    out[8 * (size_t)i + 5] = a[i] / b[i];
    out[8 * (size_t)i + 6] = (double)(float)a[i];
    double u0, u1;
    noise_pair(12345ull, (uint64_t)i, 7u, 3u, &u0, &u1);
    out[8 * (size_t)i + 7] = u0 + u1;
}
hipError_t launch_math_probe(const double* a, const double* b, double* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a, b, out, n);
    return hipGetLastError();
}

}  // namespace slam
