// ekf_kernel.hip — variant dispatch of the fused EKF-SLAM step kernel + small auxiliary kernels.
// The kernel template lives in ekf_kernel_impl.h; each (NMAX, W) variant is instantiated in its own
// translation unit (ekf_inst_*.hip) so they compile in parallel.
#include "ekf_kernel.h"

#include "../../include/slam_batch.h"
#include "slam_math.h"
#include "slam_rng.h"

namespace slam {

template <int NMAX, int W, int KG, int UNR, class ST>
hipError_t launch_variant(const EkfStepParams& p, hipStream_t stream);
extern template hipError_t launch_variant<103, 4, 4, 4, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<103, 4, 3, 4, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<103, 2, 3, 8, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<103, 2, 4, 8, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<103, 8, 4, 2, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<103, 4, 4, 8, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<103, 4, 2, 4, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<43, 2, 4, 4, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<43, 4, 4, 4, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<43, 1, 4, 8, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<43, 2, 4, 8, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<43, 2, 2, 4, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<43, 1, 2, 4, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<103, 4, 4, 4, float>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<203, 4, 4, 4, double>(const EkfStepParams&, hipStream_t);
extern template hipError_t launch_variant<43, 2, 4, 4, float>(const EkfStepParams&, hipStream_t);

// NMAX only sizes the LDS arrays (L_max <= 20 -> n <= 43, L_max <= 50 -> n <= 103).  `wpf` selects a tuning
// variant: 0 = default, W (wavefronts per filter) or the 3-digit code W*100 + KG*10 + UNR.
hipError_t launch_ekf_step(const EkfStepParams& p, int wpf, int f32_storage, hipStream_t stream) {
    const int nmax = 3 + 2 * p.L_max;
    if (f32_storage) {  // fp32 storage of x and P (BASELINE configs[3]); one tuning variant per size class
        if (nmax <= 43) return launch_variant<43, 2, 4, 4, float>(p, stream);
        if (nmax <= 103) return launch_variant<103, 4, 4, 4, float>(p, stream);
        return hipErrorInvalidValue;
    }
    if (nmax <= 43) {
        switch (wpf) {
            case 244: return launch_variant<43, 2, 4, 4, double>(p, stream);
            case 444: return launch_variant<43, 4, 4, 4, double>(p, stream);
            case 148: return launch_variant<43, 1, 4, 8, double>(p, stream);
            case 248: return launch_variant<43, 2, 4, 8, double>(p, stream);
            case 224: return launch_variant<43, 2, 2, 4, double>(p, stream);
            case 124: return launch_variant<43, 1, 2, 4, double>(p, stream);
            case 4: return launch_variant<43, 4, 4, 4, double>(p, stream);
            case 1: return launch_variant<43, 1, 4, 8, double>(p, stream);
            // large batches: one wavefront per filter and groups of 2 detections (smallest footprint, most filters
            // resident per CU) win by ~10 %; small batches are latency-bound and prefer two wavefronts per filter
            default: return p.B >= 16384 ? launch_variant<43, 1, 2, 4, double>(p, stream) : launch_variant<43, 2, 4, 4, double>(p, stream);
        }
    }
    if (nmax <= 103) {
        switch (wpf) {
            case 444: return launch_variant<103, 4, 4, 4, double>(p, stream);
            case 434: return launch_variant<103, 4, 3, 4, double>(p, stream);
            case 238: return launch_variant<103, 2, 3, 8, double>(p, stream);
            case 248: return launch_variant<103, 2, 4, 8, double>(p, stream);
            case 842: return launch_variant<103, 8, 4, 2, double>(p, stream);
            case 448: return launch_variant<103, 4, 4, 8, double>(p, stream);
            case 424: return launch_variant<103, 4, 2, 4, double>(p, stream);
            case 8: return launch_variant<103, 8, 4, 2, double>(p, stream);
            case 2: return launch_variant<103, 2, 4, 8, double>(p, stream);
            default: return launch_variant<103, 4, 4, 4, double>(p, stream);
        }
    }
    if (nmax <= 203) return launch_variant<203, 4, 4, 4, double>(p, stream);   // L_max <= 100: only the LDS arrays grow
    return hipErrorInvalidValue;
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

template <class ST>
__global__ void ekf_init_kernel(const EkfInitParams p) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    ST* P = static_cast<ST*>(p.P) + (size_t)b * p.pstride;
    ST* x = static_cast<ST*>(p.x) + (size_t)b * p.xstride;
    for (int i = 0; i < 9; ++i) P[i] = (ST)0;
    P[0] = (ST)(0.01 * 0.01); P[4] = (ST)(0.01 * 0.01); P[8] = (ST)(0.005 * 0.005);   // ekf.cpp:11-14
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
    out[8 * (size_t)i + 3] = remainder(a[i], kTwoPi);
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
