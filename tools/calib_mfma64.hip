// calib_mfma64.hip - issue rate of v_mfma_f64_16x16x4_f64 on gfx950, alone and next to dependent fp64 VALU work on the same SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/calib_mfma64 tools/calib_mfma64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dbl4_t __attribute__((ext_vector_type(4)));
// mode 0: every wavefront runs NACC independent accumulators of MFMAs; mode 1: wavefront 0 runs a dependent v_fma_f64 chain
// instead (what the pose-graph producer does) and reports ITS time; mode 2: only the two wavefronts of one SIMD run MFMAs;
// mode 3: the chain alone; mode 4: the chain, its SIMD-mate idle, MFMAs on the other three SIMDs
template <int NACC>
__global__ __launch_bounds__(512) void k(double* out, unsigned long long* cyc, int iters, int mode) {
    const int w = threadIdx.x >> 6;
    dbl4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (dbl4_t){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6, x = a;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    if ((mode == 1 || mode == 3 || mode == 4) && w == 0) {
        for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) x = __builtin_fma(x, b, a);
        }
    } else if (mode == 3 || (mode == 4 && w == 4)) {
        // mode 3: the chain alone; mode 4: the chain with an idle SIMD-mate, MFMAs on the other three SIMDs only
    } else if (!(mode == 2 && w != 0 && w != 4)) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    double s = x;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 8 + w) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + w) * 2 + 1] = w1 - w0; }
}
int main() {
    const int nb = 256, iters = 20000;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * nb * 512); hipMalloc(&cyc, sizeof(unsigned long long) * nb * 16);
    unsigned long long h[nb * 16];
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<8>, dim3(nb), dim3(512), 0, 0, out, cyc, 100, mode);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<8>, dim3(nb), dim3(512), 0, 0, out, cyc, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const double nm = (double)iters * 8;
        printf("mode %d: %.3f ms; wavefront 0: %.1f shader cycles, %.1f ns (100 MHz clock) per %s; wavefront 1: %.1f ns per MFMA; ",
               mode, ms, (double)h[0] / ((mode == 1 || mode >= 3) ? nm * 4 : nm), (double)h[1] * 10.0 / ((mode == 1 || mode >= 3) ? nm * 4 : nm), (mode == 1 || mode >= 3) ? "dependent FMA" : "MFMA", (double)h[3] * 10.0 / nm);
        const double flops = (mode == 0 ? 8.0 : (mode == 1 ? 7.0 : (mode == 2 ? 2.0 : (mode == 3 ? 0.0 : 6.0)))) * nb * nm * 2048.0;
        printf("MFMA rate %.1f TFLOP/s\n", flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
