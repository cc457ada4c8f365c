// Latency of the scalar building blocks of the thin phase on ONE wavefront (dependent chains, s_memtime):
// fp64 division, sqrt, det_sincos, det_atan2, remainder(x, 2pi), Philox noise pair, an LDS round trip.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I live_ekf_slam_amd/csrc tools/ubench_chain.hip -o tools/ubench_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "slam_math.h"
#include "slam_rng.h"
using namespace slam;
#define REP 64
__global__ void k(double* out, unsigned long long* cyc, double a, double b) {
    __shared__ double sh[64];
    double x = a + threadIdx.x * 1e-9, y = b;
    unsigned long long t0, t1;
    int i = 0;
#define TIME(idx, ...) t0 = __builtin_readcyclecounter(); _Pragma("unroll 1") for (int r = 0; r < REP; ++r) { __VA_ARGS__; } t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) cyc[idx] = (t1 - t0) / REP;
    TIME(0, x = x / y + 1.0)
    TIME(1, x = sqrt(x + 2.0))
    TIME(2, { double s, c; det_sincos(x, &s, &c); x = s + c + 3.0; })
    TIME(3, x = det_atan2(x, y) + 2.0)
    TIME(4, x = remainder(x + 7.5, kTwoPi) + 0.1)
    TIME(5, x = remainder(x + 300.0, kTwoPi) + 0.1)
    TIME(6, { double u0, u1; noise_pair(12345ull, (uint64_t)threadIdx.x, (uint32_t)(x * 1000.0), 3u, &u0, &u1); x = u0 + u1; })
    TIME(7, { sh[threadIdx.x] = x; __builtin_amdgcn_s_waitcnt(0xc07f); x = sh[(threadIdx.x + 1) & 63] + 1.0; })
    TIME(8, x = x * y + 1.0)
    TIME(9, x = (double)(float)x + 1.0)
    out[threadIdx.x] = x;
}
int main() {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.2345, 1.0000001);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[] = {"div+add", "sqrt", "det_sincos", "det_atan2", "remainder small", "remainder 300", "philox pair", "LDS round trip", "mul+add", "cvt f32 round trip"};
    for (int i = 0; i < 10; ++i) printf("%-20s %llu cycles\n", nm[i], h[i]);
    return 0;
}
