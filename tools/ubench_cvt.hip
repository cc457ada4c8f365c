// micro-benchmark: issue rate of f64 <-> f32 conversions vs plain f64 adds on gfx950 (one wave per SIMD busy)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(double* out, int iters) {
    double a[8]; float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3 + i; f[i] = (float)a[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = a[i] + 1.25;                       // v_add_f64
            if (MODE == 1) { f[i] = (float)a[i]; a[i] = a[i] + 1.0; } // cvt_f32_f64 + add
            if (MODE == 2) { a[i] = (double)f[i] + a[i]; }            // cvt_f64_f32 + add
            if (MODE == 3) { a[i] = a[i] * 1.0000001; }               // v_mul_f64
            asm volatile("" : "+v"(a[i]), "+v"(f[i]));
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* d; hipMalloc(&d, 256 * 1024 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"add_f64", "cvt_f32_f64+add", "cvt_f64_f32+add", "mul_f64"};
    for (int m = 0; m < 4; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (m == 0) k<0><<<1024, 256>>>(d, iters); if (m == 1) k<1><<<1024, 256>>>(d, iters);
            if (m == 2) k<2><<<1024, 256>>>(d, iters); if (m == 3) k<3><<<1024, 256>>>(d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-18s %8.3f ms  -> %.2f cycles per wave-instruction-group(8) per SIMD\n", names[m], ms,
                            ms * 1e-3 * 2.4e9 / ((double)iters * (1024.0 * 4 / (256 * 4))));
        }
    }
    return 0;
}
