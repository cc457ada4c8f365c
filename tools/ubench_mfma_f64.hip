// ubench_mfma_f64.hip — which sequence of IEEE operations does v_mfma_f64_16x16x4_f64 perform?  Compares the instruction's
// result, bit for bit, with candidate CPU restatements (fma chains in ascending / descending k, unfused, pairwise).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/ubm tools/ubench_mfma_f64.hip && /tmp/ubm
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const double* A, const double* B, const double* C, double* D) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];   // A[row][k]
    const double b = B[(l / 16) * 16 + l % 16];  // B[k][col]
    d4 c;
    for (int i = 0; i < 4; ++i) c[i] = C[((l / 16) + 4 * i) * 16 + l % 16];
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[((l / 16) + 4 * i) * 16 + l % 16] = d[i];
}

static uint64_t rng = 88172645463325252ull;
static double rnd() {
    rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
    double u = (double)(rng >> 11) / 9007199254740992.0 - 0.5;
    int e = (int)((rng >> 3) % 20) - 10;
    return ldexp(u, e);
}

int main() {
    const int trials = 2000;
    long mism[6] = {0, 0, 0, 0, 0, 0};
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dD, 256 * 8);
    for (int t = 0; t < trials; ++t) {
        double A[64], B[64], C[256], D[256];
        for (int i = 0; i < 64; ++i) { A[i] = rnd(); B[i] = rnd(); }
        for (int i = 0; i < 256; ++i) C[i] = (t % 3 == 0) ? 0.0 : rnd();
        hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice);
        hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
        hipMemcpy(dC, C, sizeof C, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
        for (int r = 0; r < 16; ++r)
            for (int c = 0; c < 16; ++c) {
                const double* a = A + r * 4;
                double b[4];
                for (int k = 0; k < 4; ++k) b[k] = B[k * 16 + c];
                const double c0 = C[r * 16 + c];
                double cand[6];
                cand[0] = fma(a[3], b[3], fma(a[2], b[2], fma(a[1], b[1], fma(a[0], b[0], c0))));   // ascending k, fused
                cand[1] = fma(a[0], b[0], fma(a[1], b[1], fma(a[2], b[2], fma(a[3], b[3], c0))));   // descending k, fused
                {
                    volatile double s = c0;
                    for (int k = 0; k < 4; ++k) { volatile double p = a[k] * b[k]; s = s + p; }
                    cand[2] = s;                                                                       // ascending, unfused
                }
                cand[3] = c0 + fma(a[3], b[3], fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])));       // dot first, then + c
                cand[4] = fma(a[1], b[1], fma(a[0], b[0], c0)) ;                                      // (placeholder) first two only
                cand[4] = fma(a[3], b[3], fma(a[2], b[2], cand[4]));
                {   // pairwise fused: (a0b0 + a1b1) + (a2b2 + a3b3) + c
                    double p01 = fma(a[1], b[1], a[0] * b[0]), p23 = fma(a[3], b[3], a[2] * b[2]);
                    cand[5] = (p01 + p23) + c0;
                }
                const double d = D[r * 16 + c];
                for (int q = 0; q < 6; ++q)
                    if (memcmp(&d, &cand[q], 8) != 0) mism[q] += 1;
            }
    }
    const char* names[6] = {"fma chain ascending k", "fma chain descending k", "mul+add ascending k", "dot(fma) then + c",
                            "fma chain ascending (dup)", "pairwise"};
    for (int q = 0; q < 6; ++q) printf("%-28s mismatches %ld of %d\n", names[q], mism[q], trials * 256);
    return 0;
}
