// calib_pingpong.hip — ceiling of the access pattern of the multi-step EKF kernel: every workgroup copies its own
// n*n fp64 matrix back and forth between two slabs T times (16 B/lane non-temporal loads and stores, 4 workgroups of
// 256 threads per CU by LDS padding).  Working set of the resident workgroups: 1024 x 2 x 85 KB = 174 MB, i.e. it
// fits the 256 MB Infinity Cache; the same bytes streamed once (T = 1) do not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double dbl2_t __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256, 4) void pingpong(double* A, double* B, int npair, size_t stride, int T, int lds_pad, int inplace) {
    extern __shared__ double pad[];
    if (lds_pad < 0) pad[threadIdx.x] = 0.0;
    dbl2_t* a = reinterpret_cast<dbl2_t*>(A + blockIdx.x * stride);
    dbl2_t* b = reinterpret_cast<dbl2_t*>(B + blockIdx.x * stride);
    for (int t = 0; t < T; ++t) {
        const dbl2_t* src = (t & 1) && !inplace ? b : a;
        dbl2_t* dst = ((t & 1) || inplace) ? a : b;
        for (int q0 = threadIdx.x; q0 < npair; q0 += 4 * 256) {
            dbl2_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * 256;
                if (q < npair) v[u] = NT ? __builtin_nontemporal_load(src + q) : src[q];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * 256;
                if (q < npair) {
                    v[u].x += 1.0;
                    if (NT) __builtin_nontemporal_store(v[u], dst + q); else dst[q] = v[u];
                }
            }
        }
        __syncthreads();
    }
}
int main(int argc, char** argv) {
    const int B = 65536, n = 103;
    const size_t stride = 10624;   // doubles per instance slab (pstride of the library)
    const int npair = (n * n + 1) / 2;
    double *A, *Bf;
    hipMalloc(&A, stride * 8 * B); hipMalloc(&Bf, stride * 8 * B);
    hipMemset(A, 0, stride * 8 * B); hipMemset(Bf, 0, stride * 8 * B);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes_step = 2.0 * n * n * 8 * B;
    for (int inplace = 0; inplace < 2; ++inplace)
    for (int nt = 0; nt < 2; ++nt)
        for (int lds : {36 * 1024, 64 * 1024}) {
            for (int T : {1, 10, 100}) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (nt) hipLaunchKernelGGL(pingpong<true>, dim3(B), dim3(256), lds, 0, A, Bf, npair, stride, T, lds, inplace);
                    else hipLaunchKernelGGL(pingpong<false>, dim3(B), dim3(256), lds, 0, A, Bf, npair, stride, T, lds, inplace);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                printf("inplace=%d nt=%d lds=%dK (%d WG/CU) T=%3d: %.3f ms/step  %.2f TB/s\n", inplace, nt, lds / 1024, 160 * 1024 / lds, T, best / T,
                       bytes_step / (best / T * 1e-3) / 1e12);
            }
        }
    return 0;
}
