// calib_copy.hip — calibrates rocprofv3 FETCH_SIZE / WRITE_SIZE on known byte counts for the access patterns the
// step kernel uses: (a) 16 B/lane coalesced copy, (b) 8 B/lane coalesced copy, (c) strided 8-byte column gather.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy16(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void copy8(const double* __restrict__ a, double* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// each block gathers column `col` of its own 103x103 matrix (stride 103 doubles) and writes 103 doubles contiguously
__global__ void colgather(const double* __restrict__ a, double* __restrict__ out, int n, size_t stride, int col) {
    const double* m = a + blockIdx.x * stride;
    if (threadIdx.x < n) out[blockIdx.x * (size_t)128 + threadIdx.x] = m[(size_t)threadIdx.x * n + col];
}
int main() {
    const size_t bytes = (size_t)4 << 30;  // 4 GiB per buffer: far beyond L2 + MALL
    double *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipDeviceSynchronize();
    copy16<<<2048, 256>>>((const double2*)a, (double2*)b, bytes / 16);
    hipDeviceSynchronize();
    copy8<<<2048, 256>>>(a, b, bytes / 8);
    hipDeviceSynchronize();
    const int nb = 32768; const size_t stride = 10624;
    colgather<<<nb, 128>>>(a, b, 103, stride, 57);
    hipDeviceSynchronize();
    printf("copy16/copy8: read %zu write %zu bytes each; colgather: useful read %zu bytes (lines touched %zu x 64B = %zu), write %zu\n",
           bytes, bytes, (size_t)nb * 103 * 8, (size_t)nb * 103, (size_t)nb * 103 * 64, (size_t)nb * 103 * 8);
    return 0;
}
