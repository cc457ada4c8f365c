// ubench_hwid.hip — where do the wavefronts of a workgroup land?  Launches the EKF step kernel's shape (W wavefronts, LDS bytes per
// workgroup as given) and records HW_ID / XCC_ID of every wavefront of the first NB workgroups, plus entry / exit clocks.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_hwid.hip -o tools/ubench_hwid ; run: tools/ubench_hwid W LDS_BYTES [blocks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#include <algorithm>

__global__ void probe(unsigned* out, int nrec, int spin) {
    extern __shared__ double lds[];
    const int w = threadIdx.x >> 6;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = wall_clock64();
    double acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) { acc = acc * 1.0000001 + 0.5; lds[threadIdx.x] = acc; }
    if (acc == 12345.678) out[0] = 1;
    if ((threadIdx.x & 63) == 0 && (int)blockIdx.x < nrec) {
        unsigned* o = out + 4 * ((size_t)blockIdx.x * (blockDim.x >> 6) + w);
        o[0] = hw; o[1] = xcc; o[2] = (unsigned)t0; o[3] = (unsigned)wall_clock64();
    }
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 4, lds = argc > 2 ? atoi(argv[2]) : 40512, B = argc > 3 ? atoi(argv[3]) : 8192;
    const int nrec = B;
    unsigned* d;
    hipMalloc(&d, sizeof(unsigned) * 4 * (size_t)nrec * W);
    hipMemset(d, 0, sizeof(unsigned) * 4 * (size_t)nrec * W);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(probe, dim3(B), dim3(64 * W), lds, 0, d, nrec, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(4 * (size_t)nrec * W);
    hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    printf("W=%d lds=%d blocks=%d\n", W, lds, B);
    printf("first 24 workgroups: block: (xcc se sh cu | simd:waveslot per wavefront)\n");
    for (int b = 0; b < 24 && b < nrec; ++b) {
        printf("  b%-4d", b);
        for (int w = 0; w < W; ++w) {
            const unsigned hw = h[4 * ((size_t)b * W + w)], xcc = h[4 * ((size_t)b * W + w) + 1] & 0xf;
            if (w == 0) printf(" xcc%u se%u sh%u cu%-2u |", xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15);
            printf(" s%u:w%u", (hw >> 4) & 3, hw & 15);
        }
        printf("\n");
    }
    // statistics over the FIRST ROUND (workgroups that started before any finished would be ideal; take the first 1024 * 4 / W ... simply b < 256 * resident)
    std::map<unsigned, std::vector<int>> percu;   // cu key -> blocks
    long same_simd_pattern = 0, total = 0, ctrl_simd_hist[4] = {0, 0, 0, 0};
    for (int b = 0; b < nrec; ++b) {
        const unsigned hw0 = h[4 * ((size_t)b * W)], xcc = h[4 * ((size_t)b * W) + 1] & 0xf;
        const unsigned key = (xcc << 16) | (hw0 & 0xff00);
        percu[key].push_back(b);
        ctrl_simd_hist[(hw0 >> 4) & 3] += 1;
        bool ident = true;
        for (int w = 0; w < W; ++w) ident = ident && (((h[4 * ((size_t)b * W + w)] >> 4) & 3) == (unsigned)(w & 3));
        same_simd_pattern += ident; total += 1;
    }
    printf("CUs seen: %zu; workgroups whose wavefront w sits on SIMD w%%4: %ld of %ld\n", percu.size(), same_simd_pattern, total);
    printf("SIMD of wavefront 0 (the control wavefront): %ld %ld %ld %ld\n", ctrl_simd_hist[0], ctrl_simd_hist[1], ctrl_simd_hist[2], ctrl_simd_hist[3]);
    // one CU in detail: blocks in order of start time with the SIMD of their wavefront 0
    auto it = percu.begin();
    std::advance(it, percu.size() / 2);
    std::vector<int> bl = it->second;
    std::sort(bl.begin(), bl.end(), [&](int a, int c) { return h[4 * ((size_t)a * W) + 2] < h[4 * ((size_t)c * W) + 2]; });
    printf("one CU (key %x), its workgroups by start time: block start end simd/slot of each wavefront\n", it->first);
    for (size_t i = 0; i < bl.size() && i < 16; ++i) {
        const int b = bl[i];
        printf("  b%-5d %10u %10u ", b, h[4 * ((size_t)b * W) + 2], h[4 * ((size_t)b * W) + 3]);
        for (int w = 0; w < W; ++w) { const unsigned hw = h[4 * ((size_t)b * W + w)]; printf(" s%u:w%u", (hw >> 4) & 3, hw & 15); }
        printf("\n");
    }
    return 0;
}
