// ubench_pack.cpp — host cost of packing one slam_step message (65 536 instances, stride 8 -> stride 4 detections): source
// arrays cold (40 distinct 6.3 MB messages, as a caller delivers them), destination in malloc'ed or pinned memory, 1 .. 8 threads
// created per call.  hipcc -O3 -pthread -o /tmp/ubp tools/ubench_pack.cpp && /tmp/ubp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
int main() {
    const size_t B = 65536, KS = 8, KQ = 4, NM = 40;
    std::vector<std::vector<float>> src(NM, std::vector<float>(B * KS * 3, 1.0f));
    float* dsts[3];
    dsts[0] = (float*)malloc(B * KQ * 3 * 16 * sizeof(float));
    memset(dsts[0], 0, B * KQ * 3 * 16 * sizeof(float));
    hipHostMalloc((void**)&dsts[1], B * KQ * 3 * 16 * sizeof(float), hipHostMallocNonCoherent);
    hipHostMalloc((void**)&dsts[2], B * KQ * 3 * 16 * sizeof(float), hipHostMallocDefault);
    const char* names[3] = {"malloc", "pinned non-coherent", "pinned default"};
    for (int dk = 0; dk < 3; ++dk)
        for (int nt : {1, 4, 8}) {
            double tot = 0;
            for (size_t rep = 0; rep < NM; ++rep) {
                float* d = dsts[dk] + (rep % 16) * B * KQ * 3;
                const float* s = src[rep].data();
                auto t0 = std::chrono::steady_clock::now();
                auto fn = [&](size_t b0, size_t b1) { for (size_t b = b0; b < b1; ++b) memcpy(d + b * KQ * 3, s + b * KS * 3, 36); };
                std::vector<std::thread> th;
                const size_t per = (B + nt - 1) / nt;
                for (int i = 1; i < nt; ++i) th.emplace_back(fn, i * per, (i + 1) * per < B ? (i + 1) * per : B);
                fn(0, per);
                for (auto& t : th) t.join();
                tot += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            }
            printf("%-20s threads %d: %.3f ms per message (cold source)\n", names[dk], nt, tot / NM);
        }
    return 0;
}
