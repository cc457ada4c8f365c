// slam_rng.h — counter-based RNG for the per-instance noise streams (shared by device code and the CPU oracle).
//
// The reference simulator draws uniform noise from ONE sequential Mersenne-Twister stream (sim_node.py:16,
// 216-217,247-248: `2*a*random() - a`).  A Monte-Carlo batch needs an independent, reproducible stream per
// instance that does not depend on batch size or GPU count, so the build keys a Philox4x32-10 generator as
//     key = (seed_lo, seed_hi)      counter = (step t, pair index p, instance_lo, instance_hi)
// Each call yields four 32-bit words = TWO uniforms in [0,1), built exactly like CPython's random():
//     u = ((a >> 5) * 2^26 + (b >> 6)) / 2^53     (53-bit, sim_node.py draws are Python floats).
// Pair p = 0 is the command-noise pair (d, hdg) of sim_node.py:216-217; pair p = 1+v is the (range, bearing)
// noise of the v-th VISIBLE landmark in ascending id order (sim_node.py:245-249) — the reference's draw order.
#pragma once
#include <stdint.h>
#include "slam_math.h"

namespace slam {

struct Philox4 { uint32_t v[4]; };

SLAM_HD void philox_mulhilo(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

SLAM_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 10; ++i) {
        uint32_t hi0, lo0, hi1, lo1;
        philox_mulhilo(M0, c0, &hi0, &lo0);
        philox_mulhilo(M1, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4 r;
    r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
    return r;
}

SLAM_HD double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// the two uniforms of pair `p` at step `t` for global instance `inst`
SLAM_HD void noise_pair(uint64_t seed, uint64_t inst, uint32_t t, uint32_t p, double* u0, double* u1) {
    Philox4 r = philox4x32_10(t, p, (uint32_t)inst, (uint32_t)(inst >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    *u0 = u53(r.v[0], r.v[1]);
    *u1 = u53(r.v[2], r.v[3]);
}

}  // namespace slam
