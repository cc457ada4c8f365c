// sim_device.h — device-side range-bearing measurement generator, a port of get_cmd
// (reference ekf_ws/src/base_pkg/src/sim_node.py:209-250), shared by the EKF and UKF step kernels.
#pragma once
#include "slam_math.h"
#include "slam_rng.h"

namespace slam {

// Executed by ONE wavefront (lane = 0..63).  `P` is a step-parameter struct with the simulator fields
// (seed, inst0, sV00, sV11, sW00, sW11, d_max, th_max, range_max, fov_min, fov_max, map, L, truth).
// fwd, ang: the commanded motion; step: RNG step index.  tx, ty, tth: the instance's true pose (prefetched),
// advanced in place; lmx, lmy: prefetched map entry of id = lane.
// Writes the [id, range, bearing] float32 triplets of the visible landmarks (ascending id) to s_meas and returns
// their count (wave-uniform; triplets beyond KCAP are not stored, the caller caps and flags); lane 0 stores the new
// truth pose unless STORE_TRUTH is false (the EKF kernel writes it when it knows whether the instance froze).
template <int KCAP, bool STORE_TRUTH = true, class P>
__device__ __forceinline__ int sim_wave(const P& p, int b, int lane, float fwd, float ang, uint32_t step, double& tx,
                                        double& ty, double& tth, double lmx, double lmy, float* s_meas) {
    const uint64_t inst = (uint64_t)(p.inst0 + b);
    double u0, u1;
    noise_pair(p.seed, inst, step, 0u, &u0, &u1);
    double d = ((double)fwd + (2 * p.sV00) * u0) - p.sV00;          // sim_node.py:216
    double hdg = ((double)ang + (2 * p.sV11) * u1) - p.sV11;        // :217
    d = (p.d_max < d) ? p.d_max : d;                                 // min(d, d_max)        :219
    d = (0.0 < d) ? d : 0.0;                                         // max(0, .)
    hdg = (p.th_max < hdg) ? p.th_max : hdg;                         // :220
    hdg = (-p.th_max < hdg) ? hdg : -p.th_max;
    double s, c;
    det_sincos(tth, &s, &c);
    tx = tx + d * c;                                                 // :222 (yaw not wrapped)
    ty = ty + d * s;
    tth = tth + hdg;
    int count = 0;
#pragma unroll 1
    for (int base = 0; base < p.L; base += 64) {
        const int id = base + lane;
        bool vis = false;
        double r = 0.0, beta = 0.0;
        if (id < p.L) {
            if (base > 0) { lmx = p.map[2 * id]; lmy = p.map[2 * id + 1]; }   // ids 0..63 were prefetched
            const double dx = lmx - tx, dy = lmy - ty;
            r = sqrt(dx * dx + dy * dy);
            const double gb = det_atan2(dy, dx);
            beta = rem2pi(gb - tth);
            vis = !(r > p.range_max) && (beta > p.fov_min && beta < p.fov_max);
        }
        const unsigned long long mask = __ballot(vis);
        const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
        if (vis && pos < KCAP) {  // noise in visible-id order (sim_node.py:245-249), float32 wire format
            double v0, v1;
            noise_pair(p.seed, inst, step, (uint32_t)(1 + pos), &v0, &v1);
            const double rn = (r + (2 * p.sW00) * v0) - p.sW00;
            const double bn = (beta + (2 * p.sW11) * v1) - p.sW11;
            s_meas[3 * pos] = (float)id;
            s_meas[3 * pos + 1] = (float)rn;
            s_meas[3 * pos + 2] = (float)bn;
        }
        count += __popcll(mask);
    }
    if (STORE_TRUTH && lane == 0) {
        p.truth[3 * (size_t)b] = tx;
        p.truth[3 * (size_t)b + 1] = ty;
        p.truth[3 * (size_t)b + 2] = tth;
    }
    return count;   // the caller caps at KCAP and flags the overflow
}

}  // namespace slam
