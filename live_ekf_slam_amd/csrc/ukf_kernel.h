// ukf_kernel.h — parameter block and launchers of the UKF-SLAM step kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slam {

// One timestep of UKF::update (ukf.cpp:161-372) for every instance = two launches:
//   ukf_sqrt_kernel : nearestSPD + matrix square root of P_t (ukf.cpp:106-123,208)      -> sqtP
//   ukf_step_kernel : sigma points, motion model, weighted mean / covariance, landmark updates and insertions
//                     (ukf.cpp:214-372), optionally preceded by the measurement generator (sim_node.py:209-250)
struct UkfStepParams {
    // ---- filter state in HBM ----
    const double* P;    // [B][pstride]  P_t packed row-major n x n, n = 4+2*M[b]
    double* P_out;      // [B][pstride]  next P_t (ping-pong with P)
    double* x;          // [B][xstride]  x_t = [x, y, cos, sin, landmarks...]
    double* sqtP;       // [B][pstride]  scratch: matrix square root, n x n row-major (symmetric)
    int32_t* n_sq;      // [B]           dimension of the matrix currently held in sqtP
    double* Vt_store;   // [B][pstride]  V^T of the last eigen-decomposition (n_sq x n_sq, row p = eigenvector p): warm start
    int32_t* v_age;     // [B]           consecutive warm starts so far, -1 = no usable V
    double* x_prev;     // [B][xstride]  x_t the sigma points were drawn around (UKFState.X = [x, x + sqtP cols, x - sqtP cols])
    int32_t* M;         // [B]
    int32_t* ids;       // [B][L_max]
    int32_t* flags;     // [B]
    int32_t* timestep;  // [B]
    // ---- simulator state ----
    double* truth;      // [B][3]
    double* err_sum;    // [B]
    const double* map;  // [L][2]
    int32_t L;
    // ---- measurements ----
    const float* meas_in; const int32_t* meas_count_in; int32_t k_stride_in;
    float* meas_out; int32_t* meas_count_out; int32_t k_stride_out;
    float fwd, ang;
    // ---- filter config (filter.h:105-121) ----
    float v_d, v_th, w_r, w_b;
    double V00, V11, W00, W11;
    int32_t float_trig;  // unqualified cos/sin(float): float overload (1) or double function (0)
    int32_t acc_zest1, yaw_sigma;   // quirk switches ukf_accumulate_zest1 / ukf_sensing_yaw_from_sigma (include/slam_batch.h), 0 = reference
    // ---- simulator config ----
    double sV00, sV11, sW00, sW11, d_max, th_max, range_max, fov_min, fov_max;
    uint64_t seed;
    int64_t inst0;
    uint32_t step;
    int32_t B, L_max, pstride, xstride;
    int32_t b_off, b_cnt;  // the launch covers instances [b_off, b_off + b_cnt) (the batch is split over two streams)
    int32_t sim;
    int32_t loc;          // 1 = FilterChoice::UKF_LOC: every detection updates against the known map (ukf.cpp:146-154)
    unsigned long long* prof;   // optional [B][16] phase timers of the step kernel (debug), NULL otherwise
    const float* mapf;    // [L][3] float32 {id, x, y}: `filter->map` as it arrives on /truth/landmarks
    // Pass table of the fast sqrt kernel (<44, 256>): [nj / 4][pass < 21][thread 256] x 32 bytes, the LDS byte offsets of every operand a thread
    // touches in a pass of two Jacobi rounds at the padded state size nj = 4, 8, .., 44 (built once by launch_ukf_quad_table); the other
    // variants derive them from the schedule themselves
    const uint4* quad_tab;
    // workload statistics (optional): [0..7] instance-steps by detections in the message (7 = seven or more), [8] Jacobi sweeps
    // that rotated something, [9] eigen-decompositions (slam_k_histogram / slam_ukf_sweep_stats)
    unsigned long long* khist;
    // the size class beyond LDS (ukf_big_kernel.hip, n = 4 + 2 L_max > 104): [B][2 * pstride] doubles of scratch per instance (the scaled
    // symmetrised matrix and the warm-start product of the sqrt kernel, then P_pred of the step kernel); NULL for the LDS classes
    double* big_ws;
    // Messages longer than the size class holds (ukf_class_message_capacity).  0: none can occur.  1 (set by the host when one can, with
    // long_cap = that capacity and big_ws allocated): launch_ukf_step pairs the LDS step kernel, which then leaves every instance whose
    // message exceeds long_cap untouched, with ukf_big_step_kernel for exactly those (long_mode = 2 inside that launch); in SIM mode, where the
    // count is not known before the generator has run, the streamed kernel takes the whole launch.  The sqrt kernel is the class's own either way.
    int32_t long_mode, long_cap;
};
// detections ONE message may hold in the LDS size class of the step kernel (launch_ukf_step picks the class the same way)
inline int ukf_class_message_capacity(int L_max, bool loc, int L_map) { return ((loc && L_map > 20) ? 104 : 4 + 2 * L_max) <= 44 ? 20 : 50; }

static constexpr int kUkfRotThreads = 256;   // threads of the variant with the pass table
// Pass table for ukf_sqrt_kernel<44, 256> at the padded sizes nj = 4, 8, ..., 44: kUkfQuadTabEntries uint4 entries (2 MB), built on the host
static constexpr int kUkfQuadPasses = 21, kUkfQuadSizes = 12;   // block rounds T = 0 .. n / 2 - 2; n / 4 = 0 .. 11
static constexpr size_t kUkfQuadTabEntries = (size_t)kUkfQuadSizes * kUkfQuadPasses * kUkfRotThreads * 2;
hipError_t launch_ukf_quad_table(uint4* tab, hipStream_t stream);

static constexpr int kUkfLdsMaxLandmarks = 50;   // n = 4 + 2L <= 104: the fast kernels keep A, V^T and sqtP of an instance in LDS
static constexpr int kUkfMaxLandmarks = 200;     // beyond: ukf_big_kernel.hip, every n x n object in HBM / L2 (slow, bit-identical)
hipError_t launch_ukf_big_sqrt(const UkfStepParams& p, hipStream_t stream);
hipError_t launch_ukf_big_step(const UkfStepParams& p, hipStream_t stream);

hipError_t launch_ukf_sqrt(const UkfStepParams& p, hipStream_t stream);
hipError_t launch_ukf_step(const UkfStepParams& p, hipStream_t stream);

struct UkfInitParams {
    double* P; double* x; int32_t* n_sq; int32_t* v_age; int32_t* M; int32_t* flags; int32_t* timestep; double* truth; double* err_sum;
    int32_t B, pstride, xstride;
    double x0, y0, c0, s0;   // x_t = (x_0, y_0, cos(yaw_0), sin(yaw_0)) as the reference stores them (ukf.cpp:33)
    double tx, ty, tyaw;
};
hipError_t launch_ukf_init(const UkfInitParams& p, hipStream_t stream);

}  // namespace slam
