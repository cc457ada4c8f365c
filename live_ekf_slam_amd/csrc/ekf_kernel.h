// ekf_kernel.h — parameter block and launcher of the fused EKF-SLAM step kernel (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slam {

// One launch = one timestep of EKF::update (ekf.cpp:37-179) for every instance of the batch, optionally
// preceded (SIM mode) by the measurement generator get_cmd (sim_node.py:209-250) for the same instance.
struct EkfStepParams {
    // ---- filter state in HBM ----
    // P and x are stored as fp64 (SLAM_F64) or fp32 (SLAM_F32); strides are in ELEMENTS of that type
    const void* P;      // [B][pstride]  P_t: row-major n x n with leading dimension ekf_ld(n), n = 3+2*M[b]; updated IN PLACE
    void* P_out;        // [B][pstride]  second buffer: target of the steps that change the leading dimension (insertions)
    void* x;            // [B][xstride]  x_t
    double* scratch;    // [B][pstride] fp64, SLAM_F32 only: P between detection groups of one step (NULL for fp64)
    int32_t* M;         // [B]
    int32_t* ids;       // [B][L_max]    lm_IDs
    int32_t* flags;     // [B]           slam_instance_flags
    int32_t* timestep;  // [B]
    // ---- simulator state (SIM mode) ----
    double* truth;      // [B][3] true pose x_v (sim_node.py:222)
    double* err_sum;    // [B]   running sum of position errors (plotting_node.py:209-212)
    const double* map;  // [L][2]
    int32_t L;
    // ---- measurements ----
    const float* meas_in;         // EXT mode: [T][B][k_stride_in][3] (one timestep per launch: T = 1)
    const int32_t* meas_count_in; // EXT mode: [T][B]
    int32_t k_stride_in;
    float* meas_out;              // SIM mode, optional dump: [B][k_stride_out][3]
    int32_t* meas_count_out;      // SIM mode, optional: [B]
    int32_t k_stride_out;
    // ---- command (Command.msg) ----
    float fwd, ang;
    // multi-step launch: the kernel runs T consecutive timesteps per instance, step t with command
    // cmds[2t], cmds[2t+1] and RNG step index step+t; the result is always left in P.  cmds == NULL: T = 1 with (fwd, ang).
    const float* cmds;
    int32_t T;
    // ---- filter config after readCommonParams (filter.h:105-121), effective V / W ----
    float v_d, v_th, w_r, w_b;
    double V00, V11, W00, W11;
    int32_t id_known;
    float min_sep;
    // Messages longer than the size class holds (ekf_class_message_capacity).  0: none can occur (the LDS kernel alone).  Set by the host when
    // one can: launch_ekf_step then pairs two launches on external measurements - the LDS kernel with long_mode = 1 leaves every instance
    // whose message exceeds long_cap untouched, the HBM-streamed kernel with long_mode = 2 takes exactly those and walks their message where
    // it lies - and gives the whole launch to the streamed kernel in SIM mode, where the count is not known before the generator has run.
    int32_t long_mode, long_cap;
    int32_t abs_is_int, lm_from_pred;   // quirk switches ekf_abs_is_int / ekf_landmark_from_x_pred (include/slam_batch.h), 0 = reference
    // ---- simulator config (raw YAML values, used as half-widths: sim_node.py:216-217,247-248) ----
    double sV00, sV11, sW00, sW11, d_max, th_max, range_max, fov_min, fov_max;
    uint64_t seed;
    int64_t inst0;
    uint32_t step;  // RNG step index t (0-based)
    // ---- geometry ----
    int32_t B, L_max, pstride, xstride;
    int32_t sim;  // 1 = SIM mode, 0 = EXT mode
    unsigned long long* khist; // [16]; [0..7] instance-steps by detections in their message (k = 0..6, >= 7), accumulated over
                               // launches (one atomicAdd per bin and workgroup at the end of a launch); may be NULL
    unsigned long long* prof;  // optional [B][PROF_SLOTS] per-block stamps of the last launch (dbg & 4 | 32), NULL otherwise
    int32_t dbg;  // SLAM_DEBUG_FLAGS: 4 = phase cycle counters, 32 = wall-clock stamp + detection count per timestep of a
                  // multi-step launch; 1 / 2 / 16 (ablations, WRONG results) only in a -DSLAM_ABLATE build
};

// Leading dimension (in elements) of the row-major covariance of an instance with state size n: every row starts on a
// 16-byte boundary (2 doubles / 4 floats), so one lane's 16-byte vector never straddles two rows and the bulk stream
// can hand a lane the same column group in several consecutive rows (its (H P) operands are then read once per strip
// instead of once per element).  The pad elements (columns n .. ld-1) are kept at zero.
__host__ __device__ constexpr int ekf_ld(int n, int elem_bytes) {
    return elem_bytes == 8 ? ((n + 1) & ~1) : ((n + 3) & ~3);
}

static constexpr int kEkfProfSlots = 128;   // per-block slots of EkfStepParams::prof
static constexpr int kEkfTrafficSlot = 10;  // khist[10..13]: bytes moved by the P-stream passes (read + write), other global bytes
                                            // (thin gathers, vehicle rows / columns, state vectors), passes, updates applied by passes

// Largest landmark capacity of the LDS size classes of the fused kernel (n = 3+2L <= 403: 145 KB of the CU's 160 KB of LDS, one workgroup
// per CU; n <= 203: 72 KB, two per CU; the limit is LDS, not registers).  Beyond it ekf_big_kernel.hip takes over (round 4): the same
// EKF::update with the covariance streamed through HBM / L2 in every phase, one launch per timestep, fp64 storage only - slow, but the
// reference's state grows without a limit (ekf.cpp:144-146) and 200 landmarks was one.  Its own limit is the LDS for x, K and H P
// (6 n doubles) and the 2 x n^2 doubles of an instance in HBM.  fp32 storage: LDS classes up to 50 landmarks, the streamed kernel beyond (round 5: it reads and
// writes floats and runs the timestep in the handle's fp64 slab).
static constexpr int kEkfLdsMaxLandmarks = 200;
// detections ONE message may hold in the LDS size class a handle of capacity L_max runs (the class's landmark capacity; the surplus is
// not dropped any more).  The HBM-streamed kernel walks messages of any length: the instances with a longer message go to it
// (EkfStepParams::long_mode), in either storage type.
inline int ekf_class_message_capacity(int L_max) { return L_max <= 20 ? 20 : (L_max <= 50 ? 50 : (L_max <= 100 ? 100 : 200)); }
static constexpr int kEkfMaxLandmarks = 1000;
static constexpr int kEkfLdsMaxLandmarksF32 = 50;   // fp32 storage: the LDS size classes instantiated (20, 50); beyond them the streamed kernel, like fp64 beyond 200

// Tuning variants of the step kernel.  Every instantiation unit (ekf_inst.hip compiled with -DV_NMAX=.. -DV_W=.. -DV_KG=..
// -DV_UNR=.. -DV_F32=.. -DV_PIPE=.., see build.py) registers its launcher at load time; launch_ekf_step picks one by size
// class (smallest NMAX that fits), storage type and variant code = PIPE*1000 + W*100 + KG*10 + UNR:
//   W    wavefronts per filter instance (workgroup = 64*W threads)
//   KG   detections per group (updates applied in one pass over P)
//   UNR  rows per strip of the bulk stream = 16-byte vectors in flight per lane
//   PIPE 1 = the stream loads the next chunk while it updates and stores the current one
// The release library holds the defaults only; `SLAM_SWEEP=1 python -m live_ekf_slam_amd.build` adds the sweep set.
// what the runtime says about one instantiation (slam_kernel_info): name as the profiler prints it, static LDS per workgroup,
// registers, threads per workgroup, workgroups one CU holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor)
struct EkfKernelInfo {
    char name[96];
    int lds_bytes, vgprs, sgprs, threads, wg_per_cu;
};
struct EkfVariant {
    int nmax, code, f32;
    hipError_t (*launch)(const EkfStepParams&, hipStream_t);
    hipError_t (*info)(int multi, EkfKernelInfo*);
    EkfVariant* next;
};
void register_ekf_variant(EkfVariant* v);

// 1 if this build holds that variant for the size class of L_max (codes as above; 0 = the default, always present)
int ekf_variant_available(int L_max, int f32_storage, int variant);

// variant: 0 = the library's default for the size class and batch; otherwise a variant code (or just W).
hipError_t launch_ekf_step(const EkfStepParams& p, int variant, int f32_storage, hipStream_t stream);
// the instantiation launch_ekf_step would pick for (L_max, batch, variant, storage); multi = multi-step launch
hipError_t ekf_kernel_info(int L_max, int B, int variant, int f32_storage, int multi, EkfKernelInfo* out);

// the size class beyond the LDS classes (ekf_big_kernel.hip): p.T timesteps as p.T launches
hipError_t launch_ekf_big_step(const EkfStepParams& p, hipStream_t stream, int f32_storage = 0);   // f32: long messages of the fp32 LDS classes only
hipError_t ekf_big_kernel_info(EkfKernelInfo* out);

// sum over instances of 2*(n^2+n)*8 bytes into *out (device double, must be zeroed)
// per-instance average position error (err_sum / timestep; zeros in [B, pad)) into a device buffer
hipError_t launch_avg_error(const double* err_sum, const int32_t* timestep, int B, int pad, double* out, hipStream_t stream);
hipError_t launch_algorithmic_bytes(const int32_t* M, int B, int base, int elem_bytes, double* out, hipStream_t stream);

// fill x/P/M/... for Filter::init (ekf.cpp:4-21,29-34)
struct EkfInitParams {
    void* P; void* x; int32_t* M; int32_t* flags; int32_t* timestep; double* truth; double* err_sum;
    int32_t B, pstride, xstride, f32_storage;
    float x0, y0, yaw0;
    double tx, ty, tyaw;
};
hipError_t launch_ekf_init(const EkfInitParams& p, hipStream_t stream);

// device math self-test: out[0..n) = det_sincos/atan2/remainder/sqrt/div results for bit-exactness checks
hipError_t launch_math_probe(const double* a, const double* b, double* out, int n, hipStream_t stream);

}  // namespace slam
