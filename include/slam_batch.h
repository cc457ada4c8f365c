/* slam_batch.h — C ABI of the MI355X batched EKF/UKF-SLAM predict–update engine (libslam_hip.so).
 *
 * This is the drop-in boundary for ONE path of kevin-robb/live_ekf_slam: `Filter::update()` of the EKF / UKF
 * subclasses plus the simulator's range-bearing generator.  Every entry point below names the reference
 * interface it replaces (paths relative to the reference repo root).  Plain pointers and sizes only; no
 * C++/torch types.  All functions return SLAM_OK (0) or a negative slam_status_code; slam_last_error() gives text.
 *
 * Threading: one caller thread per handle (the reference is single-threaded: localization_node.cpp:197).
 * Kernels are enqueued on one HIP stream per handle (slam_set_stream); slam_get_ and slam_sync synchronise.
 *
 * The batch: B independent filter instances that share the landmark map and the command sequence and differ
 * only in their noise streams (Monte-Carlo seeds).  Instance b of this handle has GLOBAL index
 * instance_offset + b (slam_set_instance_offset), which keys its counter-based RNG stream, so results do not
 * depend on how a batch is sharded over GPUs.
 */
#ifndef SLAM_BATCH_H
#define SLAM_BATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FilterChoice (ekf_ws/src/localization_pkg/include/localization_pkg/filter.h:44-51) */
enum slam_filter_kind { SLAM_EKF_SLAM = 1, SLAM_UKF_LOC = 2, SLAM_UKF_SLAM = 3 };

/* storage type of x and P in HBM (arithmetic is always fp64, as in the reference's Eigen MatrixXd) */
enum slam_dtype { SLAM_F64 = 0, SLAM_F32 = 1 };

enum slam_status_code {
    SLAM_OK = 0,
    SLAM_ERR_ARG = -1,          /* bad argument / bad handle                                                  */
    SLAM_ERR_HIP = -2,          /* a HIP runtime call failed (text in slam_last_error)                          */
    SLAM_ERR_UNSUPPORTED = -3,  /* configuration not supported by this build (e.g. L_max above the kernel limit) */
    SLAM_ERR_STATE = -4,        /* call order violated (step before init, sim step before set_map, ...)        */
    SLAM_ERR_IO = -5            /* config file unreadable / malformed                                          */
};

/* per-instance status bits returned by slam_status(); an instance with any FAIL bit set is frozen, mirroring the
 * reference where the corresponding condition throws and kills the node (filter.h:5 eigen_assert -> exception). */
enum slam_instance_flags {
    SLAM_INST_OK = 0,
    SLAM_INST_NONFINITE = 1,     /* x or P became non-finite                                                    */
    SLAM_INST_S_SINGULAR = 2,    /* zero pivot while inverting the 2x2 innovation covariance (ekf.cpp:135)      */
    SLAM_INST_INDEX_OOR = 4,     /* landmark index outside x_t (ekf.cpp:115 under the duplicate/unknown-id quirk): a message repeats a
                                    NEW id that it has itself just inserted (a repeat of an id that found no room is skipped again, a
                                    repeat of a mapped id is a second update - as the reference's loop does, detection by detection) */
    SLAM_INST_CAPACITY = 8,      /* a new landmark did not fit L_max (the reference grows without limit, ekf.cpp:144-146).  NOT raised by
                                    long messages (it was, until round 5): every handle walks a message of ANY length like ekf.cpp:73 /
                                    ukf.cpp:249-287, from every entry point - the instances whose message exceeds what the LDS size class
                                    holds (EKF 20 / 50 / 100 / 200, UKF 20 / 50 detections) take that timestep through the HBM-streamed
                                    kernel (slam_step knows the counts; slam_step_dev / slam_update_dev take the caller's k_stride as the
                                    bound, so a stride within the class's capacity keeps the launch on the fast kernel alone; the
                                    simulator when the map is larger than the capacity) - bit-identical, slower for those instances */
    SLAM_INST_SQRT_FAILED = 16,  /* UKF: eigen-iteration did not converge; stale sqtP reused (ukf.cpp:207-211)  */
    SLAM_INST_WATCHDOG = 32      /* EKF: a polling loop of the step kernel's intra-workgroup protocol exceeded its budget (~0.1 s);
                                    the instance is frozen with an undefined state instead of hanging the GPU.  A defect if it ever
                                    shows (round 3's random soak raised it once: tests/test_ring_protocol_model.py)   */
};

/* Flat mirror of the YAML keys the hot path reads (ekf_ws/src/base_pkg/config/params.yaml; key names kept).
 * Filter side: Filter::readCommonParams (filter.h:105-121).  Simulator side: get_cmd (sim_node.py:209-250). */
typedef struct slam_config {
    /* process_noise.mean.{v_d,v_th} (float in filter.h:84-85), process_noise.cov.{V_00,V_11} */
    float v_d, v_th;
    double V_00, V_11;
    /* sensing_noise.mean.{w_r,w_b} (float, filter.h:88-89), sensing_noise.cov.{W_00,W_11} */
    float w_r, w_b;
    double W_00, W_11;
    /* constraints.measurements */
    int landmark_id_is_known;
    float min_landmark_separation;
    /* constraints.commands / constraints.vision — used by the measurement generator only */
    double d_max, th_max;
    double range_max, fov_min, fov_max;
    /* init_pose (params.yaml:19-22) — the simulator's start pose = the filter's init pose */
    double init_x, init_y, init_yaw;
    /* Quirk switches.  Default 1 = behave exactly like the reference.
     * replicate_vw_quirk: filter.h:116-117 store W_00,W_11 into V and leave W = I2.                       */
    int replicate_vw_quirk;
    /* UKF only: resolve unqualified cos/sin on a float argument to the float overload (1) or to double (0);
     * see SURVEY.md Appendix B precision note. */
    int ukf_float_trig;
    int reserved[2];   /* (the test oracle's batch runner keeps two private switches here; the library ignores them) */
    /* Further quirk switches (SURVEY.md Appendix D asks for one per quirk).  0 = what this build takes the reference to
     * do (default); 1 = the alternative reading, so that a maintainer with the reference binary can localise a discrepancy.
     * ekf_abs_is_int (D-6, ekf.cpp:91-92): the unqualified `abs` of the unknown-id association resolves to ::abs(int) - the difference is
     *     truncated to an int first, so the box test passes for |dx| < 1 and |dy| < 1 instead of < min_landmark_separation;
     * ekf_landmark_from_x_pred (D-2, ekf.cpp:115-116): the landmark position of an update is read from x_pred (which earlier updates of
     *     the same step have moved) instead of x_t;
     * ukf_accumulate_zest1 (D-8, ukf.cpp:310-314): z_est(1), the predicted bearing, IS accumulated as the weighted mean of the sigma
     *     points' bearings (the reference leaves it 0: S, C and the innovation then use the raw wrapped bearings);
     * ukf_sensing_yaw_from_sigma (D-9, ukf.cpp:139): sensingModel takes the yaw from its sigma-point argument (rows 2, 3 of X_pred)
     *     instead of this->x_t. */
    int ekf_abs_is_int;
    int ekf_landmark_from_x_pred;
    int ukf_accumulate_zest1;
    int ukf_sensing_yaw_from_sigma;
} slam_config;

typedef struct slam_handle slam_handle;

/* ---- configuration ------------------------------------------------------------------------------------- */
/* Fill *cfg with the values committed in the reference's params.yaml (lines 25-52). */
int slam_config_default(slam_config* cfg);
/* Minimal `key: value` reader for a params.yaml-shaped file; replaces YAML::LoadFile + readCommonParams
 * (localization_node.cpp:29-30, filter.h:105-121).  Unknown keys are ignored, missing keys keep defaults. */
int slam_config_load(slam_config* cfg, const char* yaml_path);

/* ---- lifetime ------------------------------------------------------------------------------------------ */
/* Replaces the filter factory `std::make_unique<EKF|UKF>()` + `filter->readParams(config)`
 * (localization_node.cpp:33-47).  batch = number of instances on THIS device, L_max = landmark capacity
 * (EKF_SLAM: <= 1000 in fp64 [LDS size classes 20 / 50 / 100 / 200; beyond 200 the HBM-streamed class: the same EKF::update with
 * the covariance streamed through HBM in every phase, one launch per timestep, bit-identical but slow - the state of the reference
 * grows without a limit, ekf.cpp:144-146]; fp32 storage: LDS size classes 20 / 50, the streamed class beyond; UKF_SLAM: <= 200 [LDS size classes 20 / 50, beyond 50 its HBM-streamed class]; UKF_LOC: ignored, the
 * state holds no landmarks - its map may have any size and a message any length [beyond 50 detections - 20 while the map has <= 20
 * landmarks - the instance takes the HBM-streamed step kernel]).  The reference grows the state without limit (ekf.cpp:144-146); here the limit of the
 * fast classes is what one workgroup keeps in the 160 KB of LDS of a CU, and of the streamed class 2 x n x n doubles per instance in HBM.
 * dtype SLAM_F32 (fp32 storage of x and P, fp64 arithmetic) is available for EKF_SLAM.
 * UKF results depend on the capacity CLASS: the eigen-decomposition of nearestSPD / sqrt (ukf.cpp:106-123,208) is a cyclic
 * Jacobi iteration whose pair schedule is this build's choice (the reference calls Eigen) - the LDS classes pad the state size to a multiple
 * of four and walk the schedule over quadruples, the HBM-streamed class (L_max > 50) runs the circle method over the indices - so the same
 * measurement stream gives factors that differ in the last bits between an L_max = 50 and an L_max = 51 handle (each bit-identical to the
 * oracle configured with the same L_max, each within 1e-12 of LAPACK's eigh: tests/test_oracle_ukf.py, tests/test_jacobi_schedule.py). */
int slam_create(const slam_config* cfg, int filter_kind, int batch, int L_max, int dtype, int device,
                slam_handle** out);
int slam_destroy(slam_handle* h);
/* Use an existing HIP stream (hipStream_t passed as void*) instead of the handle's own. */
int slam_set_stream(slam_handle* h, void* hip_stream);
int slam_set_instance_offset(slam_handle* h, int64_t first_global_instance);
int slam_set_seed(slam_handle* h, uint64_t seed);
/* Change the simulated sensor limits at run time (constraints.vision.*; sim_node.py:238-243). */
int slam_set_vision(slam_handle* h, double range_max, double fov_min, double fov_max);

/* ---- Filter::init (ekf.cpp:29-34, ukf.cpp:31-45; called from initCallback localization_node.cpp:90-106) -- */
/* Same start pose for every instance (all seeds share the scenario).  Also resets truth pose, timestep,
 * M, status and the error accumulators. */
int slam_init(slam_handle* h, float x_0, float y_0, float yaw_0);

/* True landmark map [L][2] (x,y), id = row index.  Needed by slam_step_sim (sim_node.py:231-236) and by
 * UKF_LOC (`filter->map`, localization_node.cpp:152-156). Host pointer. */
int slam_set_map(slam_handle* h, const double* map_xy, int L);

/* ---- Filter::update (ekf.cpp:37-179, ukf.cpp:161-195; called from iterate localization_node.cpp:131) ------ */
/* One timestep for all instances.  cmd = {fwd, ang} (Command.msg:3-5), shared by the batch.
 * meas: [batch][k_stride][3] float32 {id, range, bearing} (Float32MultiArray layout of sim_node.py:245-249,
 * padded to k_stride detections per instance); meas_count: [batch].  HOST pointers: the message is copied before the call
 * returns (mirroring `lm_meas = lmMeasMsg->data`, ekf.cpp:64), so the caller may reuse its buffers at once.  EKF: like
 * slam_step_sim, consecutive calls are queued (up to slam_set_lazy_steps timesteps, messages of at most 4 detections per
 * instance) and run as one multi-step launch; every other entry point runs what is queued first.
 * QUEUED MEANS NOT ENQUEUED: until the queue runs (it is full, another slam_* entry point is called, or slam_sync), nothing of
 * a queued step is on the handle's stream, so an event recorded or a stream synchronised right after the call sees no work
 * of that step, and a HIP error of a queued step is reported by the call that runs the queue (the queued timesteps are
 * kept, not dropped, when that launch fails).  slam_set_lazy_steps(h, 0) restores one launch per call. */
int slam_step(slam_handle* h, const float cmd[2], const float* meas, const int32_t* meas_count, int k_stride);
/* Same, with DEVICE pointers (no host copy).  The buffers are read in the order of the handle's stream: they may be
 * overwritten by work enqueued on that stream after the call returns.  By default every call enqueues its step on the stream
 * at once (caller-owned buffers on a caller-owned stream: events and stream synchronisation around the call see the step).
 * Queueing is OPT-IN for this entry point: after slam_set_lazy_steps(h, n > 1) (or with SLAM_LAZY_STEPS set) the message is
 * copied device-to-device into a queue of n entries, which run as one multi-step launch; any other entry point runs the queue
 * first, with the caveat stated at slam_step. */
int slam_step_dev(slam_handle* h, const float cmd[2], const float* d_meas, const int32_t* d_meas_count,
                  int k_stride);
/* One timestep where the device-side generator (a port of get_cmd, sim_node.py:209-250) advances each
 * instance's true pose with its own noise stream and produces its measurements, then the filter consumes
 * them in the same kernel.  Also accumulates the position error of plotting_node.py:209-212. */
int slam_step_sim(slam_handle* h, const float cmd[2]);
/* EKF: consecutive slam_step_sim / slam_step calls are queued on the host and run as one multi-step launch of up to n
 * timesteps (default 32, environment variable SLAM_LAZY_STEPS; 0 = one launch per call); every other entry point runs what is
 * queued first, so the results seen through the API do not change (a multi-step launch gives the same bits as single steps),
 * only the speed — and the moment the work reaches the stream (see slam_step).  Calling it also switches the queue of
 * slam_step_dev on (n > 1) or off. */
int slam_set_lazy_steps(slam_handle* h, int n);
/* Timesteps the per-tick entry points have accepted but not yet enqueued on the handle's stream (0 = everything the caller
 * asked for is on the stream: an event recorded now covers it). */
int slam_queued_steps(const slam_handle* h);
/* T consecutive slam_step_sim calls; cmds = [T][2] float32 host array (precomputed trajectory,
 * sim_node.py:142-152). */
int slam_run_sim(slam_handle* h, const float* cmds, int T);
/* Timesteps one kernel launch of slam_run_sim carries (EKF): 0 = the whole call (default; also the environment variable
 * SLAM_RUN_CHUNK at slam_create), 1 = one launch per timestep.  Results do not depend on it. */
int slam_set_run_chunk(slam_handle* h, int steps_per_launch);
/* UKF only: the two public halves of UKF::update, predictionStage(cmd) and updateStage(meas) (filter.h:187-188,
 * ukf.cpp:197-291).  x_t / P_t change when the update stage finishes, exactly as in the reference; the pair gives
 * bit-identical results to slam_step_dev.  d_meas / d_meas_count are DEVICE pointers (k_stride 0 = empty message). */
int slam_predict(slam_handle* h, const float cmd[2]);
int slam_update_dev(slam_handle* h, const float* d_meas, const int32_t* d_meas_count, int k_stride);

/* ---- state export: getStateVector / publishState payload (ekf.cpp:181-220, ukf.cpp:47-104) ---------------- */
/* Sizes: x needs n_max doubles, P needs n_max*n_max doubles, ids needs L_max ints, where
 * n_max = 3+2*L_max (EKF) or 4+2*L_max (UKF).  On return *M landmarks are valid, n = 3|4 + 2*M, and P holds the
 * n x n matrix ROW-MAJOR with leading dimension n (EKFState.msg:12-13 order).  Any pointer may be NULL. */
int slam_get_state(slam_handle* h, int instance, double* x, double* P, int32_t* M, int32_t* ids,
                   int32_t* timestep);
/* UKF only — UKFState.X (ukf.cpp:92-101, UKFState.msg): the sigma points of the last prediction stage, COLUMN-major
 * rows x cols = n x (2n+1) with n = 4 + 2*M at the start of that step; X needs n_max*(2*n_max+1) doubles (may be NULL). */
int slam_get_sigma_points(slam_handle* h, int instance, double* X, int32_t* rows, int32_t* cols);
/* publishState EVERY tick at batch speed (the reference's loop is update -> publishState, localization_node.cpp:131-139).
 * Every getter runs the queued timesteps of the whole batch first, so reading one instance's state after every step call
 * would force one launch per tick.  After slam_track_instance(h, b) the library runs instance b ALSO in a one-instance shadow
 * filter - same config, seed, map and global instance id, therefore the same bits (results do not depend on how a batch is
 * partitioned) - stepped at once at every slam_step / slam_step_sim / slam_step_dev / slam_run_sim call on its own stream, and
 * slam_get_state(h, b, ...) answers from the shadow while the batch's steps stay queued.  The instance's current state is
 * copied into the shadow, so tracking may start at any time; instance = -1 switches it off.  EKF handles only (the UKF kinds
 * launch every step at the call; nothing is queued there).  (slam_step_dev: the shadow's
 * step waits for what is already enqueued on the handle's stream, since it reads the caller's device buffers.) */
int slam_track_instance(slam_handle* h, int instance);
/* Vehicle pose estimate (x, y, yaw) of every instance: [batch][3] (EKFState x_v,y_v,yaw_v). */
int slam_get_poses(slam_handle* h, double* poses);
int slam_get_landmark_counts(slam_handle* h, int32_t* M);            /* [batch] */
int slam_get_truth(slam_handle* h, double* truth_poses);             /* [batch][3], sim_node.py x_v */
/* Measurements the generator produced in the last slam_step_sim: meas [batch][k_stride][3], count [batch].  The generator only
 * keeps a copy of its messages once somebody has asked: the FIRST call (or one with a larger k_stride) switches the dump on and
 * returns count = 0 for every instance; call it once before the steps whose messages are wanted. */
int slam_get_last_meas(slam_handle* h, float* meas, int32_t* meas_count, int k_stride);
/* compute_average_error (plotting_node.py:195-218): mean over steps so far of the Euclidean position error of
 * the estimate at timestep t against the true pose of step t, per instance: [batch]. */
int slam_error_stats(slam_handle* h, double* per_instance_avg_err);
int slam_status(slam_handle* h, int32_t* per_instance_flags);       /* [batch] slam_instance_flags */

/* ---- scenario generators (host side; sim_node.py:63-206) ------------------------------------------------------ */
/* generate_landmarks + generate_full_trajectory of the reference simulator for one scenario seed (the reference seeds
 * CPython's Mersenne Twister; the same generator is implemented in include/slam_scenario.hpp, so a seed gives the reference's
 * map and command sequence bit for bit).  map_type: "random" | "grid" | "demo" | "igvc1" (the two fixed maps are read from
 * fixed_maps_json = live_ekf_slam_amd/data/fixed_maps.json; num_landmarks is ignored for them and for "grid").
 * Out: map_xy [*num_landmarks_out][2] (may be NULL to query the count; capacity in landmarks), cmds [num_iterations][2] float32
 * (fwd, ang) = the Command messages of sim_node.py:142-152.  Runs on the host; no GPU needed. */
int slam_scenario_make(const char* map_type, const char* fixed_maps_json, uint64_t seed, int num_landmarks, int num_iterations,
                       double* map_xy, int map_capacity, int32_t* num_landmarks_out, float* cmds);

/* ---- checkpoint / resume --------------------------------------------------------------------------------- */
/* The reference keeps the filter only in memory (a killed node loses the run).  slam_save_state writes everything a handle
 * needs to continue - x, P, landmark counts and ids, status, timestep, the simulator's true pose, the error accumulators,
 * the RNG step / seed / instance offset (UKF: also the square root of the last prediction stage and the warm-start
 * eigenvectors) - to one binary file (raw slabs, 64 MB at a time through the host); slam_load_state restores it into a handle
 * created with the same kind, batch, L_max and dtype (the map, the config and the sensor limits are the caller's to set
 * again).  A run continued from a loaded state is bit-identical to the uninterrupted one. */
int slam_save_state(slam_handle* h, const char* path);
int slam_load_state(slam_handle* h, const char* path);

/* ---- misc ------------------------------------------------------------------------------------------------ */
int slam_sync(slam_handle* h);
int slam_batch(const slam_handle* h);
int slam_state_dim_max(const slam_handle* h);
/* Algorithmic HBM bytes of the last step summed over instances: sum_b 2*(n_b^2+n_b)*sizeof(storage)
 * (SURVEY.md §8d).  Computed on the device from the per-instance M; synchronises. */
int slam_algorithmic_bytes(slam_handle* h, double* bytes);
/* Workload statistics: instance-steps by the number of detections in their message (out[k] for k = 0..6, out[7] for
 * k >= 7), accumulated by the EKF / UKF step kernels since slam_create or the last reset (ekf.cpp:65 `num_landmarks`).  The cost
 * of EKF::update grows with k (one rank-2 downdate of P per detection, ekf.cpp:140), so a throughput figure is only
 * meaningful together with this histogram.  Synchronises. */
int slam_k_histogram(slam_handle* h, uint64_t out[8], int reset);
/* UKF handles: out[0] = Jacobi sweeps that rotated at least one pair, out[1] = eigen-decompositions (instance-steps), summed
 * by the square-root kernel since slam_create or the last reset: out[0] / out[1] is the mean number of sweeps the
 * `nearestSPD` + `.sqrt()` of ukf.cpp:106-123,208 took, which sets the arithmetic of a UKF step.  Synchronises. */
int slam_ukf_sweep_stats(slam_handle* h, uint64_t out[2], int reset);
/* EKF handles: what the step kernels moved through global memory, counted ON THE DEVICE (one accumulation per pass and
 * workgroup, summed per launch like the k histogram) since slam_create or the last reset: out[0] = bytes the passes of the
 * P stream read + wrote (`P -= K (H P)`, ekf.cpp:140, applied for a group of deferred updates at once), out[1] = every other
 * global byte (thin row / column gathers, the vehicle rows / columns the prediction changes, x / ids / scalars at launch
 * start and end), out[2] = passes, out[3] = rank-2 updates those passes applied.  bench.py's roofline.traffic is
 * out[0] + out[1] of the timed launches; profiles/r03* hold the rocprofv3 PMC cross-check.  Synchronises. */
int slam_traffic_counters(slam_handle* h, uint64_t out[4], int reset);
/* Zero the counters of slam_k_histogram and slam_traffic_counters in stream order (no host synchronisation): for measurements that
 * must not leave the device idle between their warm-up and their timed launches (bench.py). */
int slam_reset_counters_async(slam_handle* h);
/* The step-kernel instantiation this handle launches (multi_step != 0: the multi-step launch of slam_run_sim / the queues,
 * else the one-step launch): name as rocprofv3 prints it (e.g. "ekf_step_kernel<103,4,4,4,double,1,true>"), and
 * out = {static LDS bytes per workgroup, VGPRs, threads per workgroup, workgroups one CU holds at once, CUs of the device}.
 * Needs a HIP device (the runtime is asked, nothing is hard-coded). */
int slam_kernel_info(slam_handle* h, int multi_step, char* name, int name_cap, int32_t out[5]);
/* Diagnostics: flag 32 = every workgroup of the EKF multi-step kernel stamps the wall clock and its detection count per
 * timestep (read back by the profiling tools / bench.py's per-k table), flag 4 = per-phase cycle counters; 0 = off (default;
 * the environment variable SLAM_DEBUG_FLAGS sets the initial value). */
int slam_set_debug_flags(slam_handle* h, int flags);
/* Build introspection: 1 if the library holds the EKF step-kernel tuning variant `variant` (code PIPE*1000 + W*100 + KG*10 +
 * UNR, selected per handle by the environment variable SLAM_WAVES_PER_FILTER) for landmark capacity L_max and storage
 * dtype; 0 = the default, always present.  Release builds hold the defaults only; asking a handle for a variant the build
 * lacks makes its steps fail with SLAM_ERR_HIP. */
int slam_variant_available(int L_max, int dtype, int variant);
/* Diagnostics: evaluate the device's elementary functions on host arrays a[n], b[n]; out[8*n] =
 * {sin a, cos a, atan2(a,b), remainder(a,2pi), sqrt|a|, a/b, (double)(float)a, u53 noise} per element.
 * Used by the parity tests to prove the device math is bit-identical to the host's. */
int slam_math_probe(const double* a, const double* b, double* out, int n, int device);
const char* slam_last_error(void);
const char* slam_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_BATCH_H */
