/* slam_pgs.h — C ABI of the MI355X batched pose-graph SLAM solver (part of libslam_hip.so).
 *
 * Drop-in boundary for the reference's `PoseGraph : Filter` with `implementation: gtsam`
 * (ekf_ws/src/localization_pkg/src/pose_graph.cpp, class declaration filter.h:232-322): the graph-building calls of
 * every timestep and `solvePoseGraph()` (gtsam::LevenbergMarquardtOptimizer, default parameters), for a batch of B
 * independent graphs — Monte-Carlo instances that share the command sequence (BetweenFactors) and differ in their
 * measurements and in the secondary filter's estimates.  Same conventions as slam_batch.h: plain pointers and sizes,
 * int status codes (slam_status_code), text in slam_last_error(), one caller thread per handle, kernels on one stream.
 *
 * Not covered: `implementation: sesync | custom` (the reference itself throws for both, pose_graph.cpp:34-38),
 * unknown landmark ids (the reference throws, pose_graph.cpp:137), `update_landmarks_after_adding` (false in
 * params.yaml:63 and forced false when solving every iteration, pose_graph.cpp:45-48).
 */
#ifndef SLAM_PGS_H
#define SLAM_PGS_H

#include <stdint.h>

#include "slam_batch.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pgs_handle pgs_handle;

/* per-instance status bits (pgs_get_stats) */
enum pgs_instance_flags {
    PGS_INST_OK = 0,
    PGS_INST_POSE_CAP = 1,        /* more timesteps than N_max: the surplus updates were dropped                 */
    PGS_INST_LANDMARK_CAP = 2,    /* more distinct landmark ids than L_max: the surplus landmarks were dropped   */
    PGS_INST_MEAS_CAP = 4,        /* more detections in one message than k_per_pose: the surplus were dropped    */
    PGS_INST_NOT_CONVERGED = 8,   /* LM hit maxIterations (100) or the trial cap                                 */
    PGS_INST_NONFINITE = 16
};

/* Replaces `std::make_unique<PoseGraph>()` + readParams (localization_node.cpp:45-47, pose_graph.cpp:12-66): noise
 * models Diagonal::Sigmas(V00, V00, V11) for the BetweenFactors and Sigmas(W11, W00) for the BearingRangeFactors
 * (pose_graph.cpp:52-54) from the EFFECTIVE V / W of Filter::readCommonParams (filter.h:105-121, incl. its quirk).
 * N_max = pose capacity (num_iterations), L_max <= 255 landmarks, k_per_pose = detections stored per timestep. */
int pgs_create(const slam_config* cfg, int batch, int N_max, int L_max, int k_per_pose, int device, pgs_handle** out);
int pgs_destroy(pgs_handle* h);
int pgs_set_stream(pgs_handle* h, void* hip_stream);
int pgs_set_instance_offset(pgs_handle* h, int64_t first_global_instance);
int pgs_set_seed(pgs_handle* h, uint64_t seed);
int pgs_set_map(pgs_handle* h, const double* map_xy, int L);      /* simulator only (pgs_run_sim) */

/* PoseGraph::init (pose_graph.cpp:68-95): first pose node + PriorFactor with sigmas (1.3, 1.3, 1.2). */
int pgs_init(pgs_handle* h, float x_0, float y_0, float yaw_0);

/* One iterate() of localization_node.cpp:108-140 as seen by the pose graph:
 *   updateNaiveVehPoseEstimate(sec_pose)   (pose_graph.cpp:97-119; sec_pose [batch][3] = the secondary filter's
 *                                           x, y, yaw per instance; NULL keeps the previous estimate)
 *   update(cmd, meas)                      (pose_graph.cpp:199-256 without the solve/stop logic of :201-214,258-266,
 *                                           which belongs to the host mirror)
 * meas [batch][k_stride][3] float32 {id, range, bearing}, meas_count [batch].  HOST pointers. */
int pgs_update(pgs_handle* h, const float cmd[2], const float* meas, const int32_t* meas_count, int k_stride,
               const double* sec_pose);
/* Same with DEVICE pointers (e.g. the EKF engine's measurement dump and pose buffers). */
int pgs_update_dev(pgs_handle* h, const float cmd[2], const float* d_meas, const int32_t* d_meas_count, int k_stride,
                   const double* d_sec_pose);
/* T iterations with everything on the device: measurement generator get_cmd (sim_node.py:209-250) with the
 * instance's noise stream, NaiveFilter::update as the secondary filter (filter.h:342-348, params.yaml:60), then the
 * two calls above.  cmds [T][2] float32 host array. */
int pgs_run_sim(pgs_handle* h, const float* cmds, int T);

/* PoseGraph::solvePoseGraph (pose_graph.cpp:269-300) for every instance: LM from initial_estimate to `result`.
 * The trials of a batch run in lockstep; once few instances are still active each runs the next lambdas of GTSAM's retry
 * sequence (lambda, 10 lambda, ...) at once in spare slots and the sequential accept / lambda logic is replayed over them, so
 * the result, pgs_get_stats' iteration and trial counts are those of the sequential loop.  The slots multiply the LM work
 * space: SLAM_PGS_LANES slots per instance (default 4, 1 = off; reduced at pgs_create if they would take more than half of
 * the free device memory).
 * Launch shape (results do not depend on it beyond the tolerance of the SYRK's summation order): the kernels of a trial run over
 * the compacted list of running slots (SLAM_PGS_LIST=0: full-size grids); with at most 128 running slots, or a batch that fills
 * the device twice, the block-tridiagonal chain and the Schur-complement SYRK are ONE launch with Y kept in LDS, replicated on
 * 2 - 4 CUs per instance (SLAM_PGS_FUSED=0: always two launches; 2 | 3 | 4: that many workgroups per instance).  The fused
 * kernel needs 2 M + 1 <= 448, M <= 176 mapped landmarks and k_per_pose <= 32; other graphs take the two-launch path. */
int pgs_solve(pgs_handle* h);
/* The solve splits the batch into `groups` contiguous ranges that run their LM loops on separate HIP streams (the
 * latency-bound phases of one group overlap the bandwidth-bound phases of another); results do not depend on it.
 * 0 = automatic (2 from 512 instances, else 1; 4 groups measured best in a process without other HIP streams). */
int pgs_set_groups(pgs_handle* h, int groups);
/* Streaming solve (round 6): the handle holds `batch` graphs but at most `slots` of them are IN FLIGHT (0 = lockstep, all of them from
 * the first trial on; SLAM_PGS_SLOTS sets the default).  The others wait; when a graph converges the device-side decide step hands
 * its running slot to the next waiting graph, so every LM trial runs a full list of slots instead of a list that decays to the
 * batch's slowest instances, and the host - which no longer decides anything per trial - enqueues trials ahead
 * (SLAM_PGS_STREAM_DEPTH, default 3) and reads the counters of the trial that far back.  Once nothing waits and at most
 * SLAM_PGS_LANES_SWITCH graphs are left the lockstep loop with its lambda lanes finishes them.  Results, iteration and trial counts
 * do not depend on `slots`: a graph's LM sequence depends on nothing but the graph (the reference solves ONE graph per call,
 * pose_graph.cpp:269-300; the batch and its schedule are this library's). */
int pgs_set_slots(pgs_handle* h, int slots);
/* The last solve's schedule: slots[i] = running slots of trial i of solve group `group` (at most cap entries written), *n = trials of
 * that group, *groups = solve groups of the solve.  Mean occupancy = sum(slots) / (trials x slots per group). */
int pgs_last_solve_timeline(pgs_handle* h, int group, int32_t* slots, int cap, int32_t* n, int32_t* groups);
/* `this->initial_estimate = this->result` (pose_graph.cpp:263, solve_graph_every_iteration). */
int pgs_adopt_result(pgs_handle* h);
/* The reference's DEFAULT mode, solve_graph_every_iteration: true (params.yaml:64; pose_graph.cpp:258-264), with the simulator on the
 * device: T x { one tick of pgs_run_sim, pgs_solve, pgs_adopt_result }.  counts [batch][2] (may be NULL) = the LM iterations and
 * lambda trials of every instance summed over the T ticks. */
int pgs_run_sim_every_iteration(pgs_handle* h, const float* cmds, int T, int32_t* counts);
/* Phase table of the last such call when SLAM_PGS_ITER_PROF=1 was set (a stream synchronisation after every phase; timing runs leave it
 * unset): out = {host-clock ms in simulator + append, in solve, in adopt, LM trials launched, algorithmic FLOP of the call's consumed
 * trials summed over the batch: Schur-complement SYRK, dense Cholesky + substitutions (n^3/3 + 2 n^2 at n = 2 M)}; the last three are
 * always filled. */
int pgs_last_iter_phases(pgs_handle* h, double out[6]);

/* PoseGraphState payload (pose_graph.cpp:302-387, PoseGraphState.msg): which = 0 initial_estimate, 1 result.
 * poses [timestep+1][3] (x, y, yaw), landmarks [M][2], ids [M]; any pointer may be NULL. */
int pgs_get_graph(pgs_handle* h, int instance, int which, double* poses, double* landmarks, int32_t* timestep,
                  int32_t* M, int32_t* ids);
/* msg_measurement_connections (pose_graph.cpp:176-177): pairs (timestep, landmark index; -1 for a first detection).
 * conn [cap][2]; *n = number of pairs (may exceed cap). */
int pgs_get_connections(pgs_handle* h, int instance, int32_t* conn, int cap, int32_t* n);
/* Per instance [batch]: LM iterations, lambda trials, status bits; initial / final objective 0.5*sum|e|^2, final lambda. */
int pgs_get_stats(pgs_handle* h, int32_t* iterations, int32_t* trials, int32_t* flags, double* err_init,
                  double* err_final, double* lambda);
/* compute_average_error as the pose-graph plot calls it (plotting_node.py:203-213,432-434) against the simulator's
 * true poses (pgs_run_sim), per instance [batch]; which = 0 initial graph, 1 result. */
int pgs_error_stats(pgs_handle* h, int which, double* per_instance_avg_err);
/* Work of the LAST pgs_solve summed over instances and their trials: algorithmic FLOP of the Schur-complement SYRK
 * S_ext = D - Y^T Y (2 FLOP per stored lower-triangle element and per k where its row of Y^T can be non-zero, i.e. from
 * the first detection of the row's landmark; independent of the kernel's tiling) and the number of LM trials launched. */
int pgs_last_solve_work(pgs_handle* h, double* syrk_flop, int32_t* trials_launched);
/* Per-kernel timing of pgs_solve with HIP events on the handle's stream (off by default).  ms[6] = total milliseconds
 * of the LAST solve spent in {linearize, chain, syrk, chol, backsolve, evaluate}, summed over its trials. */
int pgs_set_profiling(pgs_handle* h, int on);
int pgs_last_solve_kernel_ms(pgs_handle* h, double ms[6]);
/* The last solve run with profiling on, split by how the Schur complement was formed: out = {algorithmic SYRK FLOP of the trials
 * that ran the separate SYRK kernels, FLOP of the trials that ran the fused chain + SYRK kernel, ms spent in those SYRK
 * launches, ms spent in those fused launches, FLOP of the trials of the SEGMENTED elimination (round 5: the pose chain cut at every
 * SLAM_PGS_SEG-th pose, default 32; segments' interiors side by side, then the separators - the reference's solve,
 * pose_graph.cpp:273-300, in another exact elimination order), ms in its SYRK launches, 1 if the solve ran that order (0: a
 * segment of some instance sees more than 63 landmarks, or SLAM_PGS_SEG=0: the sequential chain of rounds 1-4), the segment length}. */
int pgs_last_solve_paths_v2(pgs_handle* h, double* out, int n /* entries of out, <= 8 */);
/* The first four entries only (the ABI of rounds 1-4; ADVICE r05: a caller with `double out[4]` must stay valid). */
int pgs_last_solve_paths(pgs_handle* h, double out[4]);
int pgs_sync(pgs_handle* h);
int pgs_timestep(const pgs_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_PGS_H */
