/* slam_multi.h — multi-GPU entry points of libslam_hip.so for a SINGLE-PROCESS C / C++ host (SURVEY.md section 8(e) "Host":
 * one host thread, one handle and one HIP stream per device).  north_star: "the instance batch shards embarrassingly over the
 * 8 GPUs of one node (RCCL over xGMI only for the final error-statistics gather)".
 *
 * The reference runs ONE filter in one ROS node (localization_node.cpp:197 ros::spin); a batch of Monte-Carlo instances has no
 * counterpart there, so these calls replace nothing in the reference - they are the batch-side plumbing a C++/ROS host needs to
 * put the global batch on every GPU of the node without going through Python / torch.distributed (bench.py's path, one
 * process per GPU, stays as it is).  Global instance g lives on device shard s with first_s <= g < first_s + count_s, where
 * (first_s, count_s) is the contiguous block partition below (the same one live_ekf_slam_amd/parallel.py::shard_range uses);
 * its noise streams are keyed by g (slam_set_instance_offset), so results do not depend on the number of devices.
 * No data-path collective exists: the shards never exchange anything until slam_multi_error_stats gathers the per-instance
 * statistic of plotting_node.py:195-218.
 */
#ifndef SLAM_MULTI_H
#define SLAM_MULTI_H

#include "slam_batch.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slam_multi slam_multi;

/* Contiguous block partition of global_batch over `world` shards: the first (global_batch % world) shards get one more. */
int slam_shard_range(int64_t global_batch, int shard, int world, int64_t* first, int64_t* count);

/* One slam_handle per entry of devices[] (HIP device ordinals; a device may appear once), shard s on devices[s].
 * Replaces n_devices x { make_unique<EKF|UKF>() + readParams } (localization_node.cpp:33-47). */
int slam_multi_create(const slam_config* cfg, int filter_kind, int64_t global_batch, int L_max, int dtype, const int* devices,
                      int n_devices, slam_multi** out);
int slam_multi_destroy(slam_multi* m);
int slam_multi_devices(const slam_multi* m);
int64_t slam_multi_batch(const slam_multi* m);
/* The handle of shard s (every slam_* entry point works on it; results of instance b there belong to global instance first + b). */
slam_handle* slam_multi_handle(slam_multi* m, int shard);
int slam_multi_shard(const slam_multi* m, int shard, int64_t* first, int64_t* count);

/* The same call on every shard (asynchronous launches: the devices run concurrently; the host thread only enqueues). */
int slam_multi_set_seed(slam_multi* m, uint64_t seed);
int slam_multi_set_vision(slam_multi* m, double range_max, double fov_min, double fov_max);
int slam_multi_set_map(slam_multi* m, const double* map_xy, int L);
int slam_multi_init(slam_multi* m, float x_0, float y_0, float yaw_0);           /* Filter::init on every instance            */
int slam_multi_step_sim(slam_multi* m, const float cmd[2]);                      /* one tick: get_cmd + Filter::update        */
int slam_multi_run_sim(slam_multi* m, const float* cmds, int T);                 /* T ticks                                   */
int slam_multi_sync(slam_multi* m);

/* THE gather: per-instance average position error of all global_batch instances, in global instance order, on the host
 * (compute_average_error, plotting_node.py:195-218).  mode 0: every shard's statistics are copied to the host and
 * concatenated (what a single process needs).  mode 1: the shards all-gather their statistics device to device with RCCL
 * (ncclCommInitAll + ncclAllGather over xGMI, loaded from librccl.so at the first use), then one copy from devices[0]; fails
 * with SLAM_ERR_UNSUPPORTED if librccl.so cannot be loaded.  Both give the same array. */
int slam_multi_error_stats(slam_multi* m, double* per_instance_avg_err, int mode);
/* status flags of all instances, global order (host concat) */
int slam_multi_status(slam_multi* m, int32_t* per_instance_flags);
/* state of GLOBAL instance g (slam_get_state on the shard that owns it) */
int slam_multi_get_state(slam_multi* m, int64_t g, double* x, double* P, int32_t* M, int32_t* ids, int32_t* timestep);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_MULTI_H */
