#!/bin/bash
# round 4, session a: where wavefronts land (HW_ID probe), EKF parity on the tree without the split-control path and with the
# lockstep gather fix (one-wavefront variant back in the build), A/B of the workgroups-per-CU variants (KP = 2)
mkdir -p gpurun_out/r4a
for cfg in "4 40512" "3 30352" "2 26960" "2 30352"; do tools/ubench_hwid $cfg 16384; done > gpurun_out/r4a/hwid.txt 2>&1
head -60 gpurun_out/r4a/hwid.txt
timeout 1500 python -m pytest tests/test_parity_gpu.py -q -m gpu -x > gpurun_out/r4a/pytest_ekf.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4a/pytest_ekf.log
tail -5 gpurun_out/r4a/pytest_ekf.log
bash tools/gpu_ab_ekf.sh live_ekf_slam_amd/libslam_hip.so:0 live_ekf_slam_amd/libslam_hip.so:21344 live_ekf_slam_amd/libslam_hip.so:21354 live_ekf_slam_amd/libslam_hip.so:21364 \
  live_ekf_slam_amd/libslam_hip.so:21234 live_ekf_slam_amd/libslam_hip.so:21244 live_ekf_slam_amd/libslam_hip.so:21444 live_ekf_slam_amd/libslam_hip.so:21454 \
  live_ekf_slam_amd/libslam_hip.so:31344 live_ekf_slam_amd/libslam_hip.so:0 2>&1 | tee gpurun_out/r4a/ab.txt
