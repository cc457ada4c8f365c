#!/bin/bash
# A/B two builds of libslam_hip.so on the same box: tools/libslam_hip_old.so vs the in-tree one
cp live_ekf_slam_amd/libslam_hip.so /tmp/new.so
for rep in 1 2 3; do
  cp tools/libslam_hip_old.so live_ekf_slam_amd/libslam_hip.so; echo -n "old: "; python tools/gpu_ablate.py 0 2>&1 | tail -1
  cp /tmp/new.so live_ekf_slam_amd/libslam_hip.so; echo -n "new: "; python tools/gpu_ablate.py 0 2>&1 | tail -1
done
