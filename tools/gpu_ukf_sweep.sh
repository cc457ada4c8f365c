#!/bin/bash
# sweep UKF threads-per-instance variants: SLAM_UKF_TPB = sqrt_threads*10000 + step_threads
for v in 640128 640064 640256 1280128 2560128 2560256; do SLAM_UKF_TPB=$v python tools/gpu_ukf_time.py 20 x 2>&1 | grep "B=4096"; done
for v in 2560512 2560256 2561024 5120512 10240512 10241024; do SLAM_UKF_TPB=$v python tools/gpu_ukf_time.py 50 x 2>&1 | grep "UKF"; done
