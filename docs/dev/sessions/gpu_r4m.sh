#!/bin/bash
# round 4, session m: the V-shadow restructure on the generic fast path of the UKF sqrt kernel (L = 50: <104, 1024>): parity, A/B, phases
mkdir -p gpurun_out/r4m
L=live_ekf_slam_amd/libslam_hip.so
timeout 1200 python -m pytest tests/test_parity_ukf_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r4m/pytest_ukf.txt
bash tools/gpu_ab_ukf.sh tools/lib_ukf_prio0.so $L 2>&1 | tee gpurun_out/r4m/ab_ukf.txt
python tools/gpu_ukf_sqrt_phases.py 50 2>&1 | tee gpurun_out/r4m/phases50.txt
