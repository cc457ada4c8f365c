#!/bin/bash
# round 4, session d: UKF sqrt kernel with the V rotations in the shadow of the next parameter phase + six workgroups per CU: parity, A/B
mkdir -p gpurun_out/r4d
L=live_ekf_slam_amd/libslam_hip.so
timeout 1200 python -m pytest tests/test_parity_ukf_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r4d/pytest_ukf.txt
bash tools/gpu_ab_ukf.sh tools/lib_noilp.so $L 2>&1 | tee gpurun_out/r4d/ab_ukf.txt
