#!/bin/bash
# round 4, session f: UKF sqrt kernel: parameter addresses from the table, flattened convergence scan, priority of wavefront 0: parity, A/B, phases
mkdir -p gpurun_out/r4f
L=live_ekf_slam_amd/libslam_hip.so
timeout 1200 python -m pytest tests/test_parity_ukf_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r4f/pytest_ukf.txt
bash tools/gpu_ab_ukf.sh tools/lib_ukf_r4e.so tools/lib_ukf_prio0.so $L 2>&1 | tee gpurun_out/r4f/ab_ukf.txt
python tools/gpu_ukf_sqrt_phases.py 20 2>&1 | tee gpurun_out/r4f/phases.txt
