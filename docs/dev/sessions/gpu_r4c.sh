#!/bin/bash
# round 4, session c: slots per batch of the control wavefront's thin downdate (SG) x W; UKF with 16-byte V pairs vs before (parity + A/B)
mkdir -p gpurun_out/r4c
L=live_ekf_slam_amd/libslam_hip.so
bash tools/gpu_ab_ekf.sh $L:0 tools/lib_sg2.so:0 tools/lib_sg5.so:0 tools/lib_sg9.so:0 $L:1364 tools/lib_sg5.so:1364 tools/lib_sg9.so:1364 $L:0 2>&1 | tee gpurun_out/r4c/ab_sg.txt
timeout 1200 python -m pytest tests/test_parity_ukf_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r4c/pytest_ukf.txt
bash tools/gpu_ab_ukf.sh tools/lib_noilp.so $L 2>&1 | tee gpurun_out/r4c/ab_ukf.txt
