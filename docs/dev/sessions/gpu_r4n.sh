#!/bin/bash
# round 4, session n: incremental round-robin indices in the generic fast path of the UKF sqrt kernel: parity, A/B, phases, UKF soak with thread-count variants
mkdir -p gpurun_out/r4n
L=live_ekf_slam_amd/libslam_hip.so
timeout 1200 python -m pytest tests/test_parity_ukf_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r4n/pytest_ukf.txt
bash tools/gpu_ab_ukf.sh tools/lib_ukf_r4m.so $L 2>&1 | tee gpurun_out/r4n/ab_ukf.txt
python tools/gpu_ukf_sqrt_phases.py 50 2>&1 | tee gpurun_out/r4n/phases50.txt
for tpb in 1280128 640064 2560256 10241024 5120512; do echo "SLAM_UKF_TPB=$tpb"; SLAM_UKF_TPB=$tpb timeout 600 python -m pytest tests/test_parity_ukf_gpu.py -q -m gpu -x -k "reference_measurement_stream or many_detections" 2>&1 | tail -2; done | tee gpurun_out/r4n/tpb_variants.txt
