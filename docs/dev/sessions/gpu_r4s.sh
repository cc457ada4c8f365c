#!/bin/bash
# round 4, session s: replays of the adversarial configurations that ended with SLAM_INST_SQRT_FAILED on the GPU only (NaNs in the convergence scan), then the soak again
mkdir -p gpurun_out/r4s
for cfg in "ukf 8 23 9 0 1 20 16 294568887 -3" "ukf 8 40 1 0 1 70 8 943926498 -3" "ukf 20 44 3 0 1 20 20 363870538 0" "ukf 20 36 1 0 1 20 20 499497345 0"; do
  SOAK_REPLAY="$cfg" timeout 300 python tools/gpu_soak_adversarial.py 5 1 ukf 2>&1 | tail -2
done
timeout 500 python tools/gpu_soak_adversarial.py 300 4101 ukf 2>&1 | tail -3 | tee gpurun_out/r4s/soak_adversarial_ukf.txt
timeout 300 python tools/gpu_soak_adversarial.py 120 77 both 2>&1 | tail -2 | tee gpurun_out/r4s/soak_adversarial_both.txt
timeout 600 python -m pytest tests/test_soak_gpu.py tests/test_parity_ukf_gpu.py -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
