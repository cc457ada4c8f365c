#!/bin/bash
# round 4, session r: soaks of the UKF on the quadruple schedule (long trajectories, adversarial messages, every size class), then the whole GPU suite
mkdir -p gpurun_out/r4r
timeout 900 python tools/gpu_ukf_long_parity.py 2>&1 | tail -3 | tee gpurun_out/r4r/ukf_long_parity.txt
timeout 700 python tools/gpu_soak_adversarial.py 420 4101 ukf 2>&1 | tail -4 | tee gpurun_out/r4r/soak_adversarial_ukf.txt
timeout 2400 python -m pytest tests -q -m gpu -rs > gpurun_out/r4r/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4r/pytest.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r4r/pytest.log | tail -6
