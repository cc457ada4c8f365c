#!/bin/bash
# round 4, session z: the whole GPU suite, the variant soak (UKF thread-count variants incl. the round-by-round path on padded sizes), API soak
mkdir -p gpurun_out/r4z
timeout 2400 python -m pytest tests -q -m gpu -rs > gpurun_out/r4z/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4z/pytest.log
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r4z/pytest.log | tail -4
timeout 400 python tools/gpu_soak_ekf.py 240 9301 2>&1 | tail -3 | tee gpurun_out/r4z/soak_ekf.txt
timeout 300 python tools/gpu_soak_api.py 120 9302 2>&1 | tail -2 | tee gpurun_out/r4z/soak_api.txt
