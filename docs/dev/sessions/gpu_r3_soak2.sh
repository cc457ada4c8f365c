#!/bin/bash
# round 3: third soak pass, EKF-heavy, fresh seeds
mkdir -p gpurun_out/soak_final2
timeout 1000 python tools/gpu_soak_ekf.py 840 3001 ekf  > gpurun_out/soak_final2/ekf.log 2>&1;  tail -1 gpurun_out/soak_final2/ekf.log
timeout 500 python tools/gpu_soak_ekf.py 360 3002 ukf   > gpurun_out/soak_final2/ukf.log 2>&1;  tail -1 gpurun_out/soak_final2/ukf.log
timeout 500 python tools/gpu_soak_adversarial.py 360 3003 both > gpurun_out/soak_final2/adv.log 2>&1; tail -1 gpurun_out/soak_final2/adv.log
timeout 400 python tools/gpu_soak_api.py 240 3004       > gpurun_out/soak_final2/api.log 2>&1;  tail -1 gpurun_out/soak_final2/api.log
grep -h MISMATCH gpurun_out/soak_final2/*.log | head -10 | cut -c1-500
