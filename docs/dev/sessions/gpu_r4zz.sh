#!/bin/bash
# round 4, last soak pass on the final tree (every soak tool, longer runs)
mkdir -p gpurun_out/r4zz
timeout 800 python tools/gpu_soak_adversarial.py 600 7101 both 2>&1 | tail -3 | tee gpurun_out/r4zz/soak_adversarial_both.txt
timeout 600 python tools/gpu_soak_ekf.py 420 7102 2>&1 | tail -3 | tee gpurun_out/r4zz/soak_ekf.txt
timeout 500 python tools/gpu_soak_pgs.py 300 7103 2>&1 | tail -3 | tee gpurun_out/r4zz/soak_pgs.txt
timeout 400 python tools/gpu_soak_api.py 240 7104 2>&1 | tail -2 | tee gpurun_out/r4zz/soak_api.txt
timeout 400 python tools/gpu_soak_pgs_api.py 200 7105 2>&1 | tail -2 | tee gpurun_out/r4zz/soak_pgs_api.txt
