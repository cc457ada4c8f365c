#!/bin/bash
# round 4, session w: the remaining soaks on the final tree (kernel variants incl. the UKF thread-count variants, the C API walk, pose graph)
mkdir -p gpurun_out/r4w
timeout 400 python tools/gpu_soak_ekf.py 240 9101 2>&1 | tail -3 | tee gpurun_out/r4w/soak_ekf.txt
timeout 300 python tools/gpu_soak_api.py 150 9102 2>&1 | tail -3 | tee gpurun_out/r4w/soak_api.txt
timeout 400 python tools/gpu_soak_pgs.py 200 9103 2>&1 | tail -3 | tee gpurun_out/r4w/soak_pgs.txt
