#!/bin/bash
# round 3: a second pass of every soak with fresh seeds on the final tree
mkdir -p gpurun_out/soak_final
timeout 700 python tools/gpu_soak_ekf.py 540 2001 both   > gpurun_out/soak_final/ekf.log 2>&1;  tail -1 gpurun_out/soak_final/ekf.log
timeout 500 python tools/gpu_soak_pgs.py 360 2002        > gpurun_out/soak_final/pgs.log 2>&1;  tail -1 gpurun_out/soak_final/pgs.log
timeout 400 python tools/gpu_soak_pgs.py 240 2003 big    > gpurun_out/soak_final/pgs_big.log 2>&1; tail -1 gpurun_out/soak_final/pgs_big.log
timeout 500 python tools/gpu_soak_api.py 360 2004        > gpurun_out/soak_final/api.log 2>&1;  tail -1 gpurun_out/soak_final/api.log
timeout 500 python tools/gpu_soak_adversarial.py 360 2005 both > gpurun_out/soak_final/adv.log 2>&1; tail -1 gpurun_out/soak_final/adv.log
timeout 400 python tools/gpu_soak_pgs_api.py 240 2006    > gpurun_out/soak_final/pgs_api.log 2>&1; tail -1 gpurun_out/soak_final/pgs_api.log
grep -h MISMATCH gpurun_out/soak_final/*.log | head -10 | cut -c1-400
