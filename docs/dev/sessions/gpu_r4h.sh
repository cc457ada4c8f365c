#!/bin/bash
# round 4, session h: random soaks on the round-4 tree (lockstep gather fix, one-wavefront variant back, negative ids, over-long messages with the
# oracle's message-capacity switch, UKF one-barrier Jacobi)
mkdir -p gpurun_out/r4h
python tools/gpu_soak_adversarial.py 150 41 both 2>&1 | tail -15 | tee gpurun_out/r4h/adversarial.txt
python tools/gpu_soak_ekf.py 150 42 2>&1 | tail -8 | tee gpurun_out/r4h/ekf.txt
python tools/gpu_soak_api.py 60 43 2>&1 | tail -5 | tee gpurun_out/r4h/api.txt
