#!/bin/bash
# round 4, session b: batched thin downdate of the control wavefront (SLAM_CTRL_ILP) x W = 3 / 4: A/B on the driver's command + phase tables
mkdir -p gpurun_out/r4b
L=live_ekf_slam_amd/libslam_hip.so; O=tools/lib_noilp.so
bash tools/gpu_ab_ekf.sh $O:0 $L:0 $O:1364 $L:1364 $L:1354 $O:0 $L:0 $L:1364 2>&1 | tee gpurun_out/r4b/ab.txt
for spec in "$O 1464" "$L 1464" "$O 1364" "$L 1364"; do set -- $spec; echo "== $1 $2"; SLAM_HIP_LIB=$PWD/$1 python tools/gpu_phases.py f64 $2 2>&1 | grep -v "  0 cycles"; done > gpurun_out/r4b/phases.txt 2>&1
cat gpurun_out/r4b/phases.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x 2>&1 | tail -3
