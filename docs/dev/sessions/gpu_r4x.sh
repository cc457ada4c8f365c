#!/bin/bash
# pose graph: fused multiply-adds in the chain producer (same-box A/B against the previous library), parity tests, soak
mkdir -p gpurun_out/r4x
for lib in "" tools/lib_old.so ""; do
  if [ -n "$lib" ]; then export SLAM_HIP_LIB=$PWD/$lib; else unset SLAM_HIP_LIB; fi
  echo "== ${lib:-this tree}"
  python bench.py --filter pgs --steps 5 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['kernel_ms_per_solve'], d['config']['parity_check'])"
done
unset SLAM_HIP_LIB
python tools/gpu_pgs_fused_wg.py 3 2>&1 | tail -2
timeout 900 python -m pytest tests/test_parity_pgs_gpu.py -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 500 python tools/gpu_soak_pgs.py 300 9201 2>&1 | tail -3 | tee gpurun_out/r4x/soak_pgs.txt
