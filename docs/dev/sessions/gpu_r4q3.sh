#!/bin/bash
# quad schedule in the other fast variants (passes without the table): parity, then L = 50 against the round-3 schedule
timeout 1200 python -m pytest tests/test_parity_ukf_gpu.py -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
for lib in "" tools/lib_old.so; do
  if [ -n "$lib" ]; then export SLAM_HIP_LIB=$PWD/$lib; else unset SLAM_HIP_LIB; fi
  echo "== ${lib:-this tree} L=50"
  python bench.py --filter ukf --landmarks 50 --batch 4096 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('mean_jacobi_sweeps'), d['config'].get('parity_check',{}).get('max_abs_diff'))"
done
