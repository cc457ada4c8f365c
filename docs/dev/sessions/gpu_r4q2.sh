#!/bin/bash
# quad-schedule sqrt kernel, two instantiations (pass table / round-robin table): parity tests, then the L = 20 leg against the round-3 schedule
timeout 900 python -m pytest tests/test_parity_ukf_gpu.py -x -q 2>&1 | tail -4
for lib in "" tools/lib_old.so ""; do
  if [ -n "$lib" ]; then export SLAM_HIP_LIB=$PWD/$lib; else unset SLAM_HIP_LIB; fi
  echo "== ${lib:-this tree}"
  python bench.py --filter ukf --landmarks 20 --batch 4096 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('mean_jacobi_sweeps'), d['config'].get('parity_check',{}).get('max_abs_diff'))"
done
