#!/bin/bash
# A/B of two builds of the library on the UKF bench lines: tools/gpu_ab_ukf.sh old.so [new.so = in-tree]
for rep in 1 2; do
  for lib in "$@"; do
    for L in 20 50; do
      echo -n "$(basename $lib) L=$L: "
      SLAM_HIP_LIB=$PWD/$lib python bench.py --filter ukf --batch 4096 --landmarks $L --steps $([ $L = 20 ] && echo 100 || echo 30) --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['mean_jacobi_sweeps'], d['config']['instances_flagged'])"
    done
  done
done
