#!/bin/bash
# L = 50: wavefront-0 priority in the kernels without the table, step kernel with 512 threads
for cfg in "default:" "prio:tools/lib_prio.so" "step512:" "prio+step512:tools/lib_prio.so" "default:"; do
  name=${cfg%%:*}; lib=${cfg#*:}
  if [ -n "$lib" ]; then export SLAM_HIP_LIB=$PWD/$lib; else unset SLAM_HIP_LIB; fi
  case $name in *step512*) export SLAM_UKF_TPB=10240512;; *) unset SLAM_UKF_TPB;; esac
  echo "== $name"
  python bench.py --filter ukf --landmarks 50 --batch 4096 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('parity_check',{}).get('max_abs_diff'))"
done
