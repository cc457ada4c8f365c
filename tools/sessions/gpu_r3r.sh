#!/bin/bash
# pose graph: speculative-lane switch points with the round-3 kernels
mkdir -p gpurun_out/r3r
for cfg in "64 16" "100 16" "128 16" "64 32" "100 32" "128 32" "128 64" "256 32" "100 50"; do
  set -- $cfg
  SLAM_PGS_LANES_SWITCH=$1 SLAM_PGS_LANES_SWITCH_ALL=$2 timeout 600 python bench.py --filter pgs --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3r/pgs_$1_$2.json 2> gpurun_out/r3r/pgs_$1_$2.err
  python - $1 $2 <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r3r/pgs_{sys.argv[1]}_{sys.argv[2]}.json').read().strip().splitlines()[-1])
print('switch',sys.argv[1],'all',sys.argv[2],d['value'],d['ms_per_step'],d['config']['lm_trials_launched_per_solve'])
PY
done
