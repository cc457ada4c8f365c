#!/bin/bash
# pose-graph SYRK: active-count switch between the instance-resident and the tile kernel
mkdir -p gpurun_out/r3k
for sw in 160 120 86 60 30 1; do
  SLAM_PGS_SYRK_INST_SWITCH=$sw timeout 600 python bench.py --filter pgs --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3k/pgs_sw$sw.json 2> gpurun_out/r3k/pgs_sw$sw.err
  python - $sw <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r3k/pgs_sw{sys.argv[1]}.json').read().strip().splitlines()[-1])
print('switch',sys.argv[1],d['value'],d['ms_per_step'],d['roofline']['frac'], d['config'].get('parity_check'))
PY
done
