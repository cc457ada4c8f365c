#!/bin/bash
mkdir -p gpurun_out/r3p
for sw in 100 160 60; do
  SLAM_PGS_SYRK_INST_SWITCH=$sw timeout 600 python bench.py --filter pgs --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3p/pgs_sw$sw.json 2> gpurun_out/r3p/pgs_sw$sw.err
  python - $sw <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r3p/pgs_sw{sys.argv[1]}.json').read().strip().splitlines()[-1])
print('inst switch',sys.argv[1],d['value'],d['ms_per_step'],d['roofline']['frac'])
PY
done
for B in 64 1024; do
for f in -1 0; do
  SLAM_PGS_FUSED=$f timeout 900 python bench.py --filter pgs --batch $B --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3p/pgs_B${B}_f$f.json 2> gpurun_out/r3p/pgs_B${B}_f$f.err
  python - $B $f <<'PY'
import json,sys
try:
    d=json.loads(open(f'gpurun_out/r3p/pgs_B{sys.argv[1]}_f{sys.argv[2]}.json').read().strip().splitlines()[-1])
    print('batch',sys.argv[1],'fused',sys.argv[2],d['value'],d['ms_per_step'])
except Exception as e:
    print('failed',e); print(open(f'gpurun_out/r3p/pgs_B{sys.argv[1]}_f{sys.argv[2]}.err').read()[-800:])
PY
done; done
