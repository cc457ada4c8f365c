#!/bin/bash
# pose graph: active-slot list (compacted launches) on / off, fused chain + SYRK on / off
mkdir -p gpurun_out/r3n
timeout 1200 python -m pytest tests -q -x -m gpu -k "pgs or pose" > gpurun_out/r3n/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3n/pytest.log
for cfg in "1 0" "0 0" "1 -1" "0 -1"; do
  set -- $cfg
  SLAM_PGS_LIST=$1 SLAM_PGS_FUSED=$2 timeout 600 python bench.py --filter pgs --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3n/pgs_l$1_f$2.json 2> gpurun_out/r3n/pgs_l$1_f$2.err
  python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(f'gpurun_out/r3n/pgs_l{sys.argv[1]}_f{sys.argv[2]}.json').read().strip().splitlines()[-1])
    print('list',sys.argv[1],'fused',sys.argv[2],d['value'],d['ms_per_step'],d['roofline']['frac'])
except Exception as e:
    print('list',sys.argv[1],'fused',sys.argv[2],'failed',e); print(open(f'gpurun_out/r3n/pgs_l{sys.argv[1]}_f{sys.argv[2]}.err').read()[-1500:])
PY
done
SLAM_PGS_FUSED=0 python tools/gpu_pgs_trial_kernels.py 256 > gpurun_out/r3n/trace256.log 2>&1
for col in 6 7 8; do grep "trial kernels" gpurun_out/r3n/trace256.log | awk -v c=$col '{print $c}' | tr "\n" " "; echo; done
tail -1 gpurun_out/r3n/trace256.log
