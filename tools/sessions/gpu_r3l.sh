#!/bin/bash
# pose-graph SYRK after the z-row / half-tile trimming: parity tests, then the switch sweep
mkdir -p gpurun_out/r3l
timeout 1200 python -m pytest tests -q -m gpu -k "pgs or pose" > gpurun_out/r3l/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3l/pytest.log
for sw in 160 86 1; do
  SLAM_PGS_SYRK_INST_SWITCH=$sw timeout 600 python bench.py --filter pgs --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3l/pgs_sw$sw.json 2> gpurun_out/r3l/pgs_sw$sw.err
  python - $sw <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r3l/pgs_sw{sys.argv[1]}.json').read().strip().splitlines()[-1])
print('switch',sys.argv[1],d['value'],d['ms_per_step'],d['roofline']['frac'], d['config'].get('parity_check'))
PY
done
SLAM_PGS_SYRK_INST_SWITCH=1 python tools/gpu_pgs_trial_kernels.py 256 > gpurun_out/r3l/trace256_inst.log 2>&1
grep "trial kernels" gpurun_out/r3l/trace256_inst.log | awk '{print $7}' | tr "\n" " "; echo
