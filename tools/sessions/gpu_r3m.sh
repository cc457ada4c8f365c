#!/bin/bash
# pose graph: chain + SYRK fused into one launch vs the two-launch path
mkdir -p gpurun_out/r3m
timeout 1200 python -m pytest tests -q -x -m gpu -k "pgs or pose" > gpurun_out/r3m/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3m/pytest.log
for f in -1 0; do
  SLAM_PGS_FUSED=$f timeout 600 python bench.py --filter pgs --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3m/pgs_f$f.json 2> gpurun_out/r3m/pgs_f$f.err
  python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(f'gpurun_out/r3m/pgs_f{sys.argv[1]}.json').read().strip().splitlines()[-1])
    print('fused',sys.argv[1],d['value'],d['ms_per_step'],d['roofline']['frac'], d['config'].get('trials_launched'), d['config'].get('mean_avg_error'))
except Exception as e:
    print('fused',sys.argv[1],'failed',e); print(open(f'gpurun_out/r3m/pgs_f{sys.argv[1]}.err').read()[-1500:])
PY
done
python tools/gpu_pgs_trial_kernels.py 256 > gpurun_out/r3m/trace256.log 2>&1
grep "trial kernels" gpurun_out/r3m/trace256.log | awk '{print $6}' | tr "\n" " "; echo; tail -1 gpurun_out/r3m/trace256.log
