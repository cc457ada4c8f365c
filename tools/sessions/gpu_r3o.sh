#!/bin/bash
# pose graph: fused chain + SYRK with 2 / 3 / 4 workgroups per instance chosen per trial
mkdir -p gpurun_out/r3o
timeout 1200 python -m pytest tests -q -x -m gpu -k "pgs or pose" > gpurun_out/r3o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3o/pytest.log
for f in -1 0 2; do
  SLAM_PGS_FUSED=$f timeout 600 python bench.py --filter pgs --batch 256 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3o/pgs_f$f.json 2> gpurun_out/r3o/pgs_f$f.err
  python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(f'gpurun_out/r3o/pgs_f{sys.argv[1]}.json').read().strip().splitlines()[-1])
    print('fused',sys.argv[1],d['value'],d['ms_per_step'],d['roofline']['frac'])
except Exception as e:
    print('fused',sys.argv[1],'failed',e); print(open(f'gpurun_out/r3o/pgs_f{sys.argv[1]}.err').read()[-1500:])
PY
done
python tools/gpu_pgs_trial_kernels.py 256 > gpurun_out/r3o/trace256.log 2>&1
for col in 6 7 8; do grep "trial kernels" gpurun_out/r3o/trace256.log | awk -v c=$col '{print $c}' | tr "\n" " "; echo; done
tail -1 gpurun_out/r3o/trace256.log
