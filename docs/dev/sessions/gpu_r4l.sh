#!/bin/bash
# round 4, session l: the driver's command on the round-4 tree (full line with secondary legs), rocprofv3 kernel trace + PMC of the headline, phase table
mkdir -p gpurun_out/r04a
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04a/bench_driver_line.json 2> gpurun_out/r04a/bench_driver.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04a/bench_driver_line.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], r['kernel'], 'long', c['steady_state_long_run']['value'], c['steady_state_long_run']['frac'], 'full', c['full_run_from_init']['value'], 'parity', c['parity_check']['max_abs_diff'])
o=r['once_per_step']; print(' once', o['value'], 'frac', o['frac'], 'k', o['mean_detections_per_step'], 'vs', c['mean_detections_per_step'], 'alg-equiv', o['algorithmic_equiv_frac'])
print(' device_time', d['device_time'])
for s in d['secondary']: print(' ', s['name'], s.get('value'), s.get('roofline',{}).get('frac'), s.get('config',{}).get('parity_check'), s.get('error'))
PY
bash tools/profile.sh r04a 20 5 > gpurun_out/r04a/profile.log 2>&1; tail -25 gpurun_out/r04a/profile.log
cp gpurun_out/prof_r04a/summary.json gpurun_out/prof_r04a/summary.txt gpurun_out/r04a/ 2>/dev/null
cp $(find gpurun_out/prof_r04a/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r04a/kernel_stats.csv
python tools/gpu_phases.py f64 > gpurun_out/r04a/phases.txt 2>&1; cat gpurun_out/r04a/phases.txt
rm -rf gpurun_out/prof_r04a
