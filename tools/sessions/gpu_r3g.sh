#!/bin/bash
# round 3, session g: full GPU suite on the KG = 5 defaults, the driver's bench command, rocprofv3 + PMC cross-check
mkdir -p gpurun_out/r3g
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3g/pytest.log
tail -6 gpurun_out/r3g/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3g/bench_driver.json 2> gpurun_out/r3g/bench_driver.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3g/bench_driver.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], r['kernel'], 'long', c['steady_state_long_run']['value'], c['steady_state_long_run']['frac'], 'full', c['full_run_from_init']['value'], 'once', r['once_per_step']['value'], r['once_per_step']['frac'], 'parity', c['parity_check']['max_abs_diff'])
for s in d['secondary']: print(' ', s['name'], s.get('value'), s.get('roofline',{}).get('frac'), s.get('error'))
PY
timeout 1200 bash tools/profile.sh r03g 20 5 > gpurun_out/r3g/profile.log 2>&1
tail -8 gpurun_out/r3g/profile.log
