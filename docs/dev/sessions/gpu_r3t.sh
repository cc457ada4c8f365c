#!/bin/bash
# round 3, session t: full GPU suite + the driver's bench command + smoke on the tree with the fused pose-graph path
mkdir -p gpurun_out/r3t
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r3t/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3t/pytest.log
tail -4 gpurun_out/r3t/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3t/bench_driver.json 2> gpurun_out/r3t/bench_driver.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3t/bench_driver.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], r['kernel'], 'long', c['steady_state_long_run']['value'], 'full', c['full_run_from_init']['value'], 'once', r['once_per_step']['value'], 'parity', c['parity_check']['max_abs_diff'])
for s in d['secondary']: print(' ', s['name'], s.get('value'), s.get('roofline',{}).get('frac'), s.get('error'))
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
