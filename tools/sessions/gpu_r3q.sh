#!/bin/bash
mkdir -p gpurun_out/r3q
timeout 1200 python -m pytest tests -q -x -m gpu -k "pgs or pose or bench" > gpurun_out/r3q/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3q/pytest.log
timeout 600 python bench.py --filter pgs --batch 256 --steps 5 --warmup 2 > gpurun_out/r3q/pgs.json 2> gpurun_out/r3q/pgs.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3q/pgs.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:900]); print(d['config']['kernel_ms_per_solve'], d.get('cpu_baseline'))
PY
