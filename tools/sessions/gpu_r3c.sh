#!/bin/bash
# round 3, session c: full GPU suite (KCAP = LMAX, counters), bench at L = 100, ceiling of the in-place stream pattern
mkdir -p gpurun_out/r3c
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c/pytest.log
tail -6 gpurun_out/r3c/pytest.log
timeout 600 python bench.py --landmarks 100 --batch 16384 --steps 20 --warmup 5 --no-cpu-baseline --no-long-runs > gpurun_out/r3c/bench_L100.json 2> gpurun_out/r3c/bench_L100.err; echo "L100 rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c/bench_L100.json').read().strip().splitlines()[-1])
print(d['value'], d['config']['parity_check'], d['config']['instances_flagged'], d['roofline']['kernel'])
PY
timeout 300 ./tools/calib_pingpong > gpurun_out/r3c/calib_pingpong.log 2>&1; cat gpurun_out/r3c/calib_pingpong.log
