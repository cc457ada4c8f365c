#!/bin/bash
# round 4, session q: UKF sqrt kernel on the pass table (quadruple schedule): parity with the oracle, then speed
mkdir -p gpurun_out/r4q
timeout 1500 python -m pytest tests/test_parity_ukf_gpu.py -x -q 2>&1 | tail -15
python bench.py --filter ukf --landmarks 20 --batch 4096 --steps 100 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r4q/bench_ukf_L20.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4q/bench_ukf_L20.json").read())
print(d["value"], d["ms_per_step"], d["config"].get("mean_jacobi_sweeps"), d["config"].get("parity_check"))
PY
