#!/bin/bash
# round 4, session o: pipelined left-looking Cholesky of the pose-graph solve: parity suite, random soak, A/B against the right-looking kernel, bench lines
mkdir -p gpurun_out/r4o
timeout 900 python -m pytest tests/test_parity_pgs_gpu.py -q -m gpu 2>&1 | tail -2 | tee gpurun_out/r4o/pytest.txt
python tools/gpu_pgs_chol_ab.py 256 2>&1 | tail -5 | tee gpurun_out/r4o/chol_ab.txt
SLAM_PGS_CHOL_LL=1 python tools/gpu_pgs_phases.py 2>&1 | tail -2 | tee gpurun_out/r4o/phases.txt
python tools/gpu_soak_pgs.py 150 77 2>&1 | tail -3 | tee gpurun_out/r4o/soak.txt
python tools/gpu_soak_pgs.py 120 78 big 2>&1 | tail -3 | tee -a gpurun_out/r4o/soak.txt
python bench.py --filter pgs > gpurun_out/r4o/bench_pgs_b256.json 2>/dev/null; python bench.py --filter pgs --batch 1024 --no-cpu-baseline > gpurun_out/r4o/bench_pgs_b1024.json 2>/dev/null
python - <<'PY'
import json
for f in ("b256","b1024"):
    d=json.loads(open(f"gpurun_out/r4o/bench_pgs_{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["config"]["kernel_ms_per_solve"], d["config"].get("parity_check"))
PY
