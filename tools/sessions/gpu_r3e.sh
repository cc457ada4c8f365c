#!/bin/bash
# round 3, session e: watchdog, packing fix, regression check of the headline
mkdir -p gpurun_out/r3e
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_host_driver_gpu.py -x -q -m gpu > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3e/pytest.log
tail -8 gpurun_out/r3e/pytest.log
for K in 20 100; do
  timeout 600 python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r3e/bench_K${K}.json 2> gpurun_out/r3e/bench_K${K}.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r3e/bench_K${K}.json').read().strip().splitlines()[-1])
print($K, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['parity_check']['max_abs_diff'], d['roofline']['once_per_step']['value'], d['config']['steady_state_long_run']['value'], d['config']['full_run_from_init']['value'])
PY
done
PCIE_K=320 timeout 900 python tools/gpu_pcie_rate.py > gpurun_out/r3e/pcie_K320.log 2>&1; grep -v RCCL gpurun_out/r3e/pcie_K320.log | tail -12
