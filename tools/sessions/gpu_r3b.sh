#!/bin/bash
# round 3, session b: suspend / resume of the open update group across launches - parity, then A/B timing
mkdir -p gpurun_out/r3b
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_bench_gpu.py tests/test_host_driver_gpu.py -x -q -m gpu > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest.log
tail -15 gpurun_out/r3b/pytest.log
for s in 1 0; do
  for K in 20 100; do
    SLAM_SUSPEND=$s timeout 600 python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-secondary --no-long-runs > gpurun_out/r3b/bench_s${s}_K${K}.json 2> gpurun_out/r3b/bench_s${s}_K${K}.err
    echo "suspend=$s K=$K rc=$?"; python - <<PY
import json
d=json.loads(open('gpurun_out/r3b/bench_s${s}_K${K}.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['parity_check']['max_abs_diff'], d['roofline']['once_per_step']['value'])
PY
  done
done
SLAM_SUSPEND=1 timeout 600 python tools/gpu_pcie_rate.py > gpurun_out/r3b/pcie_s1.log 2>&1; tail -12 gpurun_out/r3b/pcie_s1.log
