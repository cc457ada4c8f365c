#!/bin/bash
# round 3, first GPU session: affected tests, the driver's bench command, rocprofv3 kernel trace + PMC cross-check of the counted bytes
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_bench_gpu.py -x -q -m gpu > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
tail -5 gpurun_out/r3a/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3a/bench_driver.json 2> gpurun_out/r3a/bench_driver.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r3a/bench_driver.err
timeout 1200 bash tools/profile.sh r03a 20 5 > gpurun_out/r3a/profile.log 2>&1
tail -25 gpurun_out/r3a/profile.log
