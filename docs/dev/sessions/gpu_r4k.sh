#!/bin/bash
# round 4, session k: new GPU tests (tracked instance vs reused device buffers, corrupted checkpoints, two ranks over RCCL on one GPU, bench contract)
mkdir -p gpurun_out/r4k
timeout 1500 python -m pytest tests/test_host_driver_gpu.py tests/test_bench_gpu.py -q -m gpu -rs > gpurun_out/r4k/pytest.log 2>&1
grep -n "^FAILED\|^ERROR\|passed\|failed\|SKIPPED" gpurun_out/r4k/pytest.log | tail
