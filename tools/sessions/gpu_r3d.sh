#!/bin/bash
# round 3, session d: multi-GPU entry points, tracked publishState, per-tick rates
mkdir -p gpurun_out/r3d
timeout 1500 python -m pytest tests/test_host_driver_gpu.py tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3d/pytest.log
tail -25 gpurun_out/r3d/pytest.log
PCIE_K=320 timeout 900 python tools/gpu_pcie_rate.py > gpurun_out/r3d/pcie_K320.log 2>&1; cat gpurun_out/r3d/pcie_K320.log | tail -14
./live_ekf_slam_amd/filter_driver run_multi ekf 65536 50 100 1 1234 1
