#!/bin/bash
mkdir -p gpurun_out/r3f
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3f/pytest.log
tail -5 gpurun_out/r3f/pytest.log
PCIE_K=320 timeout 900 python tools/gpu_pcie_rate.py > gpurun_out/r3f/pcie_K320.log 2>&1; grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/r3f/pcie_K320.log | tail -12
