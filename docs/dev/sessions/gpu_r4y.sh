#!/bin/bash
# UKF: every state size on the passes (n = 2 mod 4 padded by two zero rows, in the oracle and the kernels): parity, speed, soak
mkdir -p gpurun_out/r4y
timeout 1500 python -m pytest tests/test_parity_ukf_gpu.py tests/test_host_driver_gpu.py tests/test_ros_adapter.py -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
for L in 20 50; do
  python bench.py --filter ukf --landmarks $L --batch 4096 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('L=$L', d['value'], d['ms_per_step'], d['config'].get('mean_jacobi_sweeps'), d['config'].get('parity_check',{}).get('max_abs_diff'))"
done
timeout 600 python tools/gpu_ukf_long_parity.py 2>&1 | tail -3
timeout 400 python tools/gpu_soak_adversarial.py 240 5101 ukf 2>&1 | tail -3 | tee gpurun_out/r4y/soak_adversarial_ukf.txt
