#!/bin/bash
# round 2, GPU session A: parity of the strip stream + variant sweep + per-k table
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_capi.py -x -q -m gpu > gpurun_out/r2a/parity.log 2>&1; echo "parity rc $?" >> gpurun_out/r2a/parity.log
tail -5 gpurun_out/r2a/parity.log
timeout 900 python tools/gpu_variants.py f64 1444 444 1448 448 1442 1248 1244 1842 1844 1434 1424 > gpurun_out/r2a/variants_f64.log 2>&1
cat gpurun_out/r2a/variants_f64.log
timeout 600 python tools/gpu_variants.py f32 1444 444 1448 1442 1244 1842 > gpurun_out/r2a/variants_f32.log 2>&1
cat gpurun_out/r2a/variants_f32.log
timeout 300 python tools/gpu_steptimes.py f64 644 100 > gpurun_out/r2a/steptimes_f64.log 2>&1; cat gpurun_out/r2a/steptimes_f64.log
timeout 300 python tools/gpu_steptimes.py f64 46 20 > gpurun_out/r2a/steptimes_f64_t46.log 2>&1; cat gpurun_out/r2a/steptimes_f64_t46.log
timeout 300 python tools/gpu_steptimes.py f32 644 100 > gpurun_out/r2a/steptimes_f32.log 2>&1; cat gpurun_out/r2a/steptimes_f32.log
