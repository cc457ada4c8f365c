mkdir -p gpurun_out/r06b
python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/gpu_soak_pgs.py 300 22 big > gpurun_out/r06b/soak_pgs_big.txt 2>&1; tail -2 gpurun_out/r06b/soak_pgs_big.txt
python tools/gpu_soak_pgs.py 120 31 > gpurun_out/r06b/soak_pgs2.txt 2>&1; tail -2 gpurun_out/r06b/soak_pgs2.txt
