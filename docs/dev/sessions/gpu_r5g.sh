set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
timeout 900 python -m pytest tests/test_parity_pgs_gpu.py tests/test_soak_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5g/pgs_tests.log
timeout 400 python tools/gpu_soak_pgs.py 240 601 > gpurun_out/r5g/soak_pgs.txt 2>&1
tail -4 gpurun_out/r5g/pgs_tests.log; tail -n 4 gpurun_out/r5g/soak_pgs.txt | cut -c 1-600
