set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5f/gpu_tests.log
timeout 400 python tools/gpu_soak_pgs.py 240 501 > gpurun_out/r5f/soak_pgs.txt 2>&1
timeout 300 python tools/gpu_soak_pgs.py 120 502 big > gpurun_out/r5f/soak_pgs_big.txt 2>&1
timeout 300 python bench.py --filter pgs > gpurun_out/r5f/pgs_b256.json 2> gpurun_out/r5f/pgs_b256.err
timeout 300 python bench.py --filter pgs --batch 1024 --no-cpu-baseline > gpurun_out/r5f/pgs_b1024.json 2> gpurun_out/r5f/pgs_b1024.err
tail -3 gpurun_out/r5f/gpu_tests.log; tail -2 gpurun_out/r5f/soak_pgs.txt gpurun_out/r5f/soak_pgs_big.txt
