set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5i/gpu_tests.log
tail -n 5 gpurun_out/r5i/gpu_tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5i/bench_driver_line.json 2> gpurun_out/r5i/bench.err
tail -c 3000 gpurun_out/r5i/bench_driver_line.json
