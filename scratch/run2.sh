python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -5
python bench.py --filter pgs --iterative --landmarks 20 --poses 1000 --batch 256 --k-per-pose 8 > gpurun_out/pgsit_L20.json 2> gpurun_out/pgsit_L20.err; tail -c 1500 gpurun_out/pgsit_L20.json; tail -3 gpurun_out/pgsit_L20.err
python bench.py --filter pgs --iterative --batch 256 > gpurun_out/pgsit_L200.json 2> gpurun_out/pgsit_L200.err; tail -c 1500 gpurun_out/pgsit_L200.json; tail -3 gpurun_out/pgsit_L200.err
SLAM_PGS_ITER_PROF=1 python bench.py --filter pgs --iterative --batch 256 --no-cpu-baseline --no-parity-check 2>&1 | tail -c 600
