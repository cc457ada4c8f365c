set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5c/pgs_tests.log
timeout 300 python -m pytest tests/test_parity_ukf_gpu.py -x -q -m gpu -k checkpoint 2>&1 | tail -15 > gpurun_out/r5c/ukf_ckpt.log
timeout 300 python bench.py --filter pgs --no-cpu-baseline > gpurun_out/r5c/pgs_seg.json 2> gpurun_out/r5c/pgs_seg.err
SLAM_PGS_SEG=16 timeout 300 python bench.py --filter pgs --no-cpu-baseline > gpurun_out/r5c/pgs_seg16.json 2> gpurun_out/r5c/pgs_seg16.err
SLAM_PGS_SEG=24 timeout 300 python bench.py --filter pgs --no-cpu-baseline > gpurun_out/r5c/pgs_seg24.json 2> gpurun_out/r5c/pgs_seg24.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5c/prof -o pgs -- python3 $GRAFT_REPO_ROOT/bench.py --filter pgs --steps 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r5c/prof_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/r5c/prof.err
cd $GRAFT_REPO_ROOT
cat gpurun_out/r5c/*.log | tail -12
