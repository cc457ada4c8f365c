set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 900 python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5b/pgs_tests.log
timeout 300 python -m pytest tests/test_parity_ukf_gpu.py -x -q -m gpu -k checkpoint 2>&1 | tail -15 > gpurun_out/r5b/ukf_ckpt.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5b/prof -o pgs -- python3 $GRAFT_REPO_ROOT/bench.py --filter pgs --steps 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r5b/prof_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/r5b/prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/r5b/prof -name "*kernel_stats*" | head; find gpurun_out/r5b/prof -name "*kernel_trace*" -size +20M -delete
cat gpurun_out/r5b/*.log | tail -12
