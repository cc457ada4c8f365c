set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r5a/pgs_tests.log
timeout 300 python -m pytest tests/test_parity_ukf_gpu.py -x -q -m gpu -k checkpoint 2>&1 | tail -15 > gpurun_out/r5a/ukf_ckpt.log
timeout 300 python bench.py --filter pgs > gpurun_out/r5a/pgs_seg.json 2> gpurun_out/r5a/pgs_seg.err
SLAM_PGS_SEG=0 timeout 300 python bench.py --filter pgs > gpurun_out/r5a/pgs_seq.json 2> gpurun_out/r5a/pgs_seq.err
SLAM_PGS_SEG=16 timeout 300 python bench.py --filter pgs > gpurun_out/r5a/pgs_seg16.json 2> gpurun_out/r5a/pgs_seg16.err
tail -3 gpurun_out/r5a/*.log; tail -c 600 gpurun_out/r5a/*.err
