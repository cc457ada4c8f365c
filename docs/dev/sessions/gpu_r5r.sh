# round 5, session r: the segments' chains as lanes of one workgroup per slot (pgs_seg_chain_kernel) + the column kernel
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5r
timeout 900 python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r5r/tests.log
tail -n 3 gpurun_out/r5r/tests.log
timeout 300 python bench.py --filter pgs --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5r/pgs.json
SLAM_PGS_GROUPS=1 timeout 300 python bench.py --filter pgs --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5r/pgs_g1.json
timeout 300 python bench.py --filter pgs --batch 1024 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5r/pgs_b1024.json
python -c "
import json
for f in ('pgs','pgs_g1','pgs_b1024'):
    d=json.loads(open('gpurun_out/r5r/%s.json'%f).read()); print(f, d['value'], d['config']['kernel_ms_per_solve'], d['config']['parity_check']['max_abs_diff_m'], d['config']['parity_check']['lm_iteration_and_trial_counts_equal'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5r/prof -o pgs -- python3 $GRAFT_REPO_ROOT/bench.py --filter pgs --steps 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r5r/prof_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/r5r/prof.err
cd $GRAFT_REPO_ROOT
timeout 300 python tools/gpu_soak_pgs.py 120 1101 > gpurun_out/r5r/soak.txt 2>&1; tail -n 1 gpurun_out/r5r/soak.txt | cut -c 1-300
