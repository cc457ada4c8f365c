# round 5, session q: dense Cholesky with the inverse of the diagonal block (MFMA panel solve, backward substitution as a product)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
timeout 900 python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r5q/tests.log
tail -n 4 gpurun_out/r5q/tests.log
timeout 300 python bench.py --filter pgs --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5q/pgs.json
SLAM_PGS_GROUPS=1 timeout 300 python bench.py --filter pgs --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5q/pgs_g1.json
python -c "
import json
for f in ('pgs','pgs_g1'):
    d=json.loads(open('gpurun_out/r5q/%s.json'%f).read()); print(f, d['value'], d['config']['kernel_ms_per_solve'], d['config']['parity_check'])"
SLAM_PGS_PROF=1 python tools/gpu_pgs_phases.py 2>&1 | tail -2
timeout 300 python tools/gpu_soak_pgs.py 150 1001 > gpurun_out/r5q/soak.txt 2>&1; tail -n 2 gpurun_out/r5q/soak.txt | cut -c 1-400
