# round 5, session aq: Cholesky on 768 threads (168 registers per lane, no scratch), four and eight 16-k blocks of A operands in flight (SLAM_PGS_CHOL_LL=4 / 5) against 1024 threads (=1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5aq
SLAM_PGS_CHOL_LL=5 timeout 900 python3 -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu > gpurun_out/r5aq/pgs_tests_ll5.txt 2>&1; tail -2 gpurun_out/r5aq/pgs_tests_ll5.txt
for i in 1 2; do
for v in 1 4 5; do
SLAM_PGS_CHOL_LL=$v python3 bench.py --filter pgs --steps 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('LL=$v B256:', round(d['value'],1), d['config'].get('kernel_ms_per_solve')['chol'], d['config'].get('parity_check',{}).get('max_abs_diff_m'))"
done
done
for v in 1 4 5; do
SLAM_PGS_CHOL_LL=$v SLAM_PGS_PROF=1 python3 tools/gpu_pgs_phases.py 2>&1 | tail -2 | head -1 | cut -c1-250
done
