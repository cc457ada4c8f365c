# round 5, session ao: Cholesky on 512 threads, tiles formed in pairs with eight 16-k blocks in flight each (SLAM_PGS_CHOL_LL=3) against the left-looking kernel (=1), one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ao
SLAM_PGS_CHOL_LL=3 timeout 900 python3 -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu > gpurun_out/r5ao/pgs_tests_ll3.txt 2>&1; tail -3 gpurun_out/r5ao/pgs_tests_ll2.txt
for i in 1 2; do
for v in 1 3; do
SLAM_PGS_CHOL_LL=$v python3 bench.py --filter pgs --steps 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('LL=$v B256:', round(d['value'],1), d['config'].get('kernel_ms_per_solve')['chol'], d['config'].get('parity_check',{}).get('max_abs_diff_m'))"
done
done
for v in 1 3; do
SLAM_PGS_CHOL_LL=$v python3 bench.py --filter pgs --batch 1024 --steps 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('LL=$v B1024:', round(d['value'],1))"
SLAM_PGS_CHOL_LL=$v SLAM_PGS_PROF=1 python3 tools/gpu_pgs_phases.py 2>&1 | tail -2 | head -1 | cut -c1-250
done
SLAM_PGS_CHOL_LL=3 timeout 300 python3 tools/gpu_soak_pgs.py 200 7101 > gpurun_out/r5ao/soak_pgs.txt 2>&1; tail -n 1 gpurun_out/r5ao/soak_pgs.txt | cut -c 1-300
