# round 5, session ar: the 768-thread Cholesky as the default (SLAM_PGS_CHOL_LL=2) with and without the A-operand prefetch across the half-phase barrier (=3), against 1024 threads (=1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ar
timeout 900 python3 -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu > gpurun_out/r5ar/pgs_tests.txt 2>&1; tail -2 gpurun_out/r5ar/pgs_tests.txt
SLAM_PGS_CHOL_LL=3 timeout 900 python3 -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu > gpurun_out/r5ar/pgs_tests_ll3.txt 2>&1; tail -2 gpurun_out/r5ar/pgs_tests_ll3.txt
for i in 1 2; do
for v in 1 2 3; do
SLAM_PGS_CHOL_LL=$v python3 bench.py --filter pgs --steps 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('LL=$v B256:', round(d['value'],1), d['config'].get('kernel_ms_per_solve')['chol'], d['config'].get('parity_check',{}).get('max_abs_diff_m'))"
done
done
for v in 2 3; do
SLAM_PGS_CHOL_LL=$v python3 bench.py --filter pgs --batch 1024 --steps 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('LL=$v B1024:', round(d['value'],1))"
done
timeout 300 python3 tools/gpu_soak_pgs.py 200 9901 > gpurun_out/r5ar/soak_pgs.txt 2>&1; tail -n 1 gpurun_out/r5ar/soak_pgs.txt | cut -c 1-300
timeout 200 python3 tools/gpu_soak_pgs.py 100 9902 big > gpurun_out/r5ar/soak_pgs_big.txt 2>&1; tail -n 1 gpurun_out/r5ar/soak_pgs_big.txt | cut -c 1-300
