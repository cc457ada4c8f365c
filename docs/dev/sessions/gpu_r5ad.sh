# round 5, session ad: dense Cholesky with the previous panel's L kept in LDS for the next panel's completion step (phase F without a round trip through L2)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ad
timeout 900 python3 -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu > gpurun_out/r5ad/pgs_tests.txt 2>&1; tail -3 gpurun_out/r5ad/pgs_tests.txt
SLAM_PGS_PROF=1 python3 tools/gpu_pgs_phases.py 2>&1 | tail -2 | cut -c1-250
for i in 1 2; do
python3 bench.py --filter pgs --steps 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B256:', round(d['value'],1), d['config'].get('kernel_ms_per_solve'), d['config'].get('parity_check'))"
done
python3 bench.py --filter pgs --batch 1024 --steps 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B1024:', round(d['value'],1))"
timeout 300 python3 tools/gpu_soak_pgs.py 200 5101 > gpurun_out/r5ad/soak_pgs.txt 2>&1; tail -n 1 gpurun_out/r5ad/soak_pgs.txt | cut -c 1-300
timeout 200 python3 tools/gpu_soak_pgs.py 100 5102 big > gpurun_out/r5ad/soak_pgs_big.txt 2>&1; tail -n 1 gpurun_out/r5ad/soak_pgs_big.txt | cut -c 1-300
