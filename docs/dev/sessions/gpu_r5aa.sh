# round 5, session aa: tile-kernel epilogue with a ballot over the segments (one round trip for the relevance test) and pipelined index loads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5aa
timeout 900 python3 -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu > gpurun_out/r5aa/pgs_tests.txt 2>&1; tail -3 gpurun_out/r5aa/pgs_tests.txt
for i in 1 2; do
python3 bench.py --filter pgs --steps 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B256:', round(d['value'],1), d['config'].get('kernel_ms_per_solve'))"
done
python3 bench.py --filter pgs --batch 1024 --steps 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B1024:', round(d['value'],1))"
timeout 300 python3 tools/gpu_soak_pgs.py 200 3101 > gpurun_out/r5aa/soak_pgs.txt 2>&1; tail -n 1 gpurun_out/r5aa/soak_pgs.txt | cut -c 1-300
