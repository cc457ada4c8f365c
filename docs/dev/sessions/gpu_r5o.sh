# round 5, session o: the pose-graph secondary leg inside the driver's command - hardware queues (GPU_MAX_HW_QUEUES) and solve groups
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['config']['secondary_digest'])" $1 >> gpurun_out/r5o/summary.txt; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5o/default.json; show gpurun_out/r5o/default.json
GPU_MAX_HW_QUEUES=8 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5o/q8.json; show gpurun_out/r5o/q8.json
SLAM_PGS_GROUPS=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5o/g1.json; show gpurun_out/r5o/g1.json
GPU_MAX_HW_QUEUES=2 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5o/q2.json; show gpurun_out/r5o/q2.json
cat gpurun_out/r5o/summary.txt
python3 bench.py --filter pgs --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('filter pgs steps 2', d['value'])" >> gpurun_out/r5o/summary.txt
tail -n 1 gpurun_out/r5o/summary.txt
