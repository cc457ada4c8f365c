# round 5, session p: solve-group streams on distinct priorities (hardware queues) - the driver command's pose-graph leg, and --filter pgs
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
show() { python -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['config'].get('secondary_digest'))" $1 >> gpurun_out/r5p/summary.txt; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5p/prio.json; show gpurun_out/r5p/prio.json
SLAM_PGS_GROUP_PRIO=0 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5p/noprio.json; show gpurun_out/r5p/noprio.json
python3 bench.py --filter pgs --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5p/pgs_prio.json; show gpurun_out/r5p/pgs_prio.json
SLAM_PGS_GROUP_PRIO=0 python3 bench.py --filter pgs --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5p/pgs_noprio.json; show gpurun_out/r5p/pgs_noprio.json
python3 bench.py --filter pgs --batch 1024 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5p/pgs_prio_1024.json; show gpurun_out/r5p/pgs_prio_1024.json
timeout 600 python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu -k "groups or shard" 2>&1 | tail -3 >> gpurun_out/r5p/summary.txt
cat gpurun_out/r5p/summary.txt
