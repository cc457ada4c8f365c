# round 5, session z: the segmented solve's tile kernel on 64 x 64 wavefront tiles from N running slots (SLAM_PGS_SEG_TILE64=N): parity with it forced, then the bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5z
SLAM_PGS_SEG_TILE64=1 timeout 900 python3 -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu > gpurun_out/r5z/pgs_tests_tile64.txt 2>&1; tail -3 gpurun_out/r5z/pgs_tests_tile64.txt
for thr in 1073741824 192 128 96 64 1073741824 128; do
  SLAM_PGS_SEG_TILE64=$thr python3 bench.py --filter pgs --steps 4 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B256 tile64 from $thr:', round(d['value'],1), d['config'].get('parity_check',{}).get('max_abs_diff_vs_oracle'))"
done
for thr in 1073741824 128; do
  SLAM_PGS_SEG_TILE64=$thr python3 bench.py --filter pgs --batch 1024 --steps 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B1024 tile64 from $thr:', round(d['value'],1))"
done
