# round 5, session v: A/B on one box of the library before / after the long-message path (an early return at the entry of the EKF and UKF step kernels)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5v
for i in 1 2; do
  for v in head new; do
    SLAM_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libslam_hip_$v.so python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r5v/driver_${v}_$i.json
    python3 -c "
import json
d=json.loads(open('gpurun_out/r5v/driver_${v}_$i.json').read()); print('$v', $i, d['value'], d['roofline']['frac'], d['config']['secondary_digest'])"
  done
done
for v in head new head new; do
  SLAM_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libslam_hip_$v.so python3 bench.py --filter ukf --batch 4096 --landmarks 20 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v ukf', d['value'])"
done
