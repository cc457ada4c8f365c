# round 5, session l: pose graph at batch 256 - solve groups and lambda-lane thresholds on the segmented path
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
for g in 1 2 3 4; do
  echo "groups=$g" >> gpurun_out/r5l/groups.txt
  SLAM_PGS_GROUPS=$g timeout 300 python bench.py --filter pgs --no-cpu-baseline --steps 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['lm_trials_launched_per_solve'], d['config']['parity_check']['max_abs_diff_m'])" >> gpurun_out/r5l/groups.txt
done
for sw in "64 16" "128 32" "256 64" "32 8"; do
  set -- $sw
  echo "lanes_switch=$1 all=$2" >> gpurun_out/r5l/groups.txt
  SLAM_PGS_LANES_SWITCH=$1 SLAM_PGS_LANES_SWITCH_ALL=$2 timeout 300 python bench.py --filter pgs --no-cpu-baseline --steps 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['lm_trials_launched_per_solve'], d['config']['parity_check']['max_abs_diff_m'])" >> gpurun_out/r5l/groups.txt
done
cat gpurun_out/r5l/groups.txt
