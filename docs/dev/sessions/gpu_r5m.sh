# round 5, session m: pose graph at batch 256, two solve groups (new default) x lambda-lane thresholds; batch 128 / 512 / 1024 groups
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
run() { python bench.py --filter pgs --no-cpu-baseline --steps 4 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config']['lm_trials_launched_per_solve'], d['config']['parity_check']['max_abs_diff_m'], d['config']['parity_check']['lm_iteration_and_trial_counts_equal'])" >> gpurun_out/r5m/sweep.txt; }
for sw in "64 16" "96 24" "128 32" "160 40" "128 16" "192 32"; do
  set -- $sw
  echo "B=256 G=2 lanes_switch=$1 all=$2" >> gpurun_out/r5m/sweep.txt
  SLAM_PGS_LANES_SWITCH=$1 SLAM_PGS_LANES_SWITCH_ALL=$2 run
done
for bg in "128 1" "128 2" "512 2" "512 3" "1024 2" "1024 3" "1024 4"; do
  set -- $bg
  echo "B=$1 G=$2" >> gpurun_out/r5m/sweep.txt
  SLAM_PGS_GROUPS=$2 run --batch $1
done
cat gpurun_out/r5m/sweep.txt
