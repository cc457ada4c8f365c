run() { echo "--- $*"; env $1 python bench.py --filter pgs --iterative --batch $2 --no-cpu-baseline --no-parity-check 2>&1 | grep -o "\"value\": [0-9.]*\|lm_trials_launched_per_tick\": [0-9.]*\|seconds_per_run\": [0-9.]*" | tr '\n' ' '; echo; }
run SLAM_PGS_GROUPS=1 1024
run SLAM_PGS_GROUPS=2 1024
run SLAM_PGS_GROUPS=4 1024
run SLAM_PGS_GROUPS=2 2048
run SLAM_PGS_GROUPS=4 2048
python scratch/two.py 1 0 999 2>&1 | tail -1
