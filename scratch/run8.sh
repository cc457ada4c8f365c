run() { echo "--- $*"; env "$@" python bench.py --filter pgs --iterative --batch 256 --no-cpu-baseline --no-parity-check 2>&1 | grep -o "\"value\": [0-9.]*\|lm_trials_launched_per_tick\": [0-9.]*" | tr '\n' ' '; echo; }
run X=1
run SLAM_PGS_LANES_SWITCH_ALL=64
run SLAM_PGS_LANES=8 SLAM_PGS_LANES_SWITCH_ALL=64
run SLAM_PGS_LANES=8 SLAM_PGS_LANES_SWITCH_ALL=32 SLAM_PGS_LANES_SWITCH=128
run SLAM_PGS_LANES=6 SLAM_PGS_LANES_SWITCH_ALL=32
run SLAM_PGS_GROUPS=1 SLAM_PGS_LANES=8 SLAM_PGS_LANES_SWITCH_ALL=64
one() { echo "--- one-shot $*"; env "$@" python bench.py --filter pgs --batch 256 --steps 4 --no-cpu-baseline --no-parity-check 2>&1 | grep -o "\"value\": [0-9.]*\|lm_trials_launched_per_solve\": [0-9.]*" | tr '\n' ' '; echo; }
one X=1
one SLAM_PGS_LANES_SWITCH_ALL=64
one SLAM_PGS_LANES=8 SLAM_PGS_LANES_SWITCH_ALL=64
one SLAM_PGS_LANES=8 SLAM_PGS_LANES_SWITCH_ALL=32 SLAM_PGS_LANES_SWITCH=128
