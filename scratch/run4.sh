python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -5
for a in 1 0; do
echo "--- async=$a L=20"
SLAM_PGS_ITER_ASYNC=$a python bench.py --filter pgs --iterative --landmarks 20 --poses 1000 --batch 256 --k-per-pose 8 --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*\|lm_trials_launched_per_tick\": [0-9.]*\|max_abs_diff_m\": [0-9.e-]*\|counts_equal\": [a-z]*\|lm_trials_per_tick\": [0-9.]*"
echo "--- async=$a L=200"
SLAM_PGS_ITER_ASYNC=$a python bench.py --filter pgs --iterative --batch 256 --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*\|lm_trials_launched_per_tick\": [0-9.]*\|max_abs_diff_m\": [0-9.e-]*\|counts_equal\": [a-z]*\|lm_trials_per_tick\": [0-9.]*"
done
