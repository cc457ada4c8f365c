python -m pytest tests/test_parity_pgs_gpu.py -x -q -m gpu 2>&1 | tail -2
for b in 2048 256; do python bench.py --filter pgs --batch $b --steps 3 --no-cpu-baseline --no-batch-256 2>&1 | grep -o "\"value\": [0-9.]*\|max_abs_diff_m\": [0-9.e-]*\|counts_equal\": [a-z]*\|\"kernel_ms_per_solve\": {[^}]*}" | tr '\n' ' '; echo; done
python bench.py --filter pgs --iterative --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*\|max_abs_diff_m\": [0-9.e-]*\|counts_equal\": [a-z]*" | tr '\n' ' '; echo
