python tools/gpu_soak_pgs.py 150 11 wide 2>&1 | tail -4
python tools/gpu_soak_pgs.py 150 12 2>&1 | tail -4
for sl in 32 16 8; do echo "--- SLAM_PGS_SEG=$sl"; SLAM_PGS_SEG=$sl python bench.py --filter pgs --batch 256 --steps 4 --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*\|max_abs_diff_m\": [0-9.e-]*\|counts_equal\": [a-z]*\|\"kernel_ms_per_solve\": {[^}]*}" | tr '\n' ' '; echo; done
