#!/bin/bash
# round 3: pose-graph profile of the final build (kernel trace + PMC passes) and the fp64 MFMA calibration
mkdir -p gpurun_out/prof_r03i_pgs
python bench.py --filter pgs > gpurun_out/prof_r03i_pgs/bench_line.json 2> gpurun_out/prof_r03i_pgs/bench_line.err
python bench.py --filter pgs --batch 1024 --no-cpu-baseline > gpurun_out/prof_r03i_pgs/bench_line_B1024.json 2> gpurun_out/prof_r03i_pgs/bench_line_B1024.err
./tools/calib_mfma64 > gpurun_out/prof_r03i_pgs/calib_mfma64.txt 2>&1
bash tools/profile_pgs.sh r03i_pgs > gpurun_out/prof_r03i_pgs/profile.log 2>&1
python3 tools/summarize_pgs_profile.py gpurun_out/prof_r03i_pgs gpurun_out/prof_r03i_pgs/out > gpurun_out/prof_r03i_pgs/summarize.log 2>&1
tail -3 gpurun_out/prof_r03i_pgs/summarize.log; cat gpurun_out/prof_r03i_pgs/out/summary.txt | head -30
