#!/bin/bash
# round 4, session t: profile artefacts of the UKF on the quadruple schedule (kernel trace + stats, counters, bench lines at L = 20 and 50),
# and the L = 50 step kernel with 512 threads (220 VGPRs, no spills) against the default 1024 (128 VGPRs, 82 spilled)
OUT=gpurun_out/r4t; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --filter ukf --landmarks 20 --batch 4096 --steps 100 --warmup 10 2>/dev/null | tail -1 > $OUT/bench_line.json
python3 bench.py --filter ukf --landmarks 50 --batch 4096 --steps 40 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_line_L50.json
SLAM_UKF_TPB=10240512 python3 bench.py --filter ukf --landmarks 50 --batch 4096 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_line_L50_step512.json
for L in 20 50; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_L$L -o stats -- python3 bench.py --filter ukf --landmarks $L --batch 4096 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/stats_L$L.log 2>&1
  f=$(find $OUT/stats_L$L -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_L$L.csv
done
rm -rf $OUT/stats_L20 $OUT/stats_L50
bash tools/pmc_ukf.sh > $OUT/pmc_summary_quad.txt 2>&1
python3 tools/gpu_ukf_sqrt_phases.py 20 > $OUT/phases_L20.txt 2>&1
python3 tools/gpu_ukf_sqrt_phases.py 50 > $OUT/phases_L50.txt 2>&1
python3 - <<'PY'
import json
for f in ("bench_line.json","bench_line_L50.json","bench_line_L50_step512.json"):
    d=json.loads(open("gpurun_out/r4t/"+f).read()); print(f, d["value"], d["ms_per_step"], d["config"].get("parity_check",{}).get("max_abs_diff"))
PY
head -4 $OUT/kernel_stats_L20.csv; head -4 $OUT/kernel_stats_L50.csv
