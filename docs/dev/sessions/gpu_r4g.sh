#!/bin/bash
# round 4, session g: rocprofv3 kernel trace + PMC of the UKF (configs[2]) on the round-4 sqrt kernel
mkdir -p gpurun_out/r04_ukf
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --filter ukf --batch 4096 --landmarks 20 --steps 100 --no-cpu-baseline > gpurun_out/r04_ukf/bench_line.json 2>/dev/null
python3 bench.py --filter ukf --batch 4096 --landmarks 50 --steps 30 --no-cpu-baseline > gpurun_out/r04_ukf/bench_line_L50.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_ukf/stats -o stats -- python3 bench.py --filter ukf --batch 4096 --landmarks 20 --steps 100 --no-cpu-baseline --no-parity-check > gpurun_out/r04_ukf/stats.log 2>&1
cp $(find gpurun_out/r04_ukf/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r04_ukf/kernel_stats.csv
head -8 gpurun_out/r04_ukf/kernel_stats.csv
bash tools/pmc_ukf.sh > gpurun_out/r04_ukf/pmc_summary.txt 2>&1
cat gpurun_out/r04_ukf/pmc_summary.txt
rm -rf gpurun_out/r04_ukf/stats gpurun_out/prof_ukf_pmc
