#!/bin/bash
# round 4, session u: artefacts on the final tree - the driver's command, the UKF / pose-graph legs, UKF kernel trace + counters,
# the whole GPU suite and the smoke test
OUT=gpurun_out/r4u; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench_driver.err | tail -1 > $OUT/bench_driver_line.json
python3 bench.py --filter ukf --landmarks 20 --batch 4096 --steps 100 --warmup 10 2>/dev/null | tail -1 > $OUT/bench_line.json
python3 bench.py --filter ukf --landmarks 50 --batch 4096 --steps 40 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_line_L50.json
python3 bench.py --filter pgs --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_pgs_b256.json
for L in 20 50; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_L$L -o stats -- python3 bench.py --filter ukf --landmarks $L --batch 4096 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/stats_L$L.log 2>&1
  f=$(find $OUT/stats_L$L -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_L$L.csv
done
rm -rf $OUT/stats_L20 $OUT/stats_L50 $OUT/stats_L20.log $OUT/stats_L50.log
bash tools/pmc_ukf.sh > $OUT/pmc_summary_quad.txt 2>&1
python3 tools/gpu_ukf_sqrt_phases.py 20 > $OUT/phases_L20.txt 2>&1
python3 tools/gpu_ukf_sqrt_phases.py 50 > $OUT/phases_L50.txt 2>&1
timeout 2400 python3 -m pytest tests -q -m gpu -rs > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r4u/bench_driver_line.json").read())
print("driver:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("device_time"))
for s in d.get("secondary", []): print("   ", s.get("metric","")[:60], s.get("value"), s.get("unit"))
for f in ("bench_line.json","bench_line_L50.json","bench_pgs_b256.json"):
    x=json.loads(open("gpurun_out/r4u/"+f).read()); print(f, x["value"], x["ms_per_step"])
PY
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $OUT/pytest.log | tail -4; tail -3 $OUT/smoke.log
