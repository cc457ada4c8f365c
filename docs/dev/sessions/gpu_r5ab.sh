# round 5, session ab: the artefacts of the FINAL tree (replaces session s): driver command, bench lines, GPU suite, smoke, soaks, pose-graph profile + timeline
OUT=gpurun_out/r05b; rm -rf $OUT; mkdir -p $OUT gpurun_out/r05_pgs
cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench_driver.err | tail -1 > $OUT/bench_driver_line.json
python3 bench.py --filter pgs 2>/dev/null | tail -1 > $OUT/bench_pgs_b256.json
python3 bench.py --filter pgs --batch 1024 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_pgs_b1024.json
python3 bench.py --filter pgs --landmarks 20 --batch 4096 --k-per-pose 8 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_pgs_L20_b4096.json
python3 bench.py --filter ukf --landmarks 20 --batch 4096 --steps 100 --warmup 10 2>/dev/null | tail -1 > $OUT/bench_ukf_L20.json
python3 bench.py --filter ukf --landmarks 50 --batch 4096 --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_ukf_L50.json
python3 bench.py --dtype f32 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_f32_default_window.json
timeout 2400 python3 -m pytest tests -q -m gpu -rs > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
timeout 400 python3 tools/gpu_soak_ekf.py 200 8201 both > $OUT/soak_ekf.txt 2>&1
timeout 300 python3 tools/gpu_soak_api.py 120 8202 > $OUT/soak_api.txt 2>&1
timeout 400 python3 tools/gpu_soak_pgs.py 240 8203 > $OUT/soak_pgs.txt 2>&1
timeout 300 python3 tools/gpu_soak_pgs.py 150 8204 big > $OUT/soak_pgs_big.txt 2>&1
timeout 300 python3 tools/gpu_soak_pgs_api.py 100 8205 > $OUT/soak_pgs_api.txt 2>&1
timeout 300 python3 tools/gpu_soak_adversarial.py 120 8206 both > $OUT/soak_adversarial.txt 2>&1
bash tools/profile_pgs.sh r05_pgs > $OUT/profile_pgs.log 2>&1
python3 tools/summarize_pgs_profile.py gpurun_out/prof_r05_pgs gpurun_out/r05_pgs > /dev/null 2>&1
cp $(find gpurun_out/prof_r05_pgs/stats -name "*kernel_trace.csv" | head -1) gpurun_out/r05_pgs/kernel_trace.csv 2>/dev/null
rm -rf gpurun_out/prof_r05_pgs
SLAM_PGS_PROF=1 python3 tools/gpu_pgs_phases.py > gpurun_out/r05_pgs/chol_phases.txt 2>&1
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r05b/bench_driver_line.json").read())
print("driver:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["secondary_digest"], d["config"].get("steady_state_value"), d["config"].get("full_run_value"))
for f in ("bench_pgs_b256","bench_pgs_b1024","bench_pgs_L20_b4096","bench_ukf_L20","bench_ukf_L50","bench_f32_default_window"):
    try:
        x=json.loads(open("gpurun_out/r05b/"+f+".json").read()); print(f, x["value"], x["ms_per_step"], x["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $OUT/pytest.log | tail -4; tail -n 2 $OUT/smoke.log
for f in soak_ekf soak_api soak_pgs soak_pgs_big soak_pgs_api soak_adversarial; do echo "$f: $(tail -n 1 $OUT/$f.txt | cut -c 1-300)"; done
cat gpurun_out/r05_pgs/summary.txt | head -22; tail -n 2 gpurun_out/r05_pgs/chol_phases.txt
