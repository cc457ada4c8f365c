set -u
mkdir -p gpurun_out/r06b
bash tools/profile.sh r06a 20 5 > gpurun_out/r06b/profile_r06a.log 2>&1
tail -40 gpurun_out/prof_r06a/summary.txt
bash tools/profile_pgs.sh r06_pgs > gpurun_out/r06b/profile_r06_pgs.log 2>&1
python3 tools/summarize_pgs_profile.py gpurun_out/prof_r06_pgs gpurun_out/r06_pgs_summary > gpurun_out/r06b/summarize_pgs.log 2>&1
rm -rf gpurun_out/prof_r06_pgs/stats/*/*kernel_trace.csv gpurun_out/prof_r06a/stats/*/*kernel_trace.csv
find gpurun_out/prof_r06a gpurun_out/prof_r06_pgs -name "*.csv" -size +3M -delete
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06b/bench_driver_line.json 2> gpurun_out/r06b/bench_driver_line.err
python bench.py --filter ukf --batch 4096 --landmarks 20 > gpurun_out/r06b/bench_ukf_L20.json 2>/dev/null
python bench.py --filter ukf --batch 4096 --landmarks 50 --steps 20 --warmup 5 > gpurun_out/r06b/bench_ukf_L50.json 2>/dev/null
python bench.py --filter pgs > gpurun_out/r06b/bench_pgs_b2048.json 2>/dev/null
python bench.py --filter pgs --batch 1024 > gpurun_out/r06b/bench_pgs_b1024.json 2>/dev/null
python bench.py --filter pgs --batch 256 > gpurun_out/r06b/bench_pgs_b256.json 2>/dev/null
python bench.py --filter pgs --iterative > gpurun_out/r06b/bench_pgs_iterative_b256.json 2>/dev/null
python bench.py --dtype f32 --no-secondary > gpurun_out/r06b/bench_f32_default_window.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06b/smoke.log 2>&1
python -m pytest tests -q -m gpu > gpurun_out/r06b/pytest.log 2>&1
tail -3 gpurun_out/r06b/pytest.log; tail -2 gpurun_out/r06b/smoke.log
for f in gpurun_out/r06b/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["unit"], "frac", d["roofline"]["frac"], "parity", (d["config"].get("parity_check") or {}).get("max_abs_diff", (d["config"].get("parity_check") or {}).get("max_abs_diff_m")), d["config"].get("secondary_digest",""))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
