T0=$(date +%s.%N); python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r06_b.json 2> gpurun_out/bench_r06_b.err
echo "wall seconds: $(echo "$(date +%s.%N) - $T0" | bc)"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r06_b.json') if l.startswith('{')][-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "once", d["roofline"]["once_per_step_frac"], d["roofline"]["once_per_step_value"])
print("digest:", len(d["config"]["secondary_digest"]), d["config"]["secondary_digest"])
for s in d["secondary"]:
    print(s["name"][:60], s.get("value"), s.get("error"), s.get("leg_seconds"), (s.get("roofline") or {}).get("frac"), (s.get("config") or {}).get("batch_256_value"), (s.get("cpu_baseline") or {}).get("value"))
PY
