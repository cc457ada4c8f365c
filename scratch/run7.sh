python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r06_a.json 2> gpurun_out/bench_r06_a.err
tail -3 gpurun_out/bench_r06_a.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_r06_a.json') if l.startswith('{')][-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "once", d["roofline"]["once_per_step_frac"], d["roofline"]["once_per_step_value"])
print("digest:", d["config"]["secondary_digest"])
print("config keys:", list(d["config"])[:6], "roofline keys:", list(d["roofline"])[:8])
for s in d["secondary"]:
    print(s["name"], s.get("value"), s.get("error"), s.get("leg_seconds"), (s.get("roofline") or {}).get("frac"))
PY
