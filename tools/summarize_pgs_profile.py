#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_pgs.sh (bench.py --filter pgs --steps 2 --warmup 1 = 4 solves per
run) into profiles/<tag>/: kernel_stats.csv (copied), summary.json / summary.txt (per-kernel time per solve, PMC sums).
Usage: summarize_pgs_profile.py gpurun_out/prof_<tag> profiles/<tag>"""
import csv, glob, json, os, re, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
SOLVES = 4   # 1 warm-up + 2 timed + 1 per-kernel-timed solve
res = {"solves_profiled": SOLVES, "kernels": {}}
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
    for row in csv.DictReader(open(f)):
        m = re.search(r"(pgs_\w+(?:<[^>]*>)?)", row["Name"])
        if not m:
            continue
        res["kernels"][m.group(1)] = {"calls": int(row["Calls"]), "total_ms": float(row["TotalDurationNs"]) / 1e6,
                                      "avg_us": float(row["AverageNs"]) / 1e3, "min_us": float(row["MinNs"]) / 1e3,
                                      "max_us": float(row["MaxNs"]) / 1e3, "ms_per_solve": float(row["TotalDurationNs"]) / 1e6 / SOLVES}
for tag in ("pmc_mfma", "pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(src, tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(pgs_\w+(?:<[^>]*>)?)", row["Kernel_Name"])
            if not m or m.group(1) not in res["kernels"]:
                continue
            k = res["kernels"][m.group(1)].setdefault("pmc", {})
            k[row["Counter_Name"]] = k.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for name, k in res["kernels"].items():
    pmc = k.get("pmc", {})
    if "FETCH_SIZE" in pmc:   # KiB at the L2 <-> fabric boundary; x2 applies to wide coalesced streams only (guide), so raw here
        k["fetch_GB_per_solve_raw"] = pmc["FETCH_SIZE"] * 1024 / 1e9 / SOLVES
    if "WRITE_SIZE" in pmc:
        k["write_GB_per_solve"] = pmc["WRITE_SIZE"] * 1024 / 1e9 / SOLVES
    if pmc.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
        k["mfma_f64_flop_per_solve"] = pmc["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 / SOLVES   # 1 MOP = 512 FLOP
        k["mfma_executed_TFLOPs"] = k["mfma_f64_flop_per_solve"] / (k["ms_per_solve"] * 1e-3) / 1e12
for f in glob.glob(os.path.join(src, "bench_line*.json")):
    try:
        res[os.path.basename(f)[:-5]] = json.loads(open(f).read().strip().splitlines()[-1])
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
    except Exception as e:
        print("skip", f, e)
json.dump(res, open(os.path.join(dst, "summary.json"), "w"), indent=1)
with open(os.path.join(dst, "summary.txt"), "w") as o:
    o.write(f"rocprofv3 --kernel-trace --stats + PMC passes of `python bench.py --filter pgs --steps 2 --warmup 1` ({SOLVES} solves)\n")
    o.write("kernel                      calls  avg_us   min_us   max_us  ms/solve  fetch GB/solve(raw)  write GB/solve  MFMA TF executed\n")
    for name, k in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
        o.write(f"{name:26s} {k['calls']:6d} {k['avg_us']:8.1f} {k['min_us']:8.1f} {k['max_us']:8.1f} {k['ms_per_solve']:8.2f} "
                f"{k.get('fetch_GB_per_solve_raw', 0):12.2f} {k.get('write_GB_per_solve', 0):15.2f} {k.get('mfma_executed_TFLOPs', 0):12.1f}\n")
print(open(os.path.join(dst, "summary.txt")).read())
