#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats + PMC passes) into a small text/JSON summary for profiles/."""
import csv, glob, json, os, sys

out = sys.argv[1]
res = {}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", f)
    for row in csv.DictReader(open(f)):
        print("  ", {k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
        if "ekf_step_kernel" in row.get("Name", ""):
            res["kernel"] = row["Name"]; res["calls"] = int(row["Calls"]); res["avg_ns"] = float(row["AverageNs"])
import statistics
for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if "ekf_step_kernel" not in row.get("Kernel_Name", ""):
                continue
            key = (row["Counter_Name"], row["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
        names = sorted({k[0] for k in per})
        for name in names:
            vals = [v for (n, _), v in per.items() if n == name]
            # median over dispatches: the steady-state launches (the first, state-growing launch is an outlier)
            res[name] = statistics.median(vals)
            print(f"== {tag}: {name} median per dispatch = {res[name]:.6g}  (dispatches {len(vals)}, mean {sum(vals) / len(vals):.6g})")
# FETCH_SIZE / WRITE_SIZE are in KiB... the guide: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, FETCH_SIZE reads 1/2 on gfx950 wide streams
if "FETCH_SIZE" in res or "WRITE_SIZE" in res:
    # gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports 1/2 of a coalesced stream (tools/calib_copy.hip
    # confirms: 4 GiB copy -> 2 GiB FETCH, 4 GiB WRITE; a strided 8-byte gather counts 64 B per touched line).
    fb = res.get("FETCH_SIZE", 0.0) * 1024 * 2.0
    wb = res.get("WRITE_SIZE", 0.0) * 1024
    res["hbm_read_bytes_per_launch_corrected"] = fb; res["hbm_write_bytes_per_launch"] = wb
    res["hbm_bytes_per_launch"] = fb + wb
    print(f"== HBM bytes per launch (FETCH x2 corrected + WRITE): {fb + wb:.6g}  read {fb:.6g} write {wb:.6g}")
res["batch"] = 65536; res["landmarks"] = 50
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
