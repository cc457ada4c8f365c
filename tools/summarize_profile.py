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
for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        acc, cnt = {}, {}
        for row in csv.DictReader(open(f)):
            if "ekf_step_kernel" not in row.get("Kernel_Name", ""):
                continue
            name, val = row["Counter_Name"], float(row["Counter_Value"])
            acc[name] = acc.get(name, 0.0) + val; cnt[name] = cnt.get(name, 0) + 1
        for name in acc:
            # dispatch rows may be split per XCD/SE; report the per-dispatch mean of the summed rows
            disp = len({r["Dispatch_Id"] for r in csv.DictReader(open(f)) if "ekf_step_kernel" in r.get("Kernel_Name", "") and r["Counter_Name"] == name})
            res[name] = acc[name] / max(disp, 1)
            print(f"== {tag}: {name} per dispatch = {res[name]:.6g}  (rows {cnt[name]}, dispatches {disp})")
# FETCH_SIZE / WRITE_SIZE are in KiB... the guide: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, FETCH_SIZE reads 1/2 on gfx950 wide streams
if "FETCH_SIZE" in res or "WRITE_SIZE" in res:
    fb = res.get("FETCH_SIZE", 0.0) * 1024 * 2.0   # gfx950 correction (MI355X_MICROARCH.md §HBM): x2 for 16 B/lane streams
    wb = res.get("WRITE_SIZE", 0.0) * 1024
    res["hbm_read_bytes_per_launch_corrected"] = fb; res["hbm_write_bytes_per_launch"] = wb
    res["hbm_bytes_per_launch"] = fb + wb
    print(f"== HBM bytes per launch (FETCH x2 corrected + WRITE): {fb + wb:.6g}  read {fb:.6g} write {wb:.6g}")
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
