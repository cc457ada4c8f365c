#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats + PMC passes) of `bench.py --steps K --warmup W` into a small text/JSON
summary for profiles/.  Usage: summarize_profile.py <dir> [K W].

bench.py issues: 1 + PRE single-step launches (ekf_step_kernel<...,false>), then two multi-step launches
(ekf_step_kernel<...,true>): W warm-up timesteps and the K timed timesteps.  Per-timestep figures of the multi-step
kernel = its totals / (W + K); PMC counters are taken from the dispatch with the largest value (the K-step launch)."""
import csv, glob, json, os, statistics, sys

out = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
res = {"steps": K, "warmup": W}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", f)
    for row in csv.DictReader(open(f)):
        print("  ", {k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
        nm = row.get("Name", "")
        if "ekf_step_kernel" in nm and "Lb1" in nm or ("ekf_step_kernel" in nm and ", true>" in nm):
            res["kernel"] = nm; res["calls"] = int(row["Calls"]); res["avg_ns"] = float(row["AverageNs"])
            res["total_ns"] = float(row["TotalDurationNs"]); res["max_ns"] = float(row["MaxNs"])
            res["ns_per_timestep"] = res["total_ns"] / (K + W)
            print(f"== multi-step kernel: {res['calls']} launches ({W} + {K} timesteps), {res['total_ns'] / 1e6:.3f} ms total "
                  f"-> {res['ns_per_timestep'] / 1e6:.4f} ms per timestep; the {K}-step launch alone: {res['max_ns'] / 1e6:.3f} ms "
                  f"= {res['max_ns'] / K / 1e6:.4f} ms per timestep")
        elif "ekf_step_kernel" in nm:
            res["single_step_kernel"] = nm; res["single_step_calls"] = int(row["Calls"]); res["single_step_avg_ns"] = float(row["AverageNs"])
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
            res[name] = max(vals) / K       # the K-step launch, per timestep
            print(f"== {tag}: {name} of the {K}-step launch / {K} = {res[name]:.6g} per timestep  (dispatches {len(vals)}, "
                  f"median {statistics.median(vals):.6g})")
if "FETCH_SIZE" in res or "WRITE_SIZE" in res:
    # FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2
    # of a coalesced 16 B/lane stream (tools/calib_copy.hip confirms: 4 GiB copy -> 2 GiB FETCH, 4 GiB WRITE; a strided
    # 8-byte gather counts 64 B per touched line).
    fb = res.get("FETCH_SIZE", 0.0) * 1024 * 2.0
    wb = res.get("WRITE_SIZE", 0.0) * 1024
    res["hbm_read_bytes_per_step_corrected"] = fb; res["hbm_write_bytes_per_step"] = wb
    res["hbm_bytes_per_step"] = fb + wb
    print(f"== L2<->fabric bytes per timestep (FETCH x2 corrected + WRITE): {fb + wb:.6g}  read {fb:.6g} write {wb:.6g}")
res["batch"] = 65536; res["landmarks"] = 50
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
