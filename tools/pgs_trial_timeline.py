#!/usr/bin/env python3
"""Per-trial kernel durations of the per-kernel-timed solve in a rocprofv3 --kernel-trace of `bench.py --filter pgs` (the solve that runs as ONE
group on ONE stream: the stream with ~ twelve kernels x 40 trials and nothing else).  usage: pgs_trial_timeline.py kernel_trace.csv > trial_timeline.txt"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").replace("slam::", "").split("(")[0].split("<")[0]
by = collections.defaultdict(list)
for r in rows:
    if "pgs_" in r["Kernel_Name"]: by[(r["Queue_Id"], r["Stream_Id"])].append(r)
# the profiled solve: the (queue, stream) with the fewest dispatches that still holds whole trials (the two group streams hold the other three solves)
cands = [(len(v), k) for k, v in by.items() if sum(1 for r in v if "pgs_chol" in r["Kernel_Name"]) >= 10]
cands.sort()
seq = sorted(by[cands[0][1]], key=lambda r: int(r["Start_Timestamp"]))
order = ["pgs_lin_factor_kernel", "pgs_linearize_kernel", "pgs_seg_chain_kernel", "pgs_seg_kernel", "pgs_sep_kernel", "pgs_seg_gram_kernel", "pgs_syrk_kernel",
         "pgs_chol_ll_kernel", "pgs_seg_backsolve_lds_kernel", "pgs_eval_factor_kernel", "pgs_evaluate_kernel", "pgs_decide_kernel"]
trials, cur = [], {}
for r in seq:
    n = short(r["Kernel_Name"])
    if n not in order: continue
    if n in cur or (n == "pgs_lin_factor_kernel" and cur): trials.append(cur); cur = {}
    cur[n] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    if n == "pgs_seg_kernel": cur["_wg"] = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
if cur: trials.append(cur)
allk = collections.defaultdict(list)
for r in rows:
    n = short(r["Kernel_Name"])
    if n in order: allk[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
print(f"Per-trial kernel durations (us) of ONE batch-256 solve of BASELINE configs[4] (1000 poses x 200 landmarks): the dispatches of the per-kernel-timed solve")
print(f"(one group, one stream: queue/stream {cands[0][1]}, {len(trials)} trials) in a rocprofv3 --kernel-trace of `bench.py --filter pgs --steps 2 --warmup 1`.")
print("first column = average over all dispatches of the run (all solves, both groups).\n")
for n in order:
    print(f"{n:32s} avg {sum(allk[n]) / max(len(allk[n]), 1):6.1f} | " + " ".join(f"{t.get(n, 0):.0f}" for t in trials))
print(f"{'sum of the kernels of a trial':43s} | " + " ".join(f"{sum(v for k, v in t.items() if k != '_wg'):.0f}" for t in trials))
print("\nworkgroups of pgs_seg_kernel per trial (= running slots x segments): " + " ".join(str(t.get("_wg", 0)) for t in trials))
