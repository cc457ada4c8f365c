cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pgsit -o p -- python3 bench.py --filter pgs --iterative --batch 256 --no-cpu-baseline --no-parity-check > gpurun_out/prof_pgsit.log 2>&1
grep -o "\"value\": [0-9.]*\|lm_trials_launched_per_tick\": [0-9.]*" gpurun_out/prof_pgsit.log | tr '\n' ' '; echo
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_pgsit/p_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:18]:
    print(r['Name'].split('(')[0][-44:].ljust(44), r['Calls'].rjust(7), f"{float(r['TotalDurationNs'])/1e6:9.1f} ms", f"{float(r['AverageNs'])/1e3:8.1f} us", r['Percentage'])
PY
rm -f gpurun_out/prof_pgsit/p_kernel_trace.csv
