cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for H in 1 4; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_conc$H -o p -- python3 scratch/two.py $H 800 860 > gpurun_out/prof_conc$H.log 2>&1
tail -1 gpurun_out/prof_conc$H.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_conc$H/p_kernel_trace.csv')))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in rows if 'pgs_' in r['Kernel_Name'] and 'run_sim' not in r['Kernel_Name']]
ev.sort()
# only the window of the every-iteration run: last 90% of events
s=sum(e-b for b,e,_ in ev)
pts=sorted([(b,1) for b,e,_ in ev]+[(e,-1) for b,e,_ in ev])
busy=0;cur=0;last=None
for t,d in pts:
    if cur>0: busy+=t-last
    cur+=d; last=t
print("H=$H kernels", len(ev), "sum of durations ms", s/1e6, "union busy ms", busy/1e6, "mean concurrency", s/busy, "span ms", (ev[-1][1]-ev[0][0])/1e6)
PY
rm -rf gpurun_out/prof_conc$H
done
