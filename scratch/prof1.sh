cd /tmp && export TMPDIR=/tmp
export SLAM_PGS_CHOL_THREADS=1024
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pgs1024 -o p -- python3 tools/pgs_stream_table.py --graphs 1024 --slots 0 --groups 2 --solves 3 > gpurun_out/prof_pgs1024.log 2>&1
grep -v "^#   group" gpurun_out/prof_pgs1024.log | tail -3
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_pgs1024/**/*kernel_stats.csv',recursive=True)
print(f)
rows=list(csv.DictReader(open(f[0])))
for r in rows[:20]:
    print(r['Name'][:60].ljust(60), r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
