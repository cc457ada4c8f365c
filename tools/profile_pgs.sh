#!/bin/bash
# Profile bench.py --filter pgs on the GPU box: kernel-trace stats, then PMC passes (separate runs).
# Usage (on the box, from the repo root): bash tools/profile_pgs.sh <tag> [extra bench args]  -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-pgs}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --filter pgs --no-cpu-baseline --no-batch-256 --steps 2 --warmup 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o pmc -- python3 $ARGS > $OUT/bench_pmc_mfma.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/bench_pmc_write.log 2>&1
tail -1 $OUT/bench_stats.log
find $OUT -name "*kernel_stats.csv" | head -3
cat $(find $OUT -name "*kernel_stats.csv" | head -1) | head -20
python3 - <<PY
import csv,glob,collections
for tag in ("pmc_mfma","pmc_fetch","pmc_write"):
    fs=glob.glob("$OUT/%s/**/*counter_collection.csv"%tag, recursive=True)
    if not fs: print(tag,"no csv"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        import re
        m_=re.search(r"(pgs_\w+|__amd_\w+)", r["Kernel_Name"]); k=m_.group(1) if m_ else r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    for k,v in acc.items():
        print(tag,k,dict(v))
PY
