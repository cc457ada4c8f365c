# Cache / wait counters of the pose-graph kernels (rocprofv3 --pmc, one pass per counter group): L1 and L2 hit rates,
# L1-miss latency, wait fraction.  Usage on the GPU box: bash tools/pmc_pgs_cache.sh  -> gpurun_out/prof_syrk_diag/
set -u
OUT=gpurun_out/prof_syrk_diag; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --filter pgs --no-cpu-baseline --steps 1 --warmup 1"
i=0
for C in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $C -d $OUT/p$i -o pmc -- python3 $ARGS > $OUT/log$i.txt 2>&1
  python3 - <<PY
import csv,glob,collections,re
fs=glob.glob("$OUT/p$i/**/*counter_collection.csv", recursive=True)
if not fs: print("pass $i ($C): no csv"); print(open("$OUT/log$i.txt").read()[-400:])
else:
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        m_=re.search(r"(pgs_\w+)", r["Kernel_Name"]);
        if m_: acc[m_.group(1)][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k in ("pgs_syrk_kernel","pgs_chain_kernel","pgs_chol_kernel"):
        print(k, dict(acc[k]))
PY
done
