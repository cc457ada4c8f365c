# Issue / wait counters of the UKF kernels (rocprofv3 --pmc, one pass per counter group), current build.
# Usage on the GPU box: bash tools/pmc_ukf.sh  -> gpurun_out/prof_ukf_pmc/
set -u
OUT=gpurun_out/prof_ukf_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SLAM_UKF_SPLIT_MIN=100000000   # single stream: full-batch launches
ARGS="bench.py --filter ukf --batch 4096 --landmarks 20 --no-cpu-baseline --steps 40 --warmup 5"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $C -d $OUT/p$i -o pmc -- python3 $ARGS > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
for i in (1,2,3,4):
    fs=glob.glob("$OUT/p%d/**/*counter_collection.csv"%i, recursive=True)
    if not fs: print("pass",i,"no csv:", open("$OUT/log%d.txt"%i).read()[-300:]); continue
    for r in csv.DictReader(open(fs[0])):
        m=re.search(r"(ukf_\w+_kernel)", r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    print(k)
    for c in sorted(v): print("   %-24s %.4g"%(c,v[c]))
    W=v.get("SQ_WAVE_CYCLES",0)
    if W:
        for c in ("SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_ACTIVE_INST_SCA","SQ_ACTIVE_INST_ANY","SQ_WAIT_INST_ANY","SQ_WAIT_INST_LDS","SQ_WAIT_ANY"):
            if c in v: print("   %-24s / SQ_WAVE_CYCLES = %.3f"%(c, v[c]/W))
    if v.get("SQ_LDS_IDX_ACTIVE"): print("   bank-conflict share of LDS cycles = %.3f"%(v["SQ_LDS_BANK_CONFLICT"]/v["SQ_LDS_IDX_ACTIVE"]))
    t=v.get("SQ_INSTS_VALU",0)+v.get("SQ_INSTS_SALU",0)+v.get("SQ_INSTS_LDS",0)
    if t: print("   instruction mix VALU/SALU/LDS = %.2f / %.2f / %.2f"%(v["SQ_INSTS_VALU"]/t, v["SQ_INSTS_SALU"]/t, v["SQ_INSTS_LDS"]/t))
PY
