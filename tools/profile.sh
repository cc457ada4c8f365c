#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then PMC passes (separate runs, as the guide prescribes; the program
# itself follows `--`).  Usage (on the box, from the repo root): bash tools/profile.sh <tag> [K W] [extra bench args]
#   -> gpurun_out/prof_<tag>/ ; copy summary.json / summary.txt / kernel_stats.csv into profiles/<tag>/ afterwards.
set -u
TAG=${1:-r03}
K=${2:-20}
W=${3:-5}
shift $(( $# < 3 ? $# : 3 ))   # (a plain `shift 3` with fewer arguments fails and keeps them all: ADVICE r02)
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps $K --warmup $W --no-cpu-baseline --no-long-runs --no-parity-check --no-once-per-step --no-secondary $*"
python3 $ARGS > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err || { echo "unprofiled bench run failed:"; tail -5 $OUT/bench_unprofiled.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $ARGS > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $ARGS > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmc_sq -o pmc -- python3 $ARGS > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc_lds -o pmc -- python3 $ARGS > $OUT/bench_pmc_lds.log 2>&1
python3 tools/summarize_profile.py $OUT $K $W > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
