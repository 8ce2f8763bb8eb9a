#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes for the bench workload.
# Outputs under gpurun_out/prof_<tag>/ ; summaries are copied to profiles/ by hand afterwards.
set -x
TAG=${1:-r01}
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/pmc_sq -o sq -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_sq.log 2>&1
find $OUT -name "*.csv" | head -50
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
