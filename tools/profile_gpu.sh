#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes for the bench workload.
# Outputs under gpurun_out/prof_<tag>/ ; summaries are copied to profiles/ by hand afterwards.
set -x
TAG=${1:-r01}
# BENCH_ARGS: extra bench.py arguments for every pass (e.g. --no-gain for the fused k_qmf_mdct8 kernel)
BENCH_ARGS=${BENCH_ARGS:-}
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-workloads --regions 0 --no-parity $BENCH_ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-workloads --regions 0 --no-parity $BENCH_ARGS --sync-steps > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-workloads --regions 0 --no-parity $BENCH_ARGS --sync-steps > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/pmc_sq -o sq -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-workloads --regions 0 --no-parity $BENCH_ARGS --sync-steps > $OUT/bench_sq.log 2>&1
find $OUT -name "*.csv" | head -50
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
# the same command with synchronous steps: every kernel alone on the GPU (isolated durations)
rm -rf $OUT/stats
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-workloads --regions 0 --no-parity $BENCH_ARGS --sync-steps > $OUT/bench_stats_sync.log 2>&1
python3 - <<PY > $OUT/summary_isolated.txt
import glob, sqlite3
print("== kernel stats, synchronous steps (every kernel alone): rocprofv3 --kernel-trace --stats ==")
for f in glob.glob("$OUT/stats/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    for name, calls, total, avg in rows:
        n = name.split("(")[0].replace("void ", "").replace("at3::", "")[:48]
        print(f"{n:48s} {calls:6d} {total/1e3:12.1f} {avg/1e3:10.2f} {100.0*total/tot:6.2f}")
PY
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
cat $OUT/summary.txt
