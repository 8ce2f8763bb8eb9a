#!/bin/bash
# Run on the GPU box (via gpurun): per-kernel time of the bench workload for each SURVEY 8(d) input.
# usage: tools/profile_inputs.sh <tag> [inputs...]   -> gpurun_out/prof_<tag>_<input>/summary.txt
TAG=${1:-r02}; shift
INPUTS=${@:-noise burst tones}
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
for IN in $INPUTS; do
  OUT=$REPO/gpurun_out/prof_${TAG}_$IN
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --steps 5 --warmup 2 --input $IN --no-cpu-baseline --no-side-workloads $EXTRA > $OUT/bench_stats.log 2>&1
  python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
  rm -rf $OUT/stats
  # the same with synchronous steps: every kernel alone on the GPU
  rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --steps 5 --warmup 2 --input $IN --no-cpu-baseline --no-side-workloads --sync-steps $EXTRA > $OUT/bench_stats_sync.log 2>&1
  python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary_isolated.txt 2>&1
  rm -rf $OUT/stats
done
