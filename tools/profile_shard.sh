#!/bin/bash
# Run on the GPU box: per-kernel isolated times (synchronous steps) at the per-GPU shard of BASELINE configs[2].
# usage: tools/profile_shard.sh <tag> [bench args]  -> gpurun_out/prof_<tag>_shard/summary_isolated.txt
TAG=${1:-r02}; shift
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_${TAG}_shard
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --streams 1024 --frames 128 --steps 3 --warmup 1 --no-cpu-baseline --no-side-workloads --sync-steps "$@" > $OUT/bench_stats_sync.log 2>&1
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary_isolated.txt 2>&1
rm -rf $OUT/stats
head -20 $OUT/summary_isolated.txt
