#!/bin/bash
# Run on the GPU box: per-kernel time, every kernel alone (synchronous steps), for several libraries on one box.
# usage: tools/iso_kernels.sh "<bench args>" lib1.so lib2.so ...   (e.g. "--input burst")
ARGS=$1; shift
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
for L in "$@"; do
  OUT=/tmp/iso_$$; rm -rf $OUT; mkdir -p $OUT
  AT3HIP_LIB=$REPO/$L rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-workloads --no-parity --regions 0 --sync-steps $ARGS > $OUT/log 2>&1
  echo "== $L $ARGS"
  python3 $REPO/tools/summarize_prof.py $OUT 2>/dev/null | grep "^k_" | head -14
done
