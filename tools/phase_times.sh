#!/bin/bash
# usage (on the GPU box): tools/phase_times.sh <kernel-substring> "<stops>" [bench args] - kernel time per AT3HIP_DEBUG_STOP value
K=$1; STOPS=$2; shift 2
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/phase
mkdir -p $OUT
cd /tmp
for st in $STOPS; do
  rm -rf $OUT/s$st
  AT3HIP_DEBUG_STOP=$st rocprofv3 --kernel-trace -d $OUT/s$st -o t -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/run$st.log 2>&1
  python3 - <<PY
import glob, sqlite3
for f in glob.glob("$OUT/s$st/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for name, calls, avg in db.execute("select name, count(*), avg(end-start) from kernels where name like '%$K%' group by name"):
        print("stop=$st", name[:50], calls, round(avg/1e3, 2))
PY
done
