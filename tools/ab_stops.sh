#!/bin/bash
# Run on the GPU box: k_alloc_pack's isolated duration at its stage exits, for each debug library given (same box).
# usage: STOPS="1 2 3 0" tools/ab_stops.sh libA_dbg.so libB_dbg.so
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
for STOP in ${STOPS:-1 2 3 0}; do
for L in "$@"; do
  rm -rf /tmp/ph
  AT3HIP_LIB=$REPO/$L AT3HIP_DEBUG_STOP=$STOP rocprofv3 --kernel-trace --stats -d /tmp/ph -o ph -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-workloads --sync-steps > /dev/null 2>&1
  python3 - <<PY
import glob, sqlite3
for f in glob.glob("/tmp/ph/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for name, calls, avg in db.execute("select name, count(*), avg(end-start) from kernels where name like '%${KERNEL:-k_alloc_pack}%' group by name"):
        print("stop=$STOP", "$L".split('/')[-1], "avg_us=%.2f" % (avg/1e3))
PY
done; done
