#!/bin/bash
# Run on the GPU box: isolated per-kernel durations (rocprofv3 --kernel-trace, bench.py --sync-steps) for each library
# given on the command line, alternating, so that two builds are compared on the same box.
# usage: tools/ab_kernels.sh libA.so libB.so [-- bench.py flags]      (SYNC= tools/ab_kernels.sh ... : pipelined steps instead)
REPO=$(pwd)
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" = "--" ] && shift
export TMPDIR=/tmp
cd /tmp
for rep in 1 2; do
for L in "${LIBS[@]}"; do
  rm -rf /tmp/abk
  AT3HIP_LIB=$REPO/$L rocprofv3 --kernel-trace --stats -d /tmp/abk -o abk -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-workloads ${SYNC---sync-steps} "$@" > /dev/null 2>&1
  python3 - <<PY
import glob, sqlite3
out = []
for f in glob.glob("/tmp/abk/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    for name, calls, avg in db.execute("select name, count(*), avg(end-start) from kernels where name like '%at3::k_%' group by name"):
        out.append((name.split('(')[0].replace('at3::','').replace('void ','')[:18], avg/1e3))
print("$L".split('/')[-1][:24].ljust(24), " ".join("%s=%.1f" % kv for kv in sorted(out)), "sum=%.1f" % sum(v for _, v in out))
PY
done; done
