#!/bin/bash
# GPU box: rocprofv3 --kernel-trace durations of the fused kernel alone (tools/k1/lab.py) at both sizes: all launches, and the second half of them
# (the first launches of a burst run in the power manager's transient: slower). usage: tools/k1/rocprof_lab.sh lib.so [lab args]
LIB=$1; shift
export TMPDIR=/tmp
REPO=$(pwd)
for SIZE in 64x64 1024x128; do
  rm -rf /tmp/k1rp
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/k1rp -o p -- python $REPO/tools/k1/lab.py --reps 1 --sizes $SIZE "$@" $REPO/$LIB > /tmp/k1rp.log 2>&1)
  python3 - "$SIZE" <<'PY'
import glob, sqlite3, sys, statistics
size = sys.argv[1]
S, F = (int(x) for x in size.split("x"))
for f in glob.glob("/tmp/k1rp/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    d = [r[0] / 1e3 for r in db.execute("select end - start from kernels where name like '%qmf_mdct8%' order by start")]
    if not d: continue
    h = d[len(d) // 2:]
    alg = S * F * 16384
    print("k_qmf_mdct8 %s: %d launches, mean %.2f us; second half: mean %.2f median %.2f min %.2f us -> %.3f of 8 TB/s (median), %.3f (min)" % (
        size, len(d), statistics.mean(d), statistics.mean(h), statistics.median(h), min(h), alg / (statistics.median(h) * 1e-6) / 8e12, alg / (min(h) * 1e-6) / 8e12))
    for r in db.execute("select distinct workgroup_x, grid_x, lds_size, scratch_size, vgpr_count, sgpr_count from kernels where name like '%qmf_mdct8%'"):
        print("   wg=%d grid=%d lds=%d scratch=%d vgpr=%d sgpr=%d" % r)
PY
done
