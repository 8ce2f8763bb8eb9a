#!/bin/bash
# Run on the GPU box: start and end of every kernel over two steady periods of the pipelined bench, with the hardware queue each ran on:
# which kernels of which stage wait for which. usage: tools/timeline2.sh [bench args]
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --stats -d /tmp/tl -o tl -- python $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-side-workloads --regions 0 --no-parity "$@" > /dev/null 2>&1
python3 - <<'PY'
import glob, sqlite3
for f in glob.glob("/tmp/tl/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    q = "select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")
    rows = list(db.execute(q))
    ks = [(r[0].split('(')[0].replace('at3::','').replace('void ','')[:22], r[1], r[2], r[3] if qcol else 0) for r in rows if 'at3::k_' in r[0]]
    if not ks: continue
    # steady part: find k_alloc_pack starts; take from the 5th to the 7th
    ap = [s for n, s, e, qq in ks if n.startswith("k_alloc_pack")]
    if len(ap) < 9: continue
    lo, hi = ap[5], ap[7]
    print("period: %.1f us (between k_alloc_pack starts)" % ((ap[7] - ap[5]) / 2e3), "columns:", cols)
    qs = sorted(set(qq for n, s, e, qq in ks))
    for n, s, e, qq in ks:
        if s >= lo - 150e3 and s < hi:
            print("  q%d %s%-22s %8.1f -> %8.1f  (%6.1f)" % (qs.index(qq), "    " * qs.index(qq), n, (s - lo) / 1e3, (e - lo) / 1e3, (e - s) / 1e3))
PY
