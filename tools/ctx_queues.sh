#!/bin/bash
# GPU box: which hardware queue the kernels of the first, second, third context of a process run on (rocprofv3 --kernel-trace of tools/ctx_order_probe.py).
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp; rm -rf /tmp/cq
rocprofv3 --kernel-trace -d /tmp/cq -o cq -- python $REPO/tools/ctx_order_probe.py "$@" > /tmp/cq.log 2>&1
grep context /tmp/cq.log
python3 - <<'PY'
import glob, sqlite3, collections
for f in glob.glob("/tmp/cq/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = list(db.execute("select name, start, end, queue_id, stream_id from kernels order by start"))
    ks = [(r[0].split('(')[0].replace('at3::', '').replace('void ', '')[:20], r[1], r[2], r[3], r[4]) for r in rows if 'at3::k_' in r[0]]
    if not ks: continue
    # contexts are separated by long gaps without at3 kernels (context creation): split where the gap exceeds 20 ms
    groups, cur, last = [], [], None
    for k in ks:
        if last is not None and k[1] - last > 20e6:
            groups.append(cur); cur = []
        cur.append(k); last = k[2]
    groups.append(cur)
    for gi, g in enumerate(groups):
        per = collections.defaultdict(set)
        for n, s, e, q, st in g: per[n].add((q, st))
        print("context", gi + 1, {n: sorted(v) for n, v in per.items() if n in ("k_qmf_sub8", "k_gain_curve", "k_alloc_pack", "k_gain_analysis")})
PY
