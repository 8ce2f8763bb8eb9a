#!/bin/bash
# Run on the GPU box: copies and kernels of the host-buffer pipeline on one time axis (rocprofv3 kernel + memory-copy trace).
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
rm -rf /tmp/hp
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/hp -o hp -- python $REPO/tools/host_pipe_probe.py ${1:-8} 2>&1 | tail -2
python3 - <<'PY'
import glob, sqlite3
for f in glob.glob("/tmp/hp/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    ev = []
    for n, s, e in db.execute("select name, start, end from kernels"):
        if 'at3::k_' in n: ev.append((s, e, n.split('(')[0].replace('at3::', '').replace('void ', '')[:20]))
    mc = [t for t in tabs if 'memory_cop' in t.lower()]
    print("memory copy tables:", mc)
    for t in mc[:1]:
        cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
        print(cols)
        q = "select start, end, %s from %s" % ("size" if "size" in cols else cols[-1], t)
        try:
            for s, e, sz in db.execute(q):
                ev.append((s, e, "COPY %s B" % sz))
        except Exception as ex:
            print("copy query failed:", ex)
    ev.sort()
    if not ev: continue
    copies = [(s, e) for s, e, n in ev if n.startswith("COPY 33554432")]
    print("H2D copies: start-to-start periods (us):", [round((copies[i + 1][0] - copies[i][0]) / 1e3) for i in range(len(copies) - 1)])
    print("H2D copy durations (us):", [round((e - s) / 1e3) for s, e in copies])
    ev = ev[len(ev) // 2:]
    base = ev[0][0]
    for s, e, n in ev[:int(__import__("os").environ.get("TRACE_ROWS", "70"))]:
        print("%10.1f %10.1f  %s" % ((s - base) / 1e3, (e - s) / 1e3, n))
PY
