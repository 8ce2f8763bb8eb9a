#!/bin/bash
# Run on the GPU box: kernel timeline of the bench's pipelined steps - busy fraction (union of kernel intervals) and the
# largest gaps, to see whether the two streams leave the GPU idle anywhere.
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --stats -d /tmp/tl -o tl -- python $REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-side-workloads --regions 0 --no-parity "$@" > /dev/null 2>&1
python3 - <<'PY'
import glob, sqlite3
for f in glob.glob("/tmp/tl/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = list(db.execute("select name, start, end from kernels order by start"))
    ks = [(n.split('(')[0].replace('at3::','').replace('void ','')[:22], s, e) for n, s, e in rows if 'at3::k_' in n]
    print("kernels in trace:", len(rows), "at3:", len(ks))
    if not ks: continue
    # the steady part of the pipelined region: the middle of the kernel list by count (the run ends with isolated steps)
    ks = ks[len(ks) // 4: len(ks) * 6 // 10]
    ivs = sorted((s, e) for _, s, e in ks)
    busy = 0; cur_s, cur_e = ivs[0]
    gaps = []
    for s, e in ivs[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e)); cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = ivs[-1][1] - ivs[0][0]
    print("window %.1f us, busy %.1f %%, kernels %d, sum of durations %.1f us (overlap factor %.2f)" % (span / 1e3, 100.0 * busy / span, len(ks), sum(e - s for _, s, e in ks) / 1e3, sum(e - s for _, s, e in ks) / busy))
    print("largest idle gaps (us):", [round(g / 1e3, 1) for g, _ in sorted(gaps, reverse=True)[:8]], "total idle %.1f us" % (sum(g for g, _ in gaps) / 1e3))
    # one step's order
    first = ks[:30]
    base = first[0][1]
    for n, s, e in first: print("  %-22s start %8.1f  dur %7.1f" % (n, (s - base) / 1e3, (e - s) / 1e3))
PY
