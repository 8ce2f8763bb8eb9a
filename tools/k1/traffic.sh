#!/bin/bash
# GPU box: HBM-side bytes of the fused kernel alone (tools/k1/lab.py, one size): FETCH_SIZE and WRITE_SIZE in separate passes.
# FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950 (128-byte requests
# tallied at 64 B); both counters are in KB. usage: tools/k1/traffic.sh <size> lib.so [lib2.so ...]
SIZE=$1; shift
export TMPDIR=/tmp
REPO=$(pwd)
for LIB in "$@"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/k1tr_$C
    (cd /tmp && rocprofv3 --pmc $C --kernel-trace -d /tmp/k1tr_$C -o p -- python $REPO/tools/k1/lab.py --reps 1 --sizes $SIZE $REPO/$LIB > /tmp/k1tr_$C.log 2>&1)
  done
  python3 - "$LIB" "$SIZE" <<'PY'
import glob, sqlite3, sys
lib, size = sys.argv[1], sys.argv[2]
S, F = (int(x) for x in size.split("x"))
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("/tmp/k1tr_%s/**/*.db" % c, recursive=True):
        db = sqlite3.connect(f)
        for k, n, v, cnt in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%qmf_mdct8%' group by kernel_name, counter_name"):
            out[n] = (v, cnt)
alg = S * F * 16384
fetch = out.get("FETCH_SIZE", (0, 0))[0] * 1024 * 2
write = out.get("WRITE_SIZE", (0, 0))[0] * 1024
print("%s %s: fetch (x2 corrected) %.1f MB, write %.1f MB, sum %.1f MB = %.2f x the algorithmic %.1f MB   (%d dispatches)" % (
    lib, size, fetch / 1e6, write / 1e6, (fetch + write) / 1e6, (fetch + write) / alg, alg / 1e6, out.get("FETCH_SIZE", (0, 0))[1]))
PY
done
