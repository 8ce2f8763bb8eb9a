#!/bin/bash
# usage (on the GPU box): tools/pmc_custom.sh <kernel-substring> "<counters of pass 1>" ["<pass 2>" ...]
K=$1; shift
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmccustom
mkdir -p $OUT
cd /tmp
i=0
for SET in "$@"; do
  i=$((i+1)); rm -rf $OUT/p$i
  rocprofv3 --pmc $SET --kernel-trace -d $OUT/p$i -o sq -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/run.log 2>&1
done
python3 - <<PY
import glob, sqlite3
vals = {}
for f in glob.glob("$OUT/p*/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for k, c, v in db.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%$K%' group by kernel_name, counter_name"):
        vals[c] = v
print(" ".join(f"{c}={v:.4g}" for c, v in sorted(vals.items())))
PY
