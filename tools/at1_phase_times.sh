#!/bin/bash
# Run on the GPU box: k_at1_front cut short at its stage exits (debug build of the library, AT1HIP_DEBUG_STOP).
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DAT3HIP_DEBUG_KNOBS -o $REPO/gpurun_out/libat3hip_dbg.so \
  $REPO/atracdenc_amd/csrc/at3hip.hip $REPO/atracdenc_amd/csrc/at1hip.hip $REPO/atracdenc_amd/csrc/at3phip.hip $REPO/atracdenc_amd/csrc/at3_tables.cpp 2>/dev/null
export TMPDIR=/tmp
cd /tmp
for STOP in ${STOPS:-1 2 3 4 5 6 7 0}; do
  rm -rf /tmp/ph1
  AT3HIP_LIB=$REPO/gpurun_out/libat3hip_dbg.so AT1HIP_DEBUG_STOP=$STOP rocprofv3 --kernel-trace --stats -d /tmp/ph1 -o ph -- python $REPO/tools/at1_bench.py --steps 3 --warmup 1 "$@" > /dev/null 2>&1
  python3 - <<PY
import glob, sqlite3
for f in glob.glob("/tmp/ph1/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for name, calls, avg in db.execute("select name, count(*), avg(end-start) from kernels where name like '%k_at1_front%' group by name"):
        print("stop=$STOP", "avg_us=%.2f" % (avg/1e3))
PY
done
