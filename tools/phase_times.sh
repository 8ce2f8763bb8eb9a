#!/bin/bash
# Run on the GPU box: isolated duration of k_alloc_pack cut short at its stage exits (debug build of the library).
# usage: tools/phase_times.sh   (builds atracdenc_amd/libat3hip_dbg.so with -DAT3HIP_DEBUG_KNOBS)
REPO=$(pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DAT3HIP_DEBUG_KNOBS -o $REPO/gpurun_out/libat3hip_dbg.so \
  $REPO/atracdenc_amd/csrc/at3hip.hip $REPO/atracdenc_amd/csrc/at1hip.hip $REPO/atracdenc_amd/csrc/at3phip.hip $REPO/atracdenc_amd/csrc/at3_tables.cpp 2>/dev/null
export TMPDIR=/tmp
cd /tmp
for STOP in ${STOPS:-1 2 3 4 0}; do
  rm -rf /tmp/ph
  AT3HIP_LIB=$REPO/gpurun_out/libat3hip_dbg.so AT3HIP_DEBUG_STOP=$STOP AT3HIP_DEBUG_GAIN=${GAINSTOP:-0} rocprofv3 --kernel-trace --stats -d /tmp/ph -o ph -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-workloads --sync-steps "$@" > /dev/null 2>&1
  python3 - <<PY
import glob, sqlite3
for f in glob.glob("/tmp/ph/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for name, calls, avg in db.execute("select name, count(*), avg(end-start) from kernels where name like '%${KERNEL:-k_alloc_pack}%' group by name"):
        print("stop=$STOP gain=${GAINSTOP:-0}", name[:30], "avg_us=%.2f" % (avg/1e3))
PY
done
