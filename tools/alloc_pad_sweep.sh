#!/bin/bash
# Run on the GPU box: k_alloc_pack's isolated duration against the dynamic-LDS pad that sets its workgroups per CU.
REPO=$(pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DAT3HIP_DEBUG_KNOBS -o /tmp/libat3hip_dbg.so \
  $REPO/atracdenc_amd/csrc/at3hip.hip $REPO/atracdenc_amd/csrc/at1hip.hip $REPO/atracdenc_amd/csrc/at3phip.hip $REPO/atracdenc_amd/csrc/at3_tables.cpp 2>/dev/null
export TMPDIR=/tmp
cd /tmp
for PAD in ${PADS:-0 512 1024 2048 3072 4096}; do
  rm -rf /tmp/ph
  AT3HIP_LIB=/tmp/libat3hip_dbg.so AT3HIP_ALLOC_PAD=$PAD AT3HIP_DEBUG_STOP=${STOP:-0} rocprofv3 --kernel-trace --stats -d /tmp/ph -o ph -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-workloads --sync-steps "$@" > /dev/null 2>&1
  python3 - <<PY
import glob, sqlite3
for f in glob.glob("/tmp/ph/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for name, calls, avg in db.execute("select name, count(*), avg(end-start) from kernels where name like '%k_alloc_pack%' group by name"):
        print("pad=$PAD", "avg_us=%.2f" % (avg/1e3))
PY
  AT3HIP_LIB=/tmp/libat3hip_dbg.so AT3HIP_ALLOC_PAD=$PAD python $REPO/bench.py --no-side-workloads --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  pipelined ms_per_step', d['ms_per_step'])"
done
