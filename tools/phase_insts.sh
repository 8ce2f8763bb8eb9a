#!/bin/bash
# Run on the GPU box: VALU instructions per wavefront of k_alloc_pack cut short at its stage exits (debug build), from
# rocprofv3 --pmc SQ_INSTS_VALU. usage: STOPS="1 2 4 3 7 8 0" tools/phase_insts.sh
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DAT3HIP_DEBUG_KNOBS -o $REPO/gpurun_out/libat3hip_dbg.so \
  $REPO/atracdenc_amd/csrc/at3hip.hip $REPO/atracdenc_amd/csrc/at1hip.hip $REPO/atracdenc_amd/csrc/at3phip.hip $REPO/atracdenc_amd/csrc/at3_tables.cpp 2>/dev/null
export TMPDIR=/tmp
cd /tmp
for STOP in ${STOPS:-1 2 4 5 6 3 7 8 0}; do
  rm -rf /tmp/pi
  AT3HIP_LIB=$REPO/gpurun_out/libat3hip_dbg.so AT3HIP_DEBUG_STOP=$STOP rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d /tmp/pi -o pi -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-workloads --sync-steps "$@" > /dev/null 2>&1
  python3 - <<PY
import glob, sqlite3
for f in glob.glob("/tmp/pi/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    r = dict((c, v) for c, v in db.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%${KERNEL:-k_alloc_pack}%' group by counter_name"))
    w = r.get("SQ_WAVES", 1) or 1
    print("stop=$STOP", " ".join("%s/wave=%.0f" % (k, v / w) for k, v in sorted(r.items()) if k != "SQ_WAVES"), "waves=%d" % w)
PY
done
