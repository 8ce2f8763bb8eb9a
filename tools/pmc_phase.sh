#!/bin/bash
# usage (on the GPU box): tools/pmc_phase.sh <kernel-substring> "<stops>"  - SQ counters of one kernel per AT3HIP_DEBUG_STOP value
K=$1; STOPS=$2; shift 2
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmcphase
mkdir -p $OUT
cd /tmp
for st in $STOPS; do
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
             "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
    i=$((i+1))
    rm -rf $OUT/s${st}_$i
    AT3HIP_DEBUG_STOP=$st rocprofv3 --pmc $SET --kernel-trace -d $OUT/s${st}_$i -o sq -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/run.log 2>&1
  done
  python3 - <<PY
import glob, sqlite3
vals = {}
for f in glob.glob("$OUT/s${st}_*/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for k, c, v in db.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%$K%' group by kernel_name, counter_name"):
        vals[c] = v
w = vals.get("SQ_WAVES", 1)
print("stop=$st", " ".join(f"{c.replace('SQ_','')}={v/w:.0f}" for c, v in sorted(vals.items())))
PY
done
