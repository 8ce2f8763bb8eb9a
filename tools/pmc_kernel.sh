#!/bin/bash
# usage (on the GPU box): tools/pmc_kernel.sh <tag> [bench args]  -> gpurun_out/pmc_<tag>/summary.txt
TAG=$1; shift
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace -d $OUT/pmc_sq$i -o sq -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/run$i.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/run_stats.log 2>&1
python3 - <<PY > $OUT/summary.txt
import glob, sqlite3, os
out = "$OUT"
for f in sorted(glob.glob(os.path.join(out, "stats", "**", "*.db"), recursive=True)):
    db = sqlite3.connect(f)
    for name, calls, avg in db.execute("select name, count(*), avg(end-start) from kernels group by name order by 3 desc"):
        print(f"{name[:60]:60s} calls={calls} avg_us={avg/1e3:.2f}")
for d in sorted(glob.glob(os.path.join(out, "pmc_sq*"))):
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        for k, c, v in db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name"):
            if k.startswith("void at3") or k.startswith("at3"):
                print(f"{k[:40]:40s} {c:26s} {v:.6g}")
PY
cat $OUT/summary.txt
