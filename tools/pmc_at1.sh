#!/bin/bash
# usage (on the GPU box): tools/pmc_at1.sh  -> SQ counter averages per ATRAC1 kernel (separate --pmc passes, kernel trace only)
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_at1
mkdir -p $OUT
cd /tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1)); rm -rf $OUT/p$i
  rocprofv3 --pmc $SET --kernel-trace -d $OUT/p$i -o sq -- python $REPO/tools/at1_bench.py --steps 3 --warmup 1 "$@" > $OUT/run$i.log 2>&1
done
python3 - <<PY | tee $OUT/summary.txt
import glob, sqlite3
vals = {}
for f in glob.glob("$OUT/p*/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for k, c, v in db.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%at1%' group by kernel_name, counter_name"):
        vals.setdefault(k.split("(")[0][:40], {})[c] = v
for k, d in vals.items():
    print(k)
    print("   " + " ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())))
PY
