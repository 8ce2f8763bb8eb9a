#!/bin/bash
# GPU box: hardware counters of the fused kernel alone (tools/k1/lab.py, one size) for one or more builds of the library.
# usage: tools/k1/pmc.sh <size e.g. 1024x128> lib1.so [lib2.so ...]   -> stdout
SIZE=$1; shift
export TMPDIR=/tmp
REPO=$(pwd)
for LIB in "$@"; do
  echo "=== $LIB $SIZE"
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
             "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
             "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" \
             "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES_LT_64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_WAIT_INST_VMEM SQ_INSTS_FLAT SQ_INSTS_GDS"; do
    i=$((i+1))
    rm -rf /tmp/k1pmc$i
    (cd /tmp && rocprofv3 --pmc $SET --kernel-trace -d /tmp/k1pmc$i -o p -- python $REPO/tools/k1/lab.py --reps 1 --sizes $SIZE $REPO/$LIB > /tmp/k1pmc$i.log 2>&1)
  done
  python3 - <<PY
import glob, sqlite3
for i in range(1, 6):
    for f in glob.glob("/tmp/k1pmc%d/**/*.db" % i, recursive=True):
        db = sqlite3.connect(f)
        try:
            for k, c, v, n in db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%qmf_mdct8%' group by kernel_name, counter_name order by counter_name"):
                print("  %-26s %14.6g   (%d dispatches)" % (c, v, n))
        except Exception as e:
            print("  pass", i, "failed:", e)
PY
done
