#!/bin/bash
# Run on the GPU box: parity slice + isolated / pipelined step time + k_alloc_pack phase cycles of the current build.
export TMPDIR=/tmp
O=gpurun_out/${1:-quick}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python bench.py --no-cpu-baseline --no-side-workloads --regions 4 > $O/bench.json 2>$O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("bench", d["value"], d["ms_per_step"], d["parity_in_run"], d["stage_ms_per_step"])
PY
cd /tmp; rm -rf /tmp/qs; rocprofv3 --kernel-trace --stats -d /tmp/qs -o qs -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-workloads --regions 0 --no-parity --sync-steps > /dev/null 2>&1
python3 - <<PY
import glob, sqlite3
for f in glob.glob("/tmp/qs/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    for name, calls, total, avg in db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"):
        n = name.split("(")[0].replace("void ", "").replace("at3::", "")
        if n.startswith("k_"): print(f"  {n:28s} {calls:4d} {avg/1e3:9.2f} us")
PY
cd $OLDPWD
bash tools/alloc_phase_cycles.sh 2>&1 | grep -v amdgpu.ids
