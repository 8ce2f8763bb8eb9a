#!/bin/bash
# Round-4 opening measurement on the GPU box: device properties, instruction-rate micro-benchmark with the observed clock,
# GPU tests, the bench line, headline + no-gain (fused kernel) profiles.
export TMPDIR=/tmp
O=gpurun_out/r04_base
mkdir -p $O
tools/devprop > $O/devprop.txt 2>&1
( rocm-smi --showclocks > $O/clocks_idle.txt 2>&1 ) || true
tools/ubench/valu_lds_rates > $O/ubench.txt 2>&1 &
UB=$!
sleep 2; ( rocm-smi --showclocks > $O/clocks_ubench.txt 2>&1 ) || true
wait $UB
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
bash tools/profile_gpu.sh r04base > $O/profile.log 2>&1
BENCH_ARGS=--no-gain bash tools/profile_gpu.sh r04base_nogain > $O/profile_nogain.log 2>&1
tail -3 $O/pytest_gpu.txt; cat $O/devprop.txt; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_base/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","parity_in_run")}, d["roofline"].get("sclk_mhz_observed"), [ (w["workload"][:30], w.get("value"), w.get("k1_isolated_ms")) for w in d.get("other_workloads",[])])
PY
