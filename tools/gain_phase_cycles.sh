#!/bin/bash
# Run on the GPU box: where a k_gain_analysis1 wavefront's life goes - shader cycles per phase, stamped by every wavefront of a
# profiling build (-DAT3HIP_DEBUG_KNOBS) and summed through AT3HIP_TAP_CLOCK. usage: tools/gain_phase_cycles.sh [--sync-steps] [--input x]
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DAT3HIP_DEBUG_KNOBS -o $REPO/gpurun_out/libat3hip_dbg.so \
  $REPO/atracdenc_amd/csrc/at3hip.hip $REPO/atracdenc_amd/csrc/at1hip.hip $REPO/atracdenc_amd/csrc/at3phip.hip $REPO/atracdenc_amd/csrc/at3_tables.cpp 2>/dev/null
AT3HIP_LIB=$REPO/gpurun_out/libat3hip_dbg.so python3 - "$@" <<'PY'
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, bench
from atracdenc_amd import binding as B
args = sys.argv[1:]
kind = args[args.index("--input") + 1] if "--input" in args else "noise"
sync = "--sync-steps" in args
bench.DeviceJob.sync_steps = sync
job = bench.DeviceJob(0, 64, 64, bench.LP2, False, kind, seed=1)
job.warmup(3)
def read():
    c = job.enc.read_tap(B.TAP_CLOCK, np.uint64, (16 + 2 * 256 * 12,)).astype(np.float64)
    return c[16 + 256 * 12:].reshape(256, 12).sum(axis=0)
c0 = read()
job.run_steps(20)
d = read() - c0
waves = d[11]
names = ["gate + launch", "bins -> leaves", "passes m=2,8 (x2)", "exchange 1 (x2)", "passes m=32,128 (x2)", "exchange 2 (permlane)", "pass m=512 + out",
         "micro-chunks", "sub-frames + sort"]
tot = d[:9].sum()
print(f"k_gain_analysis1, input={kind} sync={sync}: {int(waves)} wavefronts past the gate, {tot / max(waves, 1):.0f} cycles per wavefront")
for k, n in enumerate(names):
    print(f"  {n:24s} {d[k] / max(waves, 1):9.0f} cycles  {100 * d[k] / tot:5.1f} %")
PY
