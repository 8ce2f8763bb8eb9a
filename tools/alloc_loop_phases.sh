#!/bin/bash
# Run on the GPU box: where a k_alloc_pack wavefront's life goes - shader cycles per phase, stamped by every wavefront of a
# profiling build (-DAT3HIP_DEBUG_KNOBS -DAT3_LOOP_PHASES) on the REAL path and summed through AT3HIP_TAP_CLOCK. usage: tools/alloc_phase_cycles.sh [bench args]
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DAT3HIP_DEBUG_KNOBS -DAT3_LOOP_PHASES -o $REPO/gpurun_out/libat3hip_dbg.so \
  $REPO/atracdenc_amd/csrc/at3hip.hip $REPO/atracdenc_amd/csrc/at1hip.hip $REPO/atracdenc_amd/csrc/at3phip.hip $REPO/atracdenc_amd/csrc/at3_tables.cpp 2>/dev/null
AT3HIP_LIB=$REPO/gpurun_out/libat3hip_dbg.so python3 - "$@" <<'PY'
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, bench
from atracdenc_amd import binding as B
args = sys.argv[1:]
kind = args[args.index("--input") + 1] if "--input" in args else "noise"
br = int(args[args.index("--bitrate") + 1]) if "--bitrate" in args else bench.LP2
sync = "--sync-steps" in args
bench.DeviceJob.sync_steps = sync
job = bench.DeviceJob(0, 64, 64, br, False, kind, seed=1)
job.warmup(3)
def read():
    c = job.enc.read_tap(B.TAP_CLOCK, np.uint64, (16 + 256 * 12,)).astype(np.float64)
    return np.concatenate([c[:2], c[16:].reshape(256, 12).sum(axis=0)])
c0 = read()
job.run_steps(20)
c1 = read()
d = c1 - c0
waves = d[13]
names = ["loads+scale", "e1 sums", "small units", "config", "loop: rest", "loop: trip head + memo", "loop: allocation + tonal", "loop: sums + decision", "loop: record + compare",
         "loop: BFU drops", "units + emission", "-"]
tot = d[2:13].sum()
print(f"input={kind} bitrate={br} sync={sync}: {int(waves)} wavefronts, {tot / waves:.0f} cycles per wavefront, sclk {job.enc.sclk_mhz():.0f} MHz")
for k, n in enumerate(names[:11]):
    print(f"  {n:20s} {d[2 + k] / waves:9.0f} cycles  {100 * d[2 + k] / tot:5.1f} %")
PY
