#!/bin/bash
# Run on the GPU box: the life of every k_alloc_pack wavefront of one synchronous step (entry and exit on the 100 MHz clock,
# stamped by a profiling build): how long items take, how the lengths spread, and what the launch's tail costs.
# usage: tools/alloc_item_times.sh [bench args]     -> gpurun_out/alloc_item_times.npy ([item][entry, exit] in 10 ns ticks)
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
br = int(args[args.index("--bitrate") + 1]) if "--bitrate" in args else bench.LP2
bench.DeviceJob.sync_steps = True
job = bench.DeviceJob(0, 64, 64, br, False, kind, seed=1)
job.warmup(3)
job.run_steps(2)
n = 8192
c = job.enc.read_tap(B.TAP_CLOCK, np.uint64, (16 + 2 * 256 * 12 + 2 * 16384 + 12 * 16384,))
ph = c[16 + 2 * 256 * 12 + 2 * 16384:][: 12 * n].reshape(n, 12).astype(np.float64)
np.save("gpurun_out/alloc_item_phases.npy", ph)
t = c[16 + 2 * 256 * 12:][: 2 * n].reshape(n, 2).astype(np.int64)
np.save("gpurun_out/alloc_item_times.npy", t)
t0 = t[:, 0].min()
ent, ext = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # us
dur = ext - ent
print(f"input={kind} bitrate={br}: kernel span {ext.max():.1f} us; item life mean {dur.mean():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f} min {dur.min():.1f} us")
first = ent < 5.0
print(f"  wavefronts that entered in the first 5 us: {first.sum()} (life mean {dur[first].mean():.1f}); the rest: {(~first).sum()} (life mean {dur[~first].mean():.1f})")
print("  entry time percentiles (us):", " ".join(f"{np.percentile(ent, q):.1f}" for q in (1, 25, 50, 51, 75, 99)))
print("  exit time percentiles (us): ", " ".join(f"{np.percentile(ext, q):.1f}" for q in (1, 25, 50, 75, 90, 99, 100)))
busy = np.array([((ent <= x) & (ext > x)).sum() for x in np.arange(0, ext.max(), 5.0)])
print("  resident wavefronts every 5 us:", " ".join(str(b) for b in busy))
print(f"  sum of lives / 4096 slots = {dur.sum() / 4096:.1f} us (the span of a perfectly packed launch at these lives)")
# lives by item class
ch = np.arange(n) & 1
names = ["loads+scale", "e1 sums", "small units", "config", "rate loop", "units: rounding", "units: e2 sums", "units: EA lists", "units: EA ties+seq", "units: VLC cost", "emission"]
tot = ph[:, :11].sum(axis=1)
print(f"  per-item shader cycles: mean {tot.mean():.0f}, corr(life, cycles) = {np.corrcoef(dur, tot)[0, 1]:.3f}")
for k, nm in enumerate(names):
    x = ph[:, k]
    print(f"    {nm:20s} mean {x.mean():8.0f}  sd {x.std():8.0f}  p1 {np.percentile(x, 1):8.0f}  p99 {np.percentile(x, 99):8.0f}  corr with total {np.corrcoef(x, tot)[0, 1]:.2f}")
print(f"  life by channel: ch0 {dur[ch == 0].mean():.1f} ch1 {dur[ch == 1].mean():.1f}")
PY
