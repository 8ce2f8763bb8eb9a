#!/usr/bin/env python3
"""Run on the GPU box (optionally under rocprofv3 --kernel-trace --memory-copy-trace): a few host-buffer calls of configs[1]
through the pinned double-buffered pipeline, timed from the host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
if "--torch" in sys.argv:
    import torch
    torch.zeros(4, device="cuda").sum().item()
import atracdenc_amd
S, F = 64, 64
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12
enc = atracdenc_amd.At3Hip(n_streams=S, max_blocks=F + 1)
ins = [enc.host_alloc((S, F, 1024, 2), np.float32) for _ in range(2)]
outs = [enc.host_alloc((S, F, enc.frame_size), np.uint8) for _ in range(2)]
for a in ins:
    a[...] = np.random.RandomState(1).randint(-8192, 8192, size=a.shape).astype(np.float32) / np.float32(32768.0)
enc.encode(np.zeros((S, 1, 1024, 2), np.float32))
for i in range(4):
    enc.encode_host_async(ins[i & 1], outs[i & 1])
enc.sync()
t0 = time.perf_counter()
stamps = []
for i in range(steps):
    if i >= 2 and "--nowait" not in sys.argv:
        enc.wait_input(1)        # (where a real caller refills the buffer)
    enc.encode_host_async(ins[i & 1], outs[i & 1])
    if i >= 1 and "--nowait" not in sys.argv:
        enc.wait_frames(1)       # (where a real caller consumes the previous call's frames): at most two calls in flight
    stamps.append(time.perf_counter() - t0)
enc.sync()
dt = time.perf_counter() - t0
print("host-buffer pipeline: %.4f ms per call; host-side queueing of the calls took" % (dt / steps * 1e3), ["%.3f" % (s * 1e3) for s in stamps], "ms")
