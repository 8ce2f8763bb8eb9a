#!/usr/bin/env python3
"""Run on the GPU box: the pipelined step of configs[1] with nothing but a host clock around it (no per-stage timings are
read), for builds whose timing events are switched off. usage: AT3HIP_LIB=... tools/step_probe.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import atracdenc_amd
S, F = 64, 64
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
enc = atracdenc_amd.At3Hip(n_streams=S, max_blocks=F + 1)
g = torch.Generator(device="cuda"); g.manual_seed(1)
pcm = (torch.randint(-8192, 8192, (S, 2 * F + 1, 1024, 2), generator=g, device="cuda", dtype=torch.int32).to(torch.float32) / 32768.0)
prime = pcm[:, :1].contiguous(); b = [pcm[:, 1:1 + F].contiguous(), pcm[:, 1 + F:].contiguous()]
out = torch.zeros((S, F, enc.frame_size), dtype=torch.uint8, device="cuda")
enc.encode_device(prime.data_ptr(), 1, out.data_ptr())
for i in range(10): enc.encode_device(b[i & 1].data_ptr(), F, out.data_ptr(), asynchronous=True)
enc.sync(); torch.cuda.synchronize()
res = []
for rep in range(5):
    t0 = time.perf_counter()
    for i in range(steps): enc.encode_device(b[i & 1].data_ptr(), F, out.data_ptr(), asynchronous=True)
    enc.sync(); torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / steps * 1e3)
print("ms per step: median %.4f min %.4f max %.4f" % (float(np.median(res)), min(res), max(res)))
