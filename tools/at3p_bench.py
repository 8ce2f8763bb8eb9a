#!/usr/bin/env python3
"""ATRAC3plus front-end timing (row f4; not the headline metric). Same audio as BASELINE configs[1]: 64 stereo streams x
65536 samples = 32 frames each, PCM resident in HBM."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atracdenc_amd import At3pHip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    rng = np.random.RandomState(1)
    pcm = (rng.randint(-8192, 8192, size=(a.streams, a.frames, 2048, 2)).astype(np.float32) / np.float32(32768.0))
    d_pcm = torch.from_numpy(pcm).cuda()
    d_specs = torch.zeros((a.streams, a.frames, 2, 2048), dtype=torch.float32, device="cuda")
    enc = At3pHip(n_streams=a.streams, max_frames=a.frames, channels=2)
    for _ in range(a.warmup):
        enc.pqf_mdct_device(d_pcm.data_ptr(), a.frames, d_specs.data_ptr())
    torch.cuda.synchronize()
    tms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        enc.pqf_mdct_device(d_pcm.data_ptr(), a.frames, d_specs.data_ptr())
        tms.append(enc.timings())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    units = a.streams * a.frames
    med = {k: float(np.median([t[k] for t in tms])) for k in tms[0]}
    algo = units * 2 * 2048 * 4 * 4   # PCM in, subbands out + in, spectrum out
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from at3_testlib import at3p_mdct, at3p_pqf, have_ref
    which = "ref" if have_ref() else "oracle"
    sample = np.ascontiguousarray(pcm[:8, :, :, 0].reshape(-1, 2048))
    t1 = time.perf_counter()
    at3p_mdct(at3p_pqf(sample, which), None, which)
    cpu = {"value": sample.shape[0] / 2 / (time.perf_counter() - t1), "unit": "frame pairs/s", "cores": 1,
           "kind": "reference" if which == "ref" else "port", "sample": f"{sample.shape[0]} mono frames, one thread"}
    print(json.dumps({"metric": "atrac3plus_frontend_frame_pairs_per_s", "value": units * a.steps / dt, "ms_per_step": 1e3 * dt / a.steps,
                      "audio_seconds_per_s": units * a.steps * 2048 / 44100 / dt, "device_ms": med,
                      "algorithmic_GBps": algo / ((med["pqf_ms"] + med["mdct_ms"]) * 1e-3) / 1e9,
                      "config": {"streams": a.streams, "frames": a.frames, "channels": 2}, "cpu_baseline": cpu}))


if __name__ == "__main__":
    main()
