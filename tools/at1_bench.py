#!/usr/bin/env python3
"""ATRAC1 path timing (row f3; not the headline metric - bench.py stays on the ATRAC3 north star).
Same audio as BASELINE configs[1]: 64 stereo streams x 65536 samples = 128 ATRAC1 blocks each, PCM resident in HBM."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from atracdenc_amd import At1Hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    from at3_testlib import at1_blocks, pcm_mix
    base = np.stack([at1_blocks(pcm_mix(a.blocks // 2, seed=s)) for s in range(8)])
    pcm = np.concatenate([base] * (a.streams // 8), axis=0)[:a.streams]
    d_pcm = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    d_out = torch.zeros((a.streams, a.blocks, 2, 212), dtype=torch.uint8, device="cuda")
    enc = At1Hip(n_streams=a.streams, max_blocks=a.blocks)
    for _ in range(a.warmup):
        enc.encode_device(d_pcm.data_ptr(), a.blocks, d_out.data_ptr())
    torch.cuda.synchronize()
    tms = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        enc.encode_device(d_pcm.data_ptr(), a.blocks, d_out.data_ptr())
        tms.append(enc.timings())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    units = a.streams * a.blocks
    # the reference encoder (oracle/_ref, else the oracle port) on one host core, bounded sample of the same material
    from at3_testlib import at1_oracle_encode, at1_ref_encode, have_ref
    sample = np.ascontiguousarray(np.concatenate([base[i] for i in range(4)], axis=0))   # 4 streams' worth of blocks, one after another
    t1 = time.perf_counter()
    (at1_ref_encode if have_ref() else at1_oracle_encode)(sample, "auto")
    cpu = {"value": sample.shape[0] / (time.perf_counter() - t1), "unit": "sound unit pairs/s", "cores": 1,
           "kind": "reference" if have_ref() else "port", "sample": f"{sample.shape[0]} stereo blocks, one thread"}
    med = {k: float(np.median([t[k] for t in tms])) for k in tms[0]}
    print(json.dumps({"metric": "atrac1_sound_unit_pairs_per_s", "value": units * a.steps / dt, "ms_per_step": 1e3 * dt / a.steps,
                      "audio_seconds_per_s": units * a.steps * 512 / 44100 / dt, "device_ms": med,
                      "config": {"streams": a.streams, "blocks": a.blocks, "channels": 2}, "cpu_baseline": cpu}))


if __name__ == "__main__":
    main()
