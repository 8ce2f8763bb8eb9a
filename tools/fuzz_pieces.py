#!/usr/bin/env python3
"""Randomised check of the carried stream state (run on the GPU box): the same streams fed in random pieces,
synchronously and with AT3HIP_ASYNC, against the one-shot oracle encode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import atracdenc_amd
from at3_testlib import LP2, LP4, oracle
import fuzz_gpu as fz

def main():
    o = oracle()
    rng = np.random.RandomState(77)
    S, nb, bad = 48, 70, 0
    for rd in range(6):
        pcm = np.stack([fz.gen(rng, nb)[1] for _ in range(S)])
        for br in (LP2, LP4):
            exp = np.stack([o.encode(pcm[i], br)[0] for i in range(S)])
            for asyn in (False, True):
                maxp = int(rng.randint(1, 17))
                enc = atracdenc_amd.At3Hip(n_streams=S, max_blocks=maxp, bitrate=br)
                pos, ins, outs, counts = 0, [], [], []
                while pos < nb:
                    k = int(min(nb - pos, rng.randint(1, maxp + 1)))
                    x = torch.from_numpy(np.ascontiguousarray(pcm[:, pos:pos + k])).cuda()
                    y = torch.zeros((S * k * enc.frame_size,), dtype=torch.uint8, device="cuda")
                    torch.cuda.synchronize()
                    counts.append(enc.encode_device(x.data_ptr(), k, y.data_ptr(), asynchronous=asyn))
                    ins.append(x); outs.append(y); pos += k
                enc.sync()
                fs = enc.frame_size
                got = np.concatenate([y.cpu().numpy()[: S * n * fs].reshape(S, n, fs) for y, n in zip(outs, counts)], axis=1)
                enc.close()
                nbad = int((got != exp).any(axis=2).sum())
                bad += nbad
                print(f"round {rd} br {br} async {asyn} max piece {maxp} calls {len(counts)}: {nbad} mismatching frames", flush=True)
    print("PIECES", "CLEAN" if bad == 0 else "FAILED")

if __name__ == "__main__":
    main()
