#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity hunt for the ATRAC3plus front end (run on the GPU box): the signal families of
fuzz_gpu.py at +-1.0 and at s16 scale, random steep-window flags, residual scaling, random pieces.
Usage: fuzz_at3p_gpu.py [rounds] [streams] [frames]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import atracdenc_amd
from at3_testlib import at3p_mdct, at3p_pqf, have_ref
from fuzz_gpu import gen


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    nf = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    pool = ThreadPoolExecutor(min(64, os.cpu_count() or 8))
    total = bad_total = 0
    t0 = time.time()
    for rd in range(rounds):
        rng = np.random.RandomState(9000 + rd)
        items = [gen(rng, 2 * nf) for _ in range(S)]
        scale = np.float32(32768.0 if rd % 2 else 1.0)
        pcm = np.stack([(p.reshape(nf, 2048, 2) * scale).astype(np.float32) for _, p in items])
        for mode in ("sine", "random", "random_residual"):
            flags = None if mode == "sine" else rng.randint(0, 65536, (S, nf, 2)).astype(np.uint16)
            rs = mode == "random_residual"
            enc = atracdenc_amd.At3pHip(n_streams=S, max_frames=nf, channels=2)
            cuts = sorted(set([0, nf] + list(rng.randint(1, nf, size=2))))
            parts = [enc.pqf_mdct(pcm[:, a:b], None if flags is None else flags[:, a:b], rs) for a, b in zip(cuts[:-1], cuts[1:])]
            enc.close()
            bands = np.concatenate([p[0] for p in parts], axis=1)
            specs = np.concatenate([p[1] for p in parts], axis=1)

            def check(i):
                bad = 0
                for c in range(2):
                    eb = at3p_pqf(pcm[i, :, :, c])
                    x = eb if not rs else (eb.astype(np.float64) / (32768.0 / 1.122018)).astype(np.float32)
                    es = at3p_mdct(x, None if flags is None else flags[i, :, c])
                    bad += int((bands[i, :, c].view(np.uint32) != eb.view(np.uint32)).any(axis=(1, 2)).sum())
                    bad += int((specs[i, :, c].view(np.uint32) != es.view(np.uint32)).any(axis=1).sum())
                    if have_ref() and i < 8:
                        if not np.array_equal(at3p_pqf(pcm[i, :, :, c], "ref").view(np.uint32), eb.view(np.uint32)) or \
                           not np.array_equal(at3p_mdct(x, None if flags is None else flags[i, :, c], "ref").view(np.uint32), es.view(np.uint32)):
                            print(f"ORACLE != REFERENCE round {rd} stream {i} ch {c}")
                            bad += 1
                return bad
            for i, b in enumerate(pool.map(check, range(S))):
                total += 2 * nf
                if b:
                    bad_total += b
                    print(f"MISMATCH round {rd} mode {mode} stream {i} family {items[i][0]}: {b} frames")
        print(f"round {rd}: {total} frames checked, {bad_total} mismatching, {time.time() - t0:.1f}s", flush=True)
    print("FUZZ", "CLEAN" if bad_total == 0 else "FAILED", total, "frames")


if __name__ == "__main__":
    main()
