#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity hunt for the ATRAC1 path (run on the GPU box): the signal families of fuzz_gpu.py,
all encoder settings, one and two channels, fed in random pieces; prints every mismatching (family, stream, unit).
Where oracle/_ref is present a slice of every round is also checked oracle-vs-reference.
Usage: fuzz_at1_gpu.py [rounds] [streams] [blocks(1024-sample)]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import atracdenc_amd
from at3_testlib import AT1_MODES, at1_blocks, at1_oracle_encode, at1_ref_encode, have_ref
from fuzz_gpu import gen


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    pool = ThreadPoolExecutor(min(64, os.cpu_count() or 8))
    total = bad_total = 0
    t0 = time.time()
    for rd in range(rounds):
        rng = np.random.RandomState(5000 + rd)
        items = [gen(rng, nb) for _ in range(S)]
        for nch in (2, 1):
            pcm = np.stack([at1_blocks(p, nch) for _, p in items])
            nu = pcm.shape[1]
            for mode in sorted(AT1_MODES):
                auto, mask, bfu = AT1_MODES[mode]
                enc = atracdenc_amd.At1Hip(n_streams=S, max_blocks=nu, channels=nch, window_auto=auto, window_mask=mask, bfu_idx_const=bfu)
                cuts = sorted(set([0, nu] + list(rng.randint(1, nu, size=2))))
                got = np.concatenate([enc.encode(pcm[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])], axis=1)
                enc.close()
                exp = list(pool.map(lambda i: at1_oracle_encode(pcm[i], mode), range(S)))
                if have_ref():
                    nref = min(S, 16)
                    rr = list(pool.map(lambda i: at1_ref_encode(pcm[i], mode), range(nref)))
                    for i in range(nref):
                        if not np.array_equal(rr[i], exp[i]):
                            bad_total += 1
                            print(f"ORACLE != REFERENCE round {rd} nch {nch} mode {mode} stream {i} family {items[i][0]}")
                for i in range(S):
                    bad = np.argwhere((got[i] != exp[i]).any(axis=2))
                    total += got.shape[1] * nch
                    if len(bad):
                        bad_total += len(bad)
                        print(f"MISMATCH round {rd} nch {nch} mode {mode} stream {i} family {items[i][0]} units {bad[:6].tolist()}")
        print(f"round {rd}: {total} sound units checked, {bad_total} mismatching, {time.time() - t0:.1f}s", flush=True)
    print("FUZZ", "CLEAN" if bad_total == 0 else "FAILED", total, "sound units")


if __name__ == "__main__":
    main()
