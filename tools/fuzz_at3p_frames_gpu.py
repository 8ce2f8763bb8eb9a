#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity hunt for the ATRAC3plus frame writer (run on the GPU box): PCM of fuzz_gpu.py's signal
families through at3phip_encode_frames in random pieces, and synthetic spectra (levels from 1e-8 to clipping, tilts,
empty bands, random window flags) through at3phip_write_frames, stereo and mono. The first streams of every round are
also checked against the reference build when oracle/_ref is present.
Usage: fuzz_at3p_frames_gpu.py [rounds] [streams] [frames]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import atracdenc_amd
from at3_testlib import at3p_mdct, at3p_pqf, at3p_write_frames, have_ref
from fuzz_gpu import gen


def synth_specs(rng, nf, nch):
    lvl = np.exp(rng.uniform(np.log(1e-8), np.log(4.0)))
    sp = lvl * rng.standard_normal((nf, nch, 2048))
    kind = rng.randint(0, 5)
    if kind == 1:
        sp *= np.exp(-np.arange(2048) / rng.uniform(30, 2000))
    elif kind == 2:
        sp[:, :, rng.randint(16, 2048):] = 0.0
    elif kind == 3:
        sp *= (rng.uniform(size=(nf, nch, 2048)) < rng.uniform(0.01, 0.5))      # sparse lines
    elif kind == 4:
        sp = np.round(sp * 8) / 8 * rng.choice([1.0, 1.0 / 63.5, 0.5])           # ties and exact grid values
    return sp.astype(np.float32)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    nf = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    pool = ThreadPoolExecutor(min(64, os.cpu_count() or 8))
    total = bad_total = 0
    t0 = time.time()
    for rd in range(rounds):
        rng = np.random.RandomState(12000 + rd)
        nch = 2 if rd % 3 else 1
        # ---- PCM to frames, in pieces ----
        items = [gen(rng, 2 * nf) for _ in range(S)]
        pcm = np.stack([p.reshape(nf, 2048, 2)[:, :, :nch].astype(np.float32) for _, p in items])
        enc = atracdenc_amd.At3pHip(n_streams=S, max_frames=nf, channels=nch)
        cuts = sorted(set([0, nf] + list(rng.randint(1, nf, size=2))))
        got_pcm = np.concatenate([enc.encode_frames(pcm[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])], axis=1)
        # ---- spectra to frames ----
        specs = np.stack([synth_specs(rng, nf, nch) for _ in range(S)])
        flags = rng.randint(0, 65536, (S, nf, nch)).astype(np.uint16)
        flags[rng.uniform(size=flags.shape) < 0.3] = 0
        flags[rng.uniform(size=flags.shape) < 0.1] = 0xFFFF
        flags[rng.uniform(size=flags.shape) < 0.1] |= 0x00FF
        got_sp = enc.write_frames(specs, flags)
        enc.close()

        def check(i):
            bad = 0
            sp = np.zeros((nf, nch, 2048), np.float32)
            for c in range(nch):
                eb = at3p_pqf(pcm[i, :, :, c])
                sp[:, c] = at3p_mdct((eb.astype(np.float64) / (32768.0 / 1.122018)).astype(np.float32))
            e1 = at3p_write_frames(sp)
            e2 = at3p_write_frames(specs[i], flags[i])
            bad += int((got_pcm[i] != e1).any(axis=1).sum()) + int((got_sp[i] != e2).any(axis=1).sum())
            if have_ref() and i < 6:
                if not np.array_equal(at3p_write_frames(sp, None, "ref"), e1) or not np.array_equal(at3p_write_frames(specs[i], flags[i], "ref"), e2):
                    print(f"ORACLE != REFERENCE round {rd} stream {i}")
                    bad += 1
            return bad
        for i, b in enumerate(pool.map(check, range(S))):
            total += 2 * nf
            if b:
                bad_total += b
                print(f"MISMATCH round {rd} nch {nch} stream {i} family {items[i][0]}: {b} frames")
        print(f"round {rd} (channels {nch}): {total} frames checked, {bad_total} mismatching, {time.time() - t0:.1f}s", flush=True)
    print("FUZZ", "CLEAN" if bad_total == 0 else "FAILED", total, "frames")


if __name__ == "__main__":
    main()
