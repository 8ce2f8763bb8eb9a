#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity hunt (run on the GPU box): many short streams of varied synthetic material, all
option sets; prints every mismatching (family, seed, stream, frame). Usage: fuzz_gpu.py [rounds] [streams] [blocks] [first round]
(round r draws its material from seed 1000 + r: a later run continues where an earlier one stopped)"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import atracdenc_amd
from at3_testlib import LP2, LP4, have_ref, oracle, ref

def gen(rng, nb):
    n = nb * 1024
    t = np.arange(n, dtype=np.float64)
    fam = rng.randint(0, 12)
    x = np.zeros((n, 2))
    if fam == 0:      # noise at a random level, from 1 LSB to full scale
        a = 2.0 ** rng.uniform(-15, 0)
        x = rng.uniform(-a, a, size=(n, 2))
    elif fam == 1:    # filtered (coloured) noise, correlated channels
        w = rng.uniform(-1, 1, size=n + 64)
        k = rng.randint(2, 64)
        c = np.convolve(w, np.ones(k) / k, mode="same")[:n] * rng.uniform(0.05, 3.0)
        x[:, 0] = c
        x[:, 1] = c * rng.uniform(-1, 1) + rng.uniform(-1, 1, size=n) * rng.uniform(0, 0.05)
    elif fam == 2:    # sines with random frequencies / amplitudes, possibly clipping
        for _ in range(rng.randint(1, 8)):
            f = rng.uniform(20, 22000); a = rng.uniform(0.001, 0.7); ph = rng.uniform(0, 6.28)
            x[:, 0] += a * np.sin(2 * np.pi * f * t / 44100 + ph)
            x[:, 1] += a * np.sin(2 * np.pi * f * t / 44100 + ph + rng.uniform(0, 3.14))
    elif fam == 3:    # sparse impulses
        k = rng.randint(50, 5000)
        x[rng.randint(0, k)::k, 0] = rng.uniform(-1, 1)
        x[rng.randint(0, k)::k, 1] = rng.uniform(-1, 1)
    elif fam == 4:    # amplitude-modulated noise (gain-control workout)
        per = rng.randint(200, 6000)
        env = np.where((t // per) % 2 == 0, rng.uniform(0.0005, 0.05), rng.uniform(0.1, 1.0))
        x = rng.uniform(-1, 1, size=(n, 2)) * env[:, None]
    elif fam == 5:    # tone bursts with exponential decays (drums)
        per = rng.randint(800, 9000)
        f = rng.uniform(60, 9000)
        env = np.exp(-(t % per) / rng.uniform(30, 2000))
        x[:, 0] = env * np.sin(2 * np.pi * f * t / 44100) * rng.uniform(0.2, 1.0)
        x[:, 1] = np.roll(x[:, 0], rng.randint(0, 400)) * rng.uniform(-1, 1)
    elif fam == 6:    # square / sawtooth at full scale
        per = rng.randint(3, 400)
        x[:, 0] = np.where((t // per) % 2 == 0, 1.0, -1.0) * rng.uniform(0.3, 1.0)
        x[:, 1] = ((t % per) / per * 2 - 1) * rng.uniform(0.3, 1.0)
    elif fam == 7:    # chirps
        f0, f1 = rng.uniform(20, 2000), rng.uniform(2000, 22050)
        ph = 2 * np.pi * (f0 * t + (f1 - f0) * t * t / (2 * n)) / 44100
        x[:, 0] = np.sin(ph) * rng.uniform(0.05, 1.0)
        x[:, 1] = np.cos(ph * rng.uniform(0.5, 1.5)) * rng.uniform(0.05, 1.0)
    elif fam == 8:    # tiny signals: a few LSB of dither over DC
        x[:, 0] = rng.uniform(-0.5, 0.5) + rng.randint(-2, 3, size=n) / 32768.0
        x[:, 1] = rng.randint(-1, 2, size=n) / 32768.0
    elif fam == 9:    # silence with isolated non-silent blocks
        x = rng.uniform(-1, 1, size=(n, 2)) * (rng.rand(nb) < 0.3).repeat(1024)[:, None] * rng.uniform(0.01, 1.0)
    elif fam == 10:   # identical channels / inverted channels (joint-stereo edge: S = 0 or M = 0)
        c = rng.uniform(-1, 1, size=n) * rng.uniform(0.01, 1.0)
        x[:, 0] = c
        x[:, 1] = c if rng.rand() < 0.5 else -c
    else:             # mixture: tones + noise bed + clicks
        x = rng.uniform(-1, 1, size=(n, 2)) * 2.0 ** rng.uniform(-12, -3)
        for _ in range(rng.randint(1, 5)):
            x[:, rng.randint(0, 2)] += rng.uniform(0.01, 0.4) * np.sin(2 * np.pi * rng.uniform(100, 15000) * t / 44100)
        x[rng.randint(0, 3000)::rng.randint(1500, 7000)] += rng.uniform(-1, 1)
    s16 = np.clip(np.round(x * 32768.0), -32768, 32767)
    return fam, (s16.astype(np.float32) / np.float32(32768.0)).reshape(nb, 1024, 2)

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    o = oracle()
    pool = ThreadPoolExecutor(min(64, os.cpu_count() or 8))
    total = bad_total = 0
    t0 = time.time()
    for rd in range(first, first + rounds):
        rng = np.random.RandomState(1000 + rd)
        items = [gen(rng, nb) for _ in range(S)]
        pcm = np.stack([p for _, p in items])
        for br in (LP2, LP4):
            for ng, nt in ((0, 0), (1, 0), (0, 1)):
                enc = atracdenc_amd.At3Hip(n_streams=S, max_blocks=nb, bitrate=br, no_gain=ng, no_tonal=nt)
                if os.environ.get("FUZZ_GAIN_FORM"):   # AT3HIP_OPT_GAIN_FORM: 1 = the one-wavefront upsampler kernel
                    enc.set_option(atracdenc_amd.binding.OPT_GAIN_FORM, int(os.environ["FUZZ_GAIN_FORM"]))
                got = enc.encode(pcm)
                if os.environ.get("FUZZ_BOUNDS"):   # a -DAT3HIP_DEBUG_KNOBS build (AT3HIP_LIB): k_alloc_pack counts the lower bounds its rate loop decided with
                    clk = enc.read_tap(atracdenc_amd.binding.TAP_CLOCK, np.uint64, (16,))      # ... that were later replaced by bits (word 13) and those above them (14)
                    bounds_seen = globals().setdefault("_bounds", [0, 0])
                    bounds_seen[0] += int(clk[13]); bounds_seen[1] += int(clk[14])
                    if int(clk[14]):
                        bad_total += 1
                        print(f"BOUND ABOVE BITS round {rd} br {br} nogain {ng} notonal {nt}: {int(clk[14])} of {int(clk[13])}")
                enc.close()
                exp = list(pool.map(lambda i: o.encode(pcm[i], br, ng, nt)[0], range(S)))
                if have_ref():   # the oracle itself against the real reference on a slice of the same material
                    nref = min(S, 24)
                    rr = list(pool.map(lambda i: ref().encode(pcm[i], br, ng, nt)[0], range(nref)))
                    for i in range(nref):
                        if not np.array_equal(rr[i], exp[i]):
                            bad_total += 1
                            print(f"ORACLE != REFERENCE round {rd} br {br} nogain {ng} notonal {nt} stream {i} family {items[i][0]}")
                for i in range(S):
                    bad = np.flatnonzero((got[i] != exp[i]).any(axis=1))
                    total += got.shape[1]
                    if len(bad):
                        bad_total += len(bad)
                        print(f"MISMATCH round {rd} br {br} nogain {ng} notonal {nt} stream {i} family {items[i][0]} frames {bad[:8].tolist()}")
        print(f"round {rd}: {total} frames checked, {bad_total} mismatching, {time.time() - t0:.1f}s", flush=True)
    if os.environ.get("FUZZ_BOUNDS"):
        print("bounds checked against the bits that replaced them:", globals().get("_bounds", [0, 0])[0], "above them:", globals().get("_bounds", [0, 0])[1])
    print("FUZZ", "CLEAN" if bad_total == 0 else "FAILED", total, "frames")

if __name__ == "__main__":
    main()
