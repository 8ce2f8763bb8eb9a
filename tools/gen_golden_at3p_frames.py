#!/usr/bin/env python3
"""Generate tests/golden/at3p_frames.npz from the REAL reference's ATRAC3plus frame writer (oracle/_ref:
TScaler<NAt3p::TScaleTable>::ScaleFrame + TAt3PBitStream::WriteFrame(channels, nullptr, sces) compiled from the unmodified
sources). Inputs are kept as float32 bit patterns. Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from at3_testlib import ROOT, at3p_specs, at3p_write_frames, have_ref  # noqa: E402


def main():
    if not have_ref():
        raise SystemExit("oracle/_ref/libat3ref.so missing")
    d = {}
    rng = np.random.RandomState(11)
    cases = {
        "burst2": at3p_specs("burst", 3, 2),        # drops to 28 and 27 quant units
        "mix2": at3p_specs("mix", 3, 2),
        "tones1": at3p_specs("tones", 3, 1),
        "rand2": rng.standard_normal((2, 2, 2048)).astype(np.float32),   # full-scale noise: 26 quant units, clipping
        "tiny1": (1e-5 * rng.standard_normal((2, 1, 2048))).astype(np.float32),
    }
    for name, sp in cases.items():
        nf, nch, _ = sp.shape
        flags = rng.randint(0, 65536, size=(nf, nch)).astype(np.uint16)
        flags[0] = 0
        if nf > 1:
            flags[1] = 0xFFFF
        d[f"{name}_specs"] = sp
        d[f"{name}_flags"] = flags
        d[f"{name}_frames_sine"] = at3p_write_frames(sp, None, "ref")
        d[f"{name}_frames_flags"] = at3p_write_frames(sp, flags, "ref")
    path = os.path.join(ROOT, "tests", "golden", "at3p_frames.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
