#!/usr/bin/env python3
"""Generate tests/golden/at1_encode.npz from the REAL reference's ATRAC1 encoder (oracle/_ref/libat3ref.so,
at1ref_encode, built from the unmodified sources under /root/reference by `make -C oracle ref`).

Run in the build container only:  python tools/gen_golden_at1.py
The fixture is data: s16 PCM inputs + the reference's 212-byte sound units; no reference source is stored."""
import os
import platform
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from at3_testlib import AT1_MODES, ROOT, SIGNALS, at1_blocks, at1_ref_encode, have_ref, pcm_stress  # noqa: E402

NBLOCKS = 8  # ATRAC3-sized blocks -> 16 ATRAC1 sound units per channel


def main():
    if not have_ref():
        raise SystemExit("oracle/_ref/libat3ref.so missing: run `make -C oracle ref` where /root/reference exists")
    d = dict(meta=np.array(repr(dict(glibc=platform.libc_ver()[1], machine=platform.machine()))))
    gens = dict(SIGNALS)
    gens["stress"] = pcm_stress
    for name, gen in gens.items():
        pcm = gen(34 if name == "stress" else NBLOCKS)
        d[f"{name}_pcm_s16"] = np.round(pcm * 32768.0).astype(np.int16)
        for nch in (2, 1):
            blocks = at1_blocks(pcm, nch)
            for mode in AT1_MODES:
                if name != "stress" and mode not in ("auto", "short", "auto_bfu3"):
                    continue
                if name == "stress" and nch == 1 and mode != "auto":
                    continue
                d[f"{name}_ch{nch}_{mode}"] = at1_ref_encode(blocks, mode)
    path = os.path.join(ROOT, "tests", "golden", "at1_encode.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path), "bytes,", len(d), "arrays")


if __name__ == "__main__":
    main()
