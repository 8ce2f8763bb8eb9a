#!/usr/bin/env python3
"""Generate tests/golden/at3p_frontend.npz from the REAL reference's ATRAC3plus front end (oracle/_ref:
at3plus_pqf_do_analyse and TAt3pMDCT::Do compiled from the unmodified sources). Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from at3_testlib import ROOT, at3p_mdct, at3p_pqf, at3p_signal, have_ref  # noqa: E402


def main():
    if not have_ref():
        raise SystemExit("oracle/_ref/libat3ref.so missing")
    d = {}
    rng = np.random.RandomState(7)
    for name, scale in (("mix", 32768.0), ("noise", 1.0), ("stress", 1.0)):
        x = at3p_signal(name, 4, scale=scale)
        s16 = np.round(x / np.float32(scale) * 32768.0).astype(np.int16)
        assert np.array_equal((s16.astype(np.float32) / np.float32(32768.0) * np.float32(scale)).astype(np.float32), x)
        d[f"{name}_pcm_s16"] = s16
        d[f"{name}_scale"] = np.float32(scale)
        bands = at3p_pqf(x, "ref")
        d[f"{name}_bands"] = bands
        flags = rng.randint(0, 65536, size=4).astype(np.uint16)
        d[f"{name}_flags"] = flags
        d[f"{name}_specs_sine"] = at3p_mdct(bands, None, "ref")
        d[f"{name}_specs_mixed"] = at3p_mdct(bands, flags, "ref")
    # the reference's own synthesis-filter test vectors (atrac3plus_pqf/ut/test_data, used by ipqf_ut.cpp): inputs and
    # expected outputs, kept as two arrays of this fixture
    tdir = "/root/reference/src/atrac/atrac3plus_pqf/ut/test_data"
    d["ipqf_ut_in"] = np.fromfile(os.path.join(tdir, "ipqftest_pcm_mr.dat"), np.float32).reshape(4, 2048)
    d["ipqf_ut_out"] = np.fromfile(os.path.join(tdir, "ipqftest_pcm_out.dat"), np.float32).reshape(4, 2048)
    path = os.path.join(ROOT, "tests", "golden", "at3p_frontend.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
