"""ATRAC3plus frame writer without tonal block (oracle/at3p_frame_oracle.c): the oracle against
golden frames written by the real reference (tools/gen_golden_at3p_frames.py), against the reference itself where
oracle/_ref exists, against the reference's own known answer for the word-length section (at3p_bitstream_ut.cpp:112-138),
and the host-built tables of the product library (no GPU needed)."""
import ctypes
import os

import numpy as np
import pytest

from at3_testlib import ORACLE_SO, _vp, at3p_specs, at3p_write_frames, have_ref

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "at3p_frames.npz"))
NAMES = sorted(k[:-6] for k in GOLD.files if k.endswith("_specs"))


@pytest.mark.parametrize("name", NAMES)
def test_golden(oracle, name):
    sp, fl = GOLD[f"{name}_specs"], GOLD[f"{name}_flags"]
    assert np.array_equal(at3p_write_frames(sp), GOLD[f"{name}_frames_sine"])
    assert np.array_equal(at3p_write_frames(sp, fl), GOLD[f"{name}_frames_flags"])


def test_golden_covers_the_unit_count_search(oracle):
    """The fixture holds frames with 32, 28, 27 and fewer quant units (the count is the frame's bits 3..7, plus one)."""
    counts = set()
    for name in NAMES:
        for fr in GOLD[f"{name}_frames_sine"]:
            counts.add(((int(fr[0]) << 8 | int(fr[1])) >> 8 & 0x1F) + 1)
    assert 32 in counts and 28 in counts and min(counts) < 28, counts


def test_wordlen_known_answer(oracle):
    """AT3PBitstream.Wordlen: six quant units of word length 6 in both channels take 28 bits."""
    lib = ctypes.CDLL(ORACLE_SO)
    wl = np.full(6, 6, np.uint8)
    assert lib.at3po_wordlen_bits(_vp(wl), _vp(wl), 6, 2, None) == 28


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("nch", [1, 2])
@pytest.mark.parametrize("name", ["noise", "tones", "burst", "mix", "silence", "stress"])
def test_vs_reference_signals(oracle, name, nch):
    sp = at3p_specs(name, 8, nch)
    rng = np.random.RandomState(3)
    flags = rng.randint(0, 65536, size=(8, nch)).astype(np.uint16)
    flags[0] = 0
    flags[1] = 0xFFFF
    flags[2] = 0x00FF      # "all steep" by the 8-bit mask of TAt3pMDCTWin::IsAllSteep
    for fl in (None, flags):
        assert np.array_equal(at3p_write_frames(sp, fl), at3p_write_frames(sp, fl, "ref"))


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("level", [0.0, 1e-7, 1e-4, 0.02, 0.3, 1.0, 3.0])
def test_vs_reference_random_levels(oracle, level, capfd):
    """White spectra from silence to clipping: scale factor search, the clip at 0.99999, the unit-count search."""
    rng = np.random.RandomState(int(level * 1000) + 1)
    sp = (level * rng.standard_normal((6, 2, 2048))).astype(np.float32)
    sp[1, :, 512:] = 0.0                      # empty upper units
    sp[2, 0] *= np.float32(0.01)              # very different channels
    got, info = at3p_write_frames(sp, info=True)
    assert np.array_equal(got, at3p_write_frames(sp, None, "ref"))
    capfd.readouterr()                        # (the reference reports clipping on stderr)
    assert np.all(info["bits_used"] <= 2048 * 8)


def test_host_write_tables(oracle):
    """at3phip_host_write_tables (no GPU): the spectrum-independent head of a silent stereo / mono frame equals the first
    bits the oracle writes, and the fixed bit counts add up to the oracle's frame length."""
    from atracdenc_amd.binding import load_library
    lib = load_library()
    nbytes = 41384
    buf = np.zeros(nbytes, np.uint8)
    assert lib.at3phip_host_write_tables(_vp(buf), nbytes) == 0
    assert lib.at3phip_host_write_tables(_vp(buf), nbytes - 4) != 0
    head = np.dtype([("words", "<u4", 8), ("nbits", "<u2"), ("fixed_bits", "<u2")])
    heads = buf[34876:34876 + 66 * head.itemsize].view(head).reshape(2, 33)   # after the code tables and scalar tables
    for nch in (1, 2):
        frames, info = at3p_write_frames(np.zeros((1, nch, 2048), np.float32), info=True)
        n = int(info["num_quant_units"][0])
        h = heads[nch - 1, n]
        want = np.unpackbits(frames[0])[: h["nbits"]]
        have = np.unpackbits(h["words"].astype(">u4").view(np.uint8))[: h["nbits"]]
        assert np.array_equal(want, have)
        tonal = (2 if nch == 2 else 0) + nch + nch + 1 + 1 + 2
        assert 3 + int(h["fixed_bits"]) + int(info["qu_bits"][0, :nch, :n].sum()) + tonal == int(info["bits_used"][0])
