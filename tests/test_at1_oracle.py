"""ATRAC1 oracle (oracle/at1_oracle.c) against the committed golden sound units generated from
the real reference (tools/gen_golden_at1.py) and, where oracle/_ref exists, against the reference itself on longer
inputs. Bit-exact bytes."""
import os

import numpy as np
import pytest

from at3_testlib import (AT1_MODES, SIGNALS, at1_blocks, at1_oracle_encode, at1_ref_encode, have_ref, pcm_stress)

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "at1_encode.npz"))
CASES = sorted(k for k in GOLD.files if "_ch" in k)


@pytest.mark.parametrize("key", CASES)
def test_golden(oracle, key):
    name, ch, mode = key.split("_", 2)
    nch = int(ch[2:])
    pcm = (GOLD[f"{name}_pcm_s16"].astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    got = at1_oracle_encode(at1_blocks(pcm, nch), mode)
    assert got.shape == GOLD[key].shape
    assert np.array_equal(got, GOLD[key])


def test_golden_covers_block_switching_and_bfu_reduction():
    """The fixture must exercise short windows in every band and several BFU-amount indices, or it pins nothing."""
    frames = np.concatenate([GOLD[k].reshape(-1, 212) for k in CASES if k.endswith("_auto")])
    modes = {int(f[0]) >> 2 for f in frames}
    assert {0b101011, 0b000000}.issubset(modes) and len(modes) >= 4
    assert len({int(f[1]) >> 5 for f in frames}) >= 6


def test_streaming_equals_one_shot(oracle):
    """State carried between calls (QMF history, delay line, overlap, detector energy, loudness) lives in the encoder
    object only: feeding one block at a time gives the same bytes."""
    import ctypes
    from at3_testlib import ORACLE_SO, _vp
    lib = ctypes.CDLL(ORACLE_SO)
    lib.at1o_create.restype = ctypes.c_void_p
    pcm = at1_blocks(SIGNALS["mix"](10))
    want = at1_oracle_encode(pcm, "auto")
    e = ctypes.c_void_p(lib.at1o_create(2, 1, 0, 0))
    out = np.zeros_like(want)
    for b in range(pcm.shape[0]):
        lib.at1o_process(e, _vp(pcm[b]), _vp(out[b]), None, None, None)
    lib.at1o_destroy(e)
    assert np.array_equal(out, want)


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("mode", sorted(AT1_MODES))
@pytest.mark.parametrize("nch", [1, 2])
def test_vs_reference(oracle, mode, nch):
    gens = dict(SIGNALS)
    gens["stress"] = pcm_stress
    for name, gen in gens.items():
        blocks = at1_blocks(gen(66 if name == "stress" else 40), nch)
        assert np.array_equal(at1_oracle_encode(blocks, mode), at1_ref_encode(blocks, mode)), name


def test_product_host_tables_equal_oracle_tables(oracle):
    """The constant tables the library builds on the host (at1hip_host_tables, no GPU) are, bit for bit, the oracle's
    restatement of the reference's static initialisers."""
    import ctypes
    import atracdenc_amd
    from atracdenc_amd.binding import at1_host_tables
    from at3_testlib import ORACLE_SO, _vp
    if not os.path.exists(atracdenc_amd.LIB_PATH):
        atracdenc_amd.build_library()
    t = at1_host_tables()
    names = ["qmf_win", "scale", "sine", "sc512", "sc256", "sc64", "tw128", "tw64", "tw16", "loud", "ath_bfu"]
    n = sum(t[k].size for k in names)
    ref = np.zeros(n, np.float32)
    assert ctypes.CDLL(ORACLE_SO).at1o_tables(_vp(ref), n) == n
    got = np.concatenate([t[k].ravel() for k in names])
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_psy_tables_vs_reference(oracle):
    """Loudness curve and the per-BFU threshold in quiet against the reference's CalcATH / CreateLoudnessCurve; the
    frame bytes are not sensitive enough to pin these (a float/double slip in the ATH formula went unnoticed by them)."""
    import ctypes
    from at3_testlib import ORACLE_SO, REF_SO, _vp
    n = 48 + 64 + 32 + 256 + 128 + 32 + 2 * (128 + 64 + 16) + 512 + 52
    tab = np.zeros(n, np.float32)
    assert ctypes.CDLL(ORACLE_SO).at1o_tables(_vp(tab), n) == n
    ath = np.zeros(512, np.float32)
    loud = np.zeros(512, np.float32)
    ctypes.CDLL(REF_SO).at1ref_psy_tables(_vp(ath), _vp(loud))
    assert np.array_equal(tab[n - 52 - 512:n - 52].view(np.uint32), loud.view(np.uint32))
    start = [0, 8, 16, 24, 32, 36, 40, 44, 48, 56, 64, 72, 80, 86, 92, 98, 104, 110, 116, 122, 128, 134, 140, 146, 152, 159, 166,
             173, 180, 189, 198, 207, 216, 226, 236, 246, 256, 268, 280, 292, 304, 316, 328, 340, 352, 372, 392, 412, 432, 452,
             472, 492, 512]
    # CalcAt1ATH (atrac1_bitalloc.cpp:118-135): pow(10, 0.1 * min over the BFU's lines), double pow, float result
    want = np.array([np.float32(np.power(10.0, 0.1 * np.float64(ath[start[b]:start[b + 1]].min()))) for b in range(52)], np.float32)
    assert np.array_equal(tab[n - 52:].view(np.uint32), want.view(np.uint32))
