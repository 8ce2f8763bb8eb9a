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
