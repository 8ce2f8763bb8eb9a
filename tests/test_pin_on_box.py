"""The oracle's pin, re-verified ON THE MACHINE THAT MEASURES: a slice of test_oracle_vs_ref.py / test_oracle_golden.py
marked `gpu` so that the driver's `-m gpu` run proves, on the GPU box (whose libm builds the product's tables at
at3hip_create), that oracle == reference build (oracle/_ref, travels as a prebuilt .so) == committed golden vectors, and
that the PRODUCT's host-built tables equal the reference's. The full matrices stay in the CPU suite."""
import numpy as np
import pytest

from at3_testlib import LP2, LP4, SIGNALS, TAP_DTYPE, have_ref, pcm_stress, ref

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def test_reference_build_travelled():
    assert have_ref(), "oracle/_ref/libat3ref.so is missing on this box: the oracle's pin cannot be re-verified here"


def test_oracle_tables_equal_golden(oracle, golden_stages):
    for k, v in oracle.tables().items():
        assert np.array_equal(bits(v), bits(golden_stages[f"table_{k}"])), k


def test_product_tables_equal_golden(golden_stages):
    """at3hip_host_tables = the block at3hip_create uploads, built with THIS host's libm, against the arrays the reference
    wrote into tests/golden/stages.npz (tools/gen_golden.py) and against the oracle's tables for the rest."""
    from atracdenc_amd import binding as B
    t = B.at3_host_tables()
    g = golden_stages
    assert np.array_equal(bits(t["scale"]), bits(g["table_scale"]))
    assert np.array_equal(bits(t["enc_win"]), bits(g["table_encwin"]))
    assert np.array_equal(bits(t["gain_level"]), bits(g["table_gainlevel"]))
    assert np.array_equal(bits(t["gain_interp"][:31]), bits(g["table_gaininterp"]))
    assert np.array_equal(bits(t["qmf_win"]), bits(g["table_qmfwin"]))
    assert np.array_equal(bits(t["loud_curve"]), bits(g["table_loud"]))
    # ATH per BFU (atrac3_bitstream.cpp:694-718): minimum of the per-line thresholds the reference's CalcATH produced
    bfu_start = [0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160, 176, 192, 224, 256, 288, 320, 352, 384, 416, 448, 480, 512,
                 576, 640, 704, 768, 896, 1024]
    ath = g["table_ath"]
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.pow.restype = ctypes.c_double
    libm.pow.argtypes = [ctypes.c_double, ctypes.c_double]
    exp = np.array([np.float32(libm.pow(10.0, float(np.float32(0.1) * np.float32(min(np.float32(999.0), ath[bfu_start[b]:bfu_start[b + 1]].min())))))
                    for b in range(32)], np.float32)
    assert np.array_equal(bits(t["ath_bfu"]), bits(exp))


@pytest.mark.parametrize("br", [LP2, LP4])
def test_oracle_equals_reference_option_sets(oracle, br):
    """One signal per option set (frames and stage taps), as test_oracle_vs_ref.py::test_frames_and_taps does for all."""
    r = ref()
    for name, (ng, nt) in zip(("mix", "burst", "tones", "noise"), ((0, 0), (1, 0), (0, 1), (1, 1))):
        pcm = SIGNALS[name](24)
        fo, to = oracle.encode(pcm, br, ng, nt, taps=True)
        fr, tr = r.encode(pcm, br, ng, nt, taps=True)
        assert np.array_equal(fo, fr), (name, ng, nt)
        for k in TAP_DTYPE.names:
            if k != "tonal_pos":
                assert np.array_equal(bits(to[k]), bits(tr[k])), (name, k)


def test_oracle_equals_reference_container_rows(oracle):
    r = ref()
    pcm = SIGNALS["mix"](16)
    stress = pcm_stress(20)
    for br, fsz in ((66150, 192), (93713, 272), (104738, 304), (132300, 384), (146081, 424), (176400, 512), (264600, 768), (352800, 1024)):
        for bfu in (0, 8):
            fo, fr = oracle.encode(pcm, br, 0, 0, bfu)[0], r.encode(pcm, br, 0, 0, bfu)[0]
            assert fo.shape == (15, fsz) and np.array_equal(fo, fr), (br, bfu)
        assert np.array_equal(oracle.encode(stress, br)[0], r.encode(stress, br)[0]), br


def test_oracle_equals_reference_flatness_and_log2f(oracle):
    """The two libm-dependent stage functions (std::log / std::exp in CalcSpectralFlatnessPerBfu, std::log2(float)) on this
    box's libm: oracle (which calls libm for the first and restates the second) against the reference build."""
    rng = np.random.RandomState(17)
    r = ref()
    for scale in (1.0, 1e-3, 1e-6):
        e = (rng.uniform(0, 1, 1024) ** 4 * scale).astype(np.float32)
        assert np.array_equal(bits(oracle.flatness(e)), bits(r.flatness(e)))
    x = np.exp(rng.uniform(-20, 20, 5000)).astype(np.float32)
    assert all(np.float32(oracle.log2f(v)).view(np.uint32) == np.float32(r.log2f(v)).view(np.uint32) for v in x)
