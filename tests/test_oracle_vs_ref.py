"""Oracle vs the real reference build (oracle/_ref). Only runs where oracle/_ref/libat3ref.so exists
(built in the build container from /root/reference; travels to the GPU box as a prebuilt .so)."""
import numpy as np
import pytest

from at3_testlib import (LP2, LP4, SIGNALS, TAP_DTYPE, capture_ref_diagnostics, flatness_threshold_walk, have_ref, oracle_diag_counts, pcm_hot,
                         pcm_stress, ref)

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")


def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("name", sorted(SIGNALS))
@pytest.mark.parametrize("br", [LP2, LP4])
def test_frames_and_taps(oracle, name, br):
    r = ref()
    pcm = SIGNALS[name](40)
    for ng, nt in ((0, 0), (1, 0), (0, 1), (1, 1)):
        fo, to = oracle.encode(pcm, br, ng, nt, taps=True)
        fr, tr = r.encode(pcm, br, ng, nt, taps=True)
        assert np.array_equal(fo, fr)
        for k in TAP_DTYPE.names:
            if k == "tonal_pos":  # reference keeps a pointer into a per-call temporary: not observable
                continue
            assert np.array_equal(bits(to[k]), bits(tr[k])), k


@pytest.mark.parametrize("name", ["burst", "mix", "tones"])
def test_mono_lp2(oracle, name):
    """One input channel, discrete-stereo container: one sound unit, stored twice (atrac3_bitstream.cpp:836-843).
    The oracle restates that path; the duplicated-channel stereo encode must give the same frame bytes, which is
    what the GPU boundary relies on."""
    pcm = SIGNALS[name](30)
    mono = np.ascontiguousarray(pcm[:, :, :1])
    fo = oracle.encode(mono, LP2)[0]
    assert np.array_equal(fo, ref().encode(mono, LP2)[0])
    assert np.array_equal(fo[:, :192], fo[:, 192:])
    assert np.array_equal(fo, oracle.encode(np.repeat(mono, 2, axis=2), LP2)[0])


# every row of the container table (atrac3.h:211-220): bitrate -> (frame size, joint stereo)
CONTAINER_ROWS = [(66150, 192, 1), (93713, 272, 1), (104738, 304, 0), (132300, 384, 0), (146081, 424, 0), (176400, 512, 0),
                  (264600, 768, 0), (352800, 1024, 0)]


@pytest.mark.parametrize("br,fsz,js", CONTAINER_ROWS)
def test_container_rows_and_bfu_idx_const(oracle, br, fsz, js):
    """All eight container rows x BfuIdxConst in {0, 1, 8, 20, 32} (atrac3_bitstream.cpp:567-585, 646: a constant BFU
    count disables the CheckBfus restart) on material with gain curves and tonal components."""
    r = ref()
    for name in ("mix", "burst", "tones"):
        pcm = SIGNALS[name](24)
        for bfu in (0, 1, 8, 20, 32):
            fo, to = oracle.encode(pcm, br, 0, 0, bfu, taps=True)
            fr, tr = r.encode(pcm, br, 0, 0, bfu, taps=True)
            assert fo.shape == (23, fsz)
            assert np.array_equal(fo, fr), (name, bfu)
            for k in TAP_DTYPE.names:
                if k != "tonal_pos":
                    assert np.array_equal(bits(to[k]), bits(tr[k])), (name, bfu, k)
    pcm = pcm_stress(34)
    for bfu in (0, 5, 32):
        assert np.array_equal(oracle.encode(pcm, br, 1, 1, bfu)[0], r.encode(pcm, br, 1, 1, bfu)[0]), bfu


@pytest.mark.parametrize("br", [66150, 93713])
@pytest.mark.parametrize("name", ["burst", "mix", "tones", "silence", "noise"])
def test_mono_joint_stereo(oracle, name, br):
    """One input channel in a joint-stereo container: the lambda appends an empty second element (one subband, no
    scaled blocks, atrac3denc.cpp:843-849), CalcMSBytesShift gives the M unit every byte it can (atrac3_bitstream.cpp
    :745-747) and the S unit is the fixed 33-bit sequence TConfigure / TAlloc produce for empty ScaledBlocks."""
    mono = np.ascontiguousarray(SIGNALS[name](40)[:, :, :1])
    for ng, nt, bfu in ((0, 0, 0), (1, 1, 0), (0, 0, 12)):
        fo = oracle.encode(mono, br, ng, nt, bfu)[0]
        assert np.array_equal(fo, ref().encode(mono, br, ng, nt, bfu)[0]), (ng, nt, bfu)
    # the S unit, byte-reversed at the end of the frame: JS id 0b0111_1111 11, numQmf-1 = 0, no gain points, ...
    assert (fo[:, -1] == 0x7F).all() and (fo[:, -2] == 0xFC).all() and (fo[:, -3] == 0x00).all() and (fo[:, -4] == 0x04).all()


@pytest.mark.parametrize("br", [LP2, LP4])
def test_stress_signal(oracle, br):
    """Full-scale, impulsive, DC / denormal-range, chirp and hard-gated material (at3_testlib.pcm_stress)."""
    r = ref()
    pcm = pcm_stress(66)
    for ng, nt in ((0, 0), (1, 1)):
        fo, to = oracle.encode(pcm, br, ng, nt, taps=True)
        fr, tr = r.encode(pcm, br, ng, nt, taps=True)
        assert np.array_equal(fo, fr)
        for k in TAP_DTYPE.names:
            if k == "tonal_pos":
                continue
            assert np.array_equal(bits(to[k]), bits(tr[k])), k


def test_flatness_threshold_walk(oracle):
    """Inputs that close in on `flat < 0.01` (the tonal-extraction decision) from both sides, down to neighbouring f32
    inputs (at3_testlib.flatness_threshold_walk): oracle and reference take the same side everywhere."""
    pcm, closest = flatness_threshold_walk(oracle)
    assert max(closest) < 1e-7, closest
    r = ref()
    for i in range(pcm.shape[0]):
        fo, to = oracle.encode(pcm[i], LP2, 1, 0, taps=True)
        fr, tr = r.encode(pcm[i], LP2, 1, 0, taps=True)
        assert np.array_equal(fo, fr), i
        assert np.array_equal(to["n_tonal"], tr["n_tonal"]), i


def test_gain_energy_scale_stage(oracle):
    """CalcGainEnergyScale (atrac3denc.cpp:175-224) on random halves / curves / carried scales incl. 0, negative, inf, nan."""
    rng = np.random.RandomState(8)
    r = ref()
    for i in range(300):
        prev = (rng.uniform(-0.5, 0.5, 256) * rng.choice([1.0, 1e-3, 0.0, 1e-12])).astype(np.float32)
        cur = (rng.uniform(-0.5, 0.5, 256) * rng.choice([1.0, 1e-2, 0.0, 1e-11])).astype(np.float32)
        k = rng.randint(0, 8)
        level = rng.randint(0, 16, k).astype(np.int32)
        loc = np.sort(rng.randint(0, 32, k)).astype(np.int32)
        ps = np.float32(rng.choice([1.0, 0.5, 3.7, 0.0, -1.0, np.inf, np.nan]))
        a, b = oracle.gain_energy_scale(prev, cur, level, loc, ps), r.gain_energy_scale(prev, cur, level, loc, ps)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), i


def test_long_noise_soak(oracle):
    pcm = SIGNALS["noise"](400, seed=21)
    for br in (LP2, LP4):
        assert np.array_equal(oracle.encode(pcm, br)[0], ref().encode(pcm, br)[0])


def test_sort_tie_order_matches_libstdcxx(oracle):
    # QuantMantisas orders candidates with std::sort on |delta| (atrac_scale.cpp:79-83); equal keys are
    # visited in libstdc++'s introsort order, which the oracle restates. Coarse grids force many ties.
    rng = np.random.RandomState(5)
    r = ref()
    for _ in range(3000):
        n = int(rng.choice([32, 64, 128]))
        mul = float(rng.choice([1.5, 2.5, 3.5, 4.5, 7.5, 15.5, 31.5]))
        grid = int(rng.choice([8, 16, 64, 256, 1 << 20]))
        v = (np.round(rng.uniform(-0.99, 0.99, size=n) * grid) / grid).astype(np.float32)
        if rng.rand() < 0.3:
            v = np.abs(v)
        m1, e1 = oracle.quant_mantisas(v, mul, 1)
        m2, e2 = r.quant_mantisas(v, mul, 1)
        assert np.array_equal(m1, m2)
        assert e1.view(np.uint32) == e2.view(np.uint32) or (np.isnan(e1) and np.isnan(e2))


@pytest.mark.parametrize("br,nch", [(LP2, 2), (LP4, 2), (LP2, 1), (LP4, 1)])
def test_scale_diagnostics_counted_like_the_reference_prints_them(oracle, br, nch):
    """TScaler::Scale reports a block above MAX_SCALE ("Scale error") and every value it clips ("clipping") on stderr only
    (atrac_scale.cpp:150-153, 163-167). The oracle counts them (at3o_diag_counts): equal to the reference's lines on input
    above full scale, zero on ordinary material, frames equal either way. These counts are what at3hip_get_counters returns."""
    for pcm, want_some in ((pcm_hot(14), True), (SIGNALS["mix"](8), False)):
        pcm = np.ascontiguousarray(pcm[:, :, :nch])
        (fr, _), n_scale, n_clip = capture_ref_diagnostics(lambda: ref().encode(pcm, br))
        oracle_diag_counts(reset=True)
        fo, _ = oracle.encode(pcm, br)
        assert np.array_equal(fo, fr)
        assert oracle_diag_counts(reset=True) == (n_scale, n_clip)
        assert (n_scale > 0 and n_clip >= n_scale) if want_some else (n_scale == 0 and n_clip == 0)
