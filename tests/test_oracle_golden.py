"""CPU oracle (oracle/at3_oracle.c) against the golden vectors generated from the real reference
(tools/gen_golden.py). Bit-exact: integer/byte outputs equal, float outputs equal as bit patterns."""
import numpy as np
import pytest

from at3_testlib import LP2, LP4, SIGNALS


def bits(a):
    a = np.asarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def test_tables(oracle, golden_stages):
    t = oracle.tables()
    for k, v in t.items():
        assert np.array_equal(bits(v), bits(golden_stages[f"table_{k}"])), k


def test_qmf(oracle, golden_stages):
    assert np.array_equal(bits(oracle.qmf(golden_stages["qmf_in"])), bits(golden_stages["qmf_out"]))


def test_mdct512(oracle, golden_stages):
    assert np.array_equal(bits(oracle.mdct512(golden_stages["mdct512_in"])), bits(golden_stages["mdct512_out"]))


def test_mdct_with_gain(oracle, golden_stages):
    g = golden_stages
    specs, bands = oracle.mdct(g["mdct_bands_in"], g["mdct_npoints"], g["mdct_level"], g["mdct_loc"])
    assert np.array_equal(bits(specs), bits(g["mdct_specs"]))
    assert np.array_equal(bits(bands), bits(g["mdct_bands_out"]))
    ges = oracle.gain_energy_scale(g["mdct_bands_in"][0, :256], g["mdct_bands_in"][0, 256:], g["mdct_level"][0, :2],
                                   g["mdct_loc"][0, :2], 1.25)
    assert np.array_equal(bits(ges), bits(g["ges_out"]))


def test_mdct_zero_and_linearity(oracle):
    # reference property tests (atrac3denc_ut.cpp:96-123): zero in -> zero out; overlap slot rewritten
    specs, bands = oracle.mdct(np.zeros((4, 512), np.float32))
    assert not specs.any() and not bands.any()
    rng = np.random.RandomState(3)
    x = rng.uniform(-0.5, 0.5, (4, 512)).astype(np.float32)
    s1, b1 = oracle.mdct(x)
    s2, _ = oracle.mdct(x * np.float32(0.5))  # power-of-two scaling is exact in fp32
    assert np.array_equal(bits(s1 * np.float32(0.5)), bits(s2))
    assert np.array_equal(bits(b1[:, 256:]), bits(x[:, 256:]))  # new half untouched without gain


def test_upsampler_and_analyze_gain(oracle, golden_stages):
    g = golden_stages
    sig, hfr = oracle.upsample(g["up_in"])
    assert np.array_equal(bits(sig), bits(g["up_out"]))
    assert np.float32(hfr).view(np.uint32) == g["up_hfr"].view(np.uint32)
    gain, lo, hi = oracle.analyze_gain(sig[1024:3072])
    assert np.array_equal(bits(gain), bits(g["ag_gain"]))
    assert np.array_equal(bits(lo), bits(g["ag_lo"]))
    assert np.array_equal(bits(hi), bits(g["ag_hi"]))


def test_analyze_gain_step_known_answer(oracle):
    # transient_detector_ut.cpp:27-54 shape: a step signal gives exact RMS plateaus
    x = np.concatenate([np.zeros(1024), np.ones(1024)]).astype(np.float32)
    gain, lo, hi = oracle.analyze_gain(x)
    assert np.array_equal(gain, np.concatenate([np.zeros(16), np.ones(16)]).astype(np.float32))
    assert np.array_equal(lo, gain) and np.array_equal(hi, gain)


def test_calc_curve(oracle, golden_stages):
    g = golden_stages
    lv, lc, ctx = oracle.calc_curve(g["cc_env"], np.array([0.01, 0.01, 0.01], np.float32), 1.9, g["cc_env"] * 0.9,
                                    g["cc_env"] * 1.1)
    assert np.array_equal(lv, g["cc_level"]) and np.array_equal(lc, g["cc_loc"])
    assert np.array_equal(bits(ctx), bits(g["cc_ctx"]))
    assert len(lv) > 0
    # negative case (gain_processor_ut.cpp:3745-3850): a stationary envelope yields no curve
    flat = np.full(32, 0.25, np.float32)
    lv, lc, _ = oracle.calc_curve(flat, np.array([0.25, 0.25, 0.25], np.float32), 1.9, flat, flat)
    assert len(lv) == 0
    # first frame of a stream (LastLevel == 0) never emits a curve (transient_detector.cpp:316-317)
    lv, _, _ = oracle.calc_curve(g["cc_env"], np.zeros(3, np.float32), 1.9, g["cc_env"], g["cc_env"])
    assert len(lv) == 0


def test_relation_to_idx(oracle, golden_stages):
    got = [oracle.relation_to_idx_hdr(x) for x in golden_stages["rti_x"]]
    assert got == list(golden_stages["rti_y"])
    # spot values pinned by the reference's own test (atrac3denc_ut.cpp:1109-1139)
    assert oracle.relation_to_idx_hdr(1.0) == 4
    assert oracle.relation_to_idx_hdr(16.0) == 0
    assert oracle.relation_to_idx_hdr(0.5) == 5
    assert oracle.relation_to_idx_hdr(0.00048828125) == 15


def test_quant_mantisas(oracle, golden_stages):
    g = golden_stages
    for i, mul in enumerate((1.5, 2.5, 4.5, 7.5, 15.5, 31.5)):
        m, e = oracle.quant_mantisas(g["quant_in"][i], mul, 1)
        assert np.array_equal(m, g["quant_mant"][i])
        assert np.float32(e).view(np.uint32) == g["quant_err"][i].view(np.uint32)


def test_quant_energy_adaptive_property(oracle):
    # atrac_scale_ut.cpp:25-67: ea=true must not increase |e2 - e1| relative to plain rounding
    rng = np.random.RandomState(9)
    for _ in range(50):
        v = rng.uniform(-0.99, 0.99, 64).astype(np.float32)
        for mul in (2.5, 7.5, 31.5):
            _, r0 = oracle.quant_mantisas(v, mul, 0)
            _, r1 = oracle.quant_mantisas(v, mul, 1)
            assert abs(float(r1) - 1.0) <= abs(float(r0) - 1.0) + 1e-6


def test_scale_and_flatness(oracle, golden_stages):
    g = golden_stages
    sfi, en, vals = oracle.scale_frame(g["scale_in"])
    assert np.array_equal(sfi, g["scale_sfi"])
    assert np.array_equal(bits(en), bits(g["scale_energy"]))
    assert np.array_equal(bits(vals), bits(g["scale_values"]))
    assert np.array_equal(bits(oracle.flatness(g["scale_in"] ** 2)), bits(g["flat_out"]))
    # atrac_psy_common_ut.cpp:333-376: uniform spectrum ~1, single tone << noise
    assert np.allclose(oracle.flatness(np.ones(1024, np.float32)), 1.0)
    tone = np.full(1024, 1e-9, np.float32)
    tone[300] = 1.0
    assert oracle.flatness(tone)[19] < 0.01


def test_log2f(oracle, golden_stages):
    y = np.array([oracle.log2f(v) for v in golden_stages["log2f_x"]], dtype=np.float32)
    assert np.array_equal(bits(y), bits(golden_stages["log2f_y"]))


@pytest.mark.parametrize("name", sorted(SIGNALS))
@pytest.mark.parametrize("mode", ["lp2", "lp4"])
@pytest.mark.parametrize("tag", ["full", "nogain", "notonal"])
def test_encode_frames(oracle, golden_encode, name, mode, tag):
    pcm = golden_encode[f"{name}_pcm_s16"].astype(np.float32) / np.float32(32768.0)
    br = LP2 if mode == "lp2" else LP4
    frames, taps = oracle.encode(pcm, br, tag == "nogain", tag == "notonal", taps=True)
    exp = golden_encode[f"{name}_{mode}_{tag}_frames"]
    assert frames.shape == exp.shape
    assert np.array_equal(frames, exp)
    assert frames.shape[1] == (384 if mode == "lp2" else 192)
    if mode == "lp2" and name != "silence":
        assert frames[0, 0] == 0xA3  # id 0x28 << 2 | (numQmf - 1)
    if tag == "full":
        assert np.array_equal(taps["n_points"].astype(np.int8), golden_encode[f"{name}_{mode}_npoints"])
        assert np.array_equal(taps["level"].astype(np.int8), golden_encode[f"{name}_{mode}_level"])
        assert np.array_equal(taps["loc"].astype(np.int8), golden_encode[f"{name}_{mode}_loc"])
        assert np.array_equal(taps["sfi"].astype(np.int8), golden_encode[f"{name}_{mode}_sfi"])
        assert np.array_equal(bits(taps["loudness_track"]), bits(golden_encode[f"{name}_{mode}_loudness"]))


def test_generators_match_golden_pcm(golden_encode):
    # the seeded generators must reproduce the committed PCM exactly (same s16 grid)
    for name, gen in SIGNALS.items():
        pcm = gen(16)
        assert np.array_equal(np.round(pcm * 32768.0).astype(np.int16), golden_encode[f"{name}_pcm_s16"]), name


def test_streaming_equals_batch(oracle):
    # N lambda calls -> N-1 frames; first call is LOOK_AHEAD (atrac3denc.cpp:715-718)
    pcm = SIGNALS["mix"](6)
    frames, _ = oracle.encode(pcm, LP2)
    assert frames.shape == (5, 384)
    f1, _ = oracle.encode(pcm[:1], LP2)
    assert f1.shape[0] == 0
