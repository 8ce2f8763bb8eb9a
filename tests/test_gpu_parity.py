"""GPU parity tests: the HIP path (through the C ABI, include/at3hip.h) against the CPU oracle and the
committed golden vectors. Bit-exact for frame bytes (integer work); spectra are compared as bit patterns
too (the kernels keep the reference's fp32 operation order, contraction off), which is stricter than the
1e-6 tolerance the reference's own MDCT tests use (atrac3denc_ut.cpp:96-1036)."""
import numpy as np
import pytest

from at3_testlib import LP2, LP4, SIGNALS, capture_ref_diagnostics, have_ref, oracle_diag_counts, pcm_hot, pcm_stress, ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import atracdenc_amd
    return atracdenc_amd


def oracle_frames(oracle, pcm, br, ng=0, nt=0):
    return np.stack([oracle.encode(pcm[i], br, ng, nt)[0] for i in range(pcm.shape[0])])


@pytest.mark.parametrize("br", [LP2, LP4])
@pytest.mark.parametrize("opts", [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_frames_all_signals(hip, oracle, br, opts):
    ng, nt = opts
    nb = 20
    names = sorted(SIGNALS)
    pcm = np.stack([SIGNALS[n](nb) for n in names])
    enc = hip.At3Hip(n_streams=len(names), max_blocks=nb, bitrate=br, no_gain=ng, no_tonal=nt)
    got = enc.encode(pcm)
    enc.close()
    exp = oracle_frames(oracle, pcm, br, ng, nt)
    assert got.shape == exp.shape
    bad = np.argwhere((got != exp).any(axis=2))
    assert bad.size == 0, f"mismatching (stream, frame): {bad[:10].tolist()} of {names}"


CONTAINER_ROWS = [(66150, 192), (93713, 272), (104738, 304), (132300, 384), (146081, 424), (176400, 512), (264600, 768), (352800, 1024)]


@pytest.mark.parametrize("br,fsz", CONTAINER_ROWS)
def test_container_rows_and_bfu_idx_const(hip, oracle, br, fsz):
    """Every container row at3hip_create accepts (atrac3.h:211-220) x BfuIdxConst in {0, 1, 8, 20, 32}
    (atrac3_bitstream.cpp:567-585, 646); the oracle is pinned against the reference on the same settings
    (test_oracle_vs_ref.py::test_container_rows_and_bfu_idx_const)."""
    nb = 24
    names = ["burst", "mix", "tones", "noise"]
    pcm = np.stack([SIGNALS[n](nb) for n in names] + [pcm_stress(nb, seed=5)])
    for bfu in (0, 1, 8, 20, 32):
        enc = hip.At3Hip(n_streams=pcm.shape[0], max_blocks=nb, bitrate=br, bfu_idx_const=bfu)
        assert enc.frame_size == fsz
        got = enc.encode(pcm)
        enc.close()
        exp = np.stack([oracle.encode(pcm[i], br, 0, 0, bfu)[0] for i in range(pcm.shape[0])])
        bad = np.argwhere((got != exp).any(axis=2))
        assert bad.size == 0, f"bfu_idx_const={bfu}: mismatching (stream, frame): {bad[:10].tolist()}"
    enc = hip.At3Hip(n_streams=pcm.shape[0], max_blocks=nb, bitrate=br, bfu_idx_const=5, no_gain=1, no_tonal=1)
    got = enc.encode(pcm)
    enc.close()
    assert np.array_equal(got, np.stack([oracle.encode(pcm[i], br, 1, 1, 5)[0] for i in range(pcm.shape[0])]))


@pytest.mark.parametrize("br", [66150, 93713])
def test_mono_joint_stereo(hip, oracle, br):
    """One input channel, joint-stereo container: M unit = the mono unit with the maximum byte shift, S unit = the empty
    element of atrac3denc.cpp:843-849 (oracle pinned against the reference: test_oracle_vs_ref.py::test_mono_joint_stereo)."""
    nb = 30
    names = ["burst", "mix", "tones", "noise", "silence"]
    pcm = np.stack([np.ascontiguousarray(SIGNALS[n](nb)[:, :, :1]) for n in names])
    for ng, nt, bfu in ((0, 0, 0), (1, 1, 0), (0, 0, 12)):
        enc = hip.At3Hip(n_streams=len(names), max_blocks=16, bitrate=br, channels=1, no_gain=ng, no_tonal=nt, bfu_idx_const=bfu)
        got = np.concatenate([enc.encode(pcm[:, :7]), enc.encode(pcm[:, 7:23]), enc.encode(pcm[:, 23:])], axis=1)
        enc.close()
        exp = np.stack([oracle.encode(pcm[i], br, ng, nt, bfu)[0] for i in range(len(names))])
        bad = np.argwhere((got != exp).any(axis=2))
        assert bad.size == 0, f"{(ng, nt, bfu)}: mismatching (stream, frame): {bad[:10].tolist()}"


@pytest.mark.parametrize("br", [LP2, LP4])
def test_stress_signal(hip, oracle, br):
    """Full-scale noise and square waves (scale-factor clamp, +-0.99999 clip), impulses, DC with denormal-range
    energies, a chirp to Nyquist, hard-gated bursts (largest gain-curve swings), one silent channel."""
    nb = 66
    pcm = np.stack([pcm_stress(nb, seed=3), pcm_stress(nb, seed=4)[::-1].copy()])
    for ng, nt in ((0, 0), (1, 1)):
        enc = hip.At3Hip(n_streams=2, max_blocks=nb, bitrate=br, no_gain=ng, no_tonal=nt)
        got = enc.encode(pcm)
        enc.close()
        exp = oracle_frames(oracle, pcm, br, ng, nt)
        bad = np.argwhere((got != exp).any(axis=2))
        assert bad.size == 0, f"mismatching (stream, frame): {bad[:10].tolist()}"


@pytest.mark.parametrize("br", [LP2, LP4])
def test_stage_taps(hip, oracle, br):
    """Stage outputs through at3hip_read_tap against the oracle's taps: gain curves, energy
    scales, scale-factor indices, BFU energies, per-channel and tracked loudness, tonal blocks - bit patterns."""
    from atracdenc_amd import binding as B
    nb = 30
    names = ["burst", "mix", "tones"]
    pcm = np.stack([SIGNALS[n](nb) for n in names])
    S, F = len(names), nb - 1
    enc = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=br)
    enc.encode(pcm)
    psy = enc.read_tap(B.TAP_PSY, B.At3Hip.PSY_DTYPE, (S, F, 2))
    loud = enc.read_tap(B.TAP_LOUDNESS, np.float32, (S, F))
    curves = enc.read_tap(B.TAP_CURVES, np.uint8, (S, nb, 2, 4, 16))
    ges = enc.read_tap(B.TAP_ENERGY_SCALE, np.float32, (S, nb, 2, 4))
    enc.close()
    u32 = lambda a: np.ascontiguousarray(a).view(np.uint32)
    for i in range(S):
        _, tap = oracle.encode(pcm[i], br, taps=True)
        assert np.array_equal(psy[i]["sfi"], tap["sfi"].astype(np.uint8))
        assert np.array_equal(u32(psy[i]["energy"]), u32(tap["energy"]))
        assert np.array_equal(u32(psy[i]["loud_ch"]), u32(tap["loudness_ch"]))
        assert np.array_equal(u32(loud[i]), u32(tap["loudness_track"][:, 0] if tap["loudness_track"].ndim > 1 else tap["loudness_track"]))
        assert np.array_equal(psy[i]["n_tonal"], tap["n_tonal"])
        assert np.array_equal(curves[i, 1:, :, :, 0], tap["n_points"].astype(np.uint8))          # frame f at index f
        assert np.array_equal(u32(ges[i, 1:]), u32(tap["ges_frame"]))
        for f, ch in zip(*np.nonzero(tap["n_points"].sum(axis=2))):
            for b in range(4):
                n = int(tap["n_points"][f, ch, b])
                assert np.array_equal(curves[i, f + 1, ch, b, 1:1 + n], tap["level"][f, ch, b, :n].astype(np.uint8))
                assert np.array_equal(curves[i, f + 1, ch, b, 8:8 + n], tap["loc"][f, ch, b, :n].astype(np.uint8))


@pytest.mark.parametrize("br", [LP2, LP4])
def test_gain_analysis_forms_agree(hip, oracle, br):
    """AT3HIP_OPT_GAIN_FORM: the upsampler / AnalyzeGain kernel as two-wavefront workgroups (default) and as one wavefront
    per item (k_gain_analysis1: leaves in registers, v_permlane swaps across the rows) - same curves, same frames, both
    equal to the oracle's, on the signals that drive the gain-control path."""
    from atracdenc_amd import binding as B
    nb = 24
    names = ["burst", "mix", "noise", "stress"]
    sig = dict(SIGNALS, stress=lambda n: pcm_stress(n, seed=7))
    pcm = np.stack([sig[n](nb) for n in names])
    exp = oracle_frames(oracle, pcm, br)
    got = {}
    for form in (B.GAIN_FORM_TWO_WAVES, B.GAIN_FORM_ONE_WAVE):
        enc = hip.At3Hip(n_streams=len(names), max_blocks=nb, bitrate=br)
        enc.set_option(B.OPT_GAIN_FORM, form)
        frames = np.concatenate([enc.encode(pcm[:, :7]), enc.encode(pcm[:, 7:])], axis=1)   # two calls: carried context
        curves = enc.read_tap(B.TAP_CURVES, np.uint8, (len(names), nb - 7, 2, 4, 16))
        enc.close()
        assert np.array_equal(frames, exp), form
        got[form] = curves
    assert np.array_equal(got[B.GAIN_FORM_TWO_WAVES], got[B.GAIN_FORM_ONE_WAVE])
    assert got[B.GAIN_FORM_TWO_WAVES][..., 0].any()          # the material does produce gain curves


@pytest.mark.parametrize("br", [LP2, LP4])
@pytest.mark.parametrize("channels", [2, 1])
def test_s16_entry_point(hip, oracle, br, channels):
    """at3hip_encode_s16: 16-bit samples converted s / 32768.0f on the device (what sf_readf_float gives the reference for a
    16-bit WAV) == at3hip_encode on those floats == the oracle, byte for byte; host and device pointers, calls of both kinds
    alternating on one context (shared stream state), full-scale and -32768 samples included."""
    import torch
    nb = 12
    rng = np.random.RandomState(21)
    base = np.stack([np.round(np.clip(SIGNALS[n](nb), -1.0, 32767.0 / 32768.0) * 32768.0).astype(np.int16) for n in ("burst", "mix", "tones")])
    wild = rng.randint(-32768, 32768, size=base[:1].shape).astype(np.int16)
    wild[0, 0, :4] = [[-32768, 32767], [32767, -32768], [0, -1], [1, 0]]
    s16 = np.concatenate([base, wild])[..., :channels].copy()
    f32 = (s16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    S = s16.shape[0]
    exp = np.stack([oracle.encode(f32[i], br)[0] for i in range(S)]) if channels == 2 else None   # (one-channel contexts: see test_mono_*)
    a = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=br, channels=channels)
    ref = a.encode(f32)
    a.close()
    if exp is not None:
        assert np.array_equal(ref, exp)
    b = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=br, channels=channels)
    got = b.encode_s16(s16)                                  # host pointer
    b.close()
    assert np.array_equal(got, ref)
    c = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=br, channels=channels)
    d_s16 = torch.from_numpy(s16).cuda()
    fsz = c.frame_size
    outs = []
    for lo, hi, kind in ((0, 3, "f32"), (3, 7, "s16dev"), (7, 8, "s16"), (8, 12, "s16dev")):   # alternating kinds, one stream state
        if kind == "f32":
            outs.append(c.encode(f32[:, lo:hi]))
        elif kind == "s16":
            outs.append(c.encode_s16(s16[:, lo:hi]))
        else:
            piece = d_s16[:, lo:hi].contiguous()
            d_out = torch.zeros((S, hi - lo, fsz), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()   # (the library's streams do not wait for torch's: its copies and fills first)
            n = c.encode_device_s16(piece.data_ptr(), hi - lo, d_out.data_ptr())
            outs.append(d_out[:, :n].cpu().numpy())
    c.close()
    assert np.array_equal(np.concatenate(outs, axis=1), ref)


def _raw_spectra(hip, pcm, br=LP2):
    """Spectra BEFORE the tonal lines are removed: the same stream with NoTonalComponents (the spectra tap is untouched)."""
    from atracdenc_amd import binding as B
    S, nb = pcm.shape[0], pcm.shape[1]
    enc = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=br, no_gain=True, no_tonal=True)
    enc.encode(pcm)
    specs = enc.read_tap(B.TAP_SPECTRA, np.float32, (S, nb - 1, 2, 1024))
    enc.close()
    return specs


def _check_flat_values(oracle, psy, specs):
    """PSY tap flat[8..28] == CalcSpectralFlatnessPerBfu of the same spectra (oracle = glibc's log / exp), as bit patterns."""
    n = 0
    for idx in np.ndindex(specs.shape[:3]):
        exp = oracle.flatness(specs[idx] * specs[idx])
        got = psy[idx]["flat"]
        assert np.array_equal(got[8:29].view(np.uint32), exp[8:29].view(np.uint32)), (idx, got[8:29], exp[8:29])
        assert not got[:8].any() and not got[29:].any()
        n += 21
    return n


@pytest.mark.parametrize("literal", [0, 1])
def test_flatness_values(hip, oracle, literal):
    """CalcSpectralFlatnessPerBfu (atrac_psy_common.cpp:158-199) by VALUE: the f32 the device hands to `flat < 0.01f` equals
    the reference's for every BFU the extraction looks at - in the default form (one log of the lines' mantissa product,
    guarded by an error bound, literal fall-back) and with AT3HIP_OPT_LITERAL_FORMS (a restated glibc log per line, the
    reference's ordered sums, the restated exp: atracdenc_amd/csrc/at3_libm64.hpp). Frames equal either way."""
    from atracdenc_amd import binding as B
    nb = 12
    names = sorted(SIGNALS)
    pcm = np.stack([SIGNALS[n](nb) for n in names] + [pcm_stress(nb, seed=2)])
    S = pcm.shape[0]
    specs = _raw_spectra(hip, pcm)
    enc = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=LP2, no_gain=True)
    enc.set_option(B.OPT_LITERAL_FORMS, literal)
    got = enc.encode(pcm)
    psy = enc.read_tap(B.TAP_PSY, B.At3Hip.PSY_DTYPE, (S, nb - 1, 2))
    enc.close()
    assert _check_flat_values(oracle, psy, specs) == S * (nb - 1) * 2 * 21
    assert np.array_equal(got, oracle_frames(oracle, pcm, LP2, 1, 0))


@pytest.mark.parametrize("literal", [0, 1])
def test_flatness_threshold_adversarial(hip, oracle, literal):
    """The `flat < 0.01` tonal decision (atrac3denc.cpp:606) on ~150 inputs that close in on a threshold crossing of four
    different BFUs from both sides, down to neighbouring f32 PCM inputs whose flatness lies within 1e-7 of 0.01: flatness
    VALUES equal the oracle's bit for bit, and so do tonal-block counts and frame bytes - in both forms of the measure."""
    from atracdenc_amd import binding as B
    from at3_testlib import flatness_threshold_walk
    pcm, closest = flatness_threshold_walk(oracle)
    assert max(closest) < 1e-7, closest            # the walk really ends at the threshold (relative distance < 1e-5)
    S, nb = pcm.shape[0], pcm.shape[1]
    specs = _raw_spectra(hip, pcm)
    enc = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=LP2, no_gain=True)
    enc.set_option(B.OPT_LITERAL_FORMS, literal)
    got = enc.encode(pcm)
    psy = enc.read_tap(B.TAP_PSY, B.At3Hip.PSY_DTYPE, (S, nb - 1, 2))
    enc.close()
    _check_flat_values(oracle, psy, specs)
    for i in range(S):
        frames, tap = oracle.encode(pcm[i], LP2, 1, 0, taps=True)
        assert np.array_equal(psy[i]["n_tonal"], tap["n_tonal"]), i
        assert np.array_equal(got[i], frames), i


@pytest.mark.parametrize("br", [LP2, LP4])
def test_dense_tonal_material(hip, oracle, br):
    """ExtractTonalComponents / MapTonalComponents (atrac3denc.cpp:606-662) where the wavefront-parallel form has its corners: all 21 BFUs
    tonal (105 extracted values: both halves of the position list), runs that continue across BFU boundaries (components of seven
    positions and the split behind them). Every field of every tonal block, the counts and the frames equal the oracle's."""
    from atracdenc_amd import binding as B
    from at3_testlib import pcm_dense_tonal
    nb = 8
    pcm = np.stack([pcm_dense_tonal(nb, 7), pcm_dense_tonal(nb, 8, amp=0.3), pcm_dense_tonal(nb, 9, amp=0.002)])
    S = pcm.shape[0]
    for ng in (1, 0):
        enc = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=br, no_gain=ng)
        got = enc.encode(pcm)
        psy = enc.read_tap(B.TAP_PSY, B.At3Hip.PSY_DTYPE, (S, got.shape[1], 2))
        enc.close()
        seen_values, seen_len7 = 0, False
        for i in range(S):
            frames, tap = oracle.encode(pcm[i], br, ng, 0, taps=True)
            assert np.array_equal(got[i], frames), (ng, i)
            assert np.array_equal(psy[i]["n_tonal"], tap["n_tonal"]), (ng, i)
            for f in range(tap["n_tonal"].shape[0]):
                for ch in range(2):
                    n = int(tap["n_tonal"][f, ch])
                    blk = psy[i, f, ch]["tonal"][:n]
                    assert np.array_equal(blk["pos"], tap["tonal_pos"][f, ch, :n]), (ng, i, f, ch)
                    assert np.array_equal(blk["len"], tap["tonal_len"][f, ch, :n]), (ng, i, f, ch)
                    assert np.array_equal(blk["sfi"], tap["tonal_sfi"][f, ch, :n]), (ng, i, f, ch)
                    assert np.array_equal(np.ascontiguousarray(blk["values"]).view(np.uint32),
                                          np.ascontiguousarray(tap["tonal_values"][f, ch, :n, :7]).view(np.uint32)), (ng, i, f, ch)
                    seen_values = max(seen_values, int(tap["tonal_len"][f, ch, :n].sum()))
                    seen_len7 = seen_len7 or bool((tap["tonal_len"][f, ch, :n] == 7).any())
        assert seen_values > 64 and seen_len7      # the material reaches the corners it was made for


def test_high_freq_ratio_short_form(hip, oracle):
    """highFreqRatio (transient_spectral_upsampler.cpp:99-118) gates the gain analysis (`< 0.05`, `< 0.3`). k_gain_spec adds the
    two f64 energy sums in lane order and keeps the f32 of the quotient only when an error bound says the reference's 257-term
    chains round to the same f32; AT3HIP_OPT_LITERAL_FORMS walks the chains for every item. The ratios of both forms are
    equal bit for bit over every item of a mixed batch, and the frames equal the oracle's in both forms."""
    from atracdenc_amd import binding as B
    nb = 10
    names = sorted(SIGNALS)
    pcm = np.stack([SIGNALS[n](nb) for n in names] + [pcm_stress(nb, seed=s) for s in range(3, 27)])
    S = pcm.shape[0]
    want = oracle_frames(oracle, pcm, LP2, 0, 0)
    ratios = []
    for literal in (0, 1):
        enc = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=LP2)
        enc.set_option(B.OPT_LITERAL_FORMS, literal)
        got = enc.encode(pcm)
        rec = enc.read_tap(B.TAP_GAIN_ANALYSIS, np.uint32, (S, nb, 2, 3, 104))
        enc.close()
        assert np.array_equal(got, want), literal
        ratios.append(rec[:, 1:, :, :, 0].copy())
    assert np.array_equal(ratios[0], ratios[1])
    r = ratios[0].view(np.float32)
    assert (r >= 0.05).sum() > r.size // 4 and (r < 0.05).sum() > 0    # both sides of the gate occur


def test_fuzz_slice(hip, oracle):
    """A slice of tools/fuzz_gpu.py (twelve families of random material from 1-LSB dither to clipped full scale):
    the long campaign (2.9 M frames, all option sets) is run by hand on the GPU box, this keeps the generator honest."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.RandomState(4242)
    S, nb = 96, 17
    pcm = np.stack([fz.gen(rng, nb)[1] for _ in range(S)])
    for br in (LP2, LP4):
        enc = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=br)
        got = enc.encode(pcm)
        enc.close()
        for i in range(S):
            assert np.array_equal(got[i], oracle.encode(pcm[i], br)[0]), (br, i)


@pytest.mark.parametrize("mode", ["lp2", "lp4"])
@pytest.mark.parametrize("tag", ["full", "nogain", "notonal"])
def test_golden_frames(hip, golden_encode, mode, tag):
    names = sorted(SIGNALS)
    pcm = np.stack([golden_encode[f"{n}_pcm_s16"].astype(np.float32) / np.float32(32768.0) for n in names])
    enc = hip.At3Hip(n_streams=len(names), max_blocks=pcm.shape[1], bitrate=LP2 if mode == "lp2" else LP4,
                     no_gain=(tag == "nogain"), no_tonal=(tag == "notonal"))
    got = enc.encode(pcm)
    enc.close()
    for i, n in enumerate(names):
        assert np.array_equal(got[i], golden_encode[f"{n}_{mode}_{tag}_frames"]), n


@pytest.mark.parametrize("br", [LP2, LP4])
def test_streaming_pieces_equal_one_shot(hip, oracle, br):
    # stream state (QMF history, overlap, curve context, loudness) carried across calls; first block = LOOK_AHEAD
    nb = 30
    pcm = np.stack([SIGNALS["burst"](nb), SIGNALS["mix"](nb, seed=5), SIGNALS["tones"](nb)])
    enc = hip.At3Hip(n_streams=3, max_blocks=16, bitrate=br)
    outs, pos = [], 0
    for piece in (1, 1, 2, 3, 7, 16):
        outs.append(enc.encode(pcm[:, pos:pos + piece]))
        pos += piece
    assert outs[0].shape[1] == 0
    got = np.concatenate(outs, axis=1)
    assert np.array_equal(got, oracle_frames(oracle, pcm, br))
    enc.reset()
    again = enc.encode(pcm[:, :16])
    enc.close()
    assert np.array_equal(again, got[:, :15])


@pytest.mark.parametrize("br,ng", [(LP2, 0), (LP2, 1), (LP4, 0), (LP4, 1)])
def test_run_length_invariance(hip, oracle, br, ng):
    """The front-end kernels cut every (stream, channel) into runs of blocks, one wavefront each, with recomputed FIR
    histories and overlap priming at every cut (AT3HIP_OPT_RUNS overrides the automatic choice): any cut gives the same bytes."""
    from atracdenc_amd import binding as B
    nb = 13
    pcm = np.stack([SIGNALS["mix"](nb, seed=9), SIGNALS["burst"](nb, phase=700)])
    exp = oracle_frames(oracle, pcm, br, ng)
    for runs in (1, 2, 5, 12, 64):
        enc = hip.At3Hip(n_streams=2, max_blocks=nb, bitrate=br, no_gain=ng)
        enc.set_option(B.OPT_RUNS, runs)
        got = np.concatenate([enc.encode(pcm[:, :6]), enc.encode(pcm[:, 6:])], axis=1)
        enc.close()
        assert np.array_equal(got, exp), runs


@pytest.mark.parametrize("ng", [0, 1])
def test_one_long_call(hip, oracle, ng):
    """More than 4096 blocks of one stream in ONE call: a run holds at most 32 blocks, so the automatic choice needs more
    than 128 runs per (stream, channel) (it used to fall back to a single wavefront walking the whole call)."""
    nb = 4200
    pcm = SIGNALS["mix"](nb, seed=3)[None]
    enc = hip.At3Hip(n_streams=1, max_blocks=nb, bitrate=LP2, no_gain=ng)
    got = enc.encode(pcm)
    tm = enc.timings()
    enc.close()
    assert np.array_equal(got[0], oracle.encode(pcm[0], LP2, ng)[0])
    assert tm["qmf_ms"] + tm["qmf_mdct_ms"] < 5.0, tm    # a single wavefront per channel took tens of milliseconds


def test_mdct_api(hip, oracle, golden_stages):
    enc = hip.At3Hip(n_streams=1, max_blocks=2)
    g = golden_stages
    specs, bands = enc.mdct(g["mdct_bands_in"][None], g["mdct_npoints"][None], g["mdct_level"][None], g["mdct_loc"][None])
    assert np.array_equal(specs[0].view(np.uint32), g["mdct_specs"].view(np.uint32))
    assert np.array_equal(bands[0].view(np.uint32), g["mdct_bands_out"].view(np.uint32))
    rng = np.random.RandomState(1)
    n = 33
    b = rng.uniform(-0.3, 0.3, (n, 4, 512)).astype(np.float32)
    npnts = rng.randint(0, 8, (n, 4)).astype(np.int32)
    level = rng.randint(0, 16, (n, 4, 8)).astype(np.int32)
    loc = np.sort(rng.randint(0, 32, (n, 4, 8)), axis=2).astype(np.int32)
    s, bo = enc.mdct(b, npnts, level, loc)
    for i in range(n):
        es, eb = oracle.mdct(b[i], npnts[i], level[i], loc[i])
        assert np.array_equal(es.view(np.uint32), s[i].view(np.uint32)), i
        assert np.array_equal(eb.view(np.uint32), bo[i].view(np.uint32)), i
    # the maxLevels overload (atrac3denc.cpp:33-58): max |new half| after modulation, per band
    s2, bo2, mx = enc.mdct(b, npnts, level, loc, max_levels=True)
    assert np.array_equal(s2.view(np.uint32), s.view(np.uint32)) and np.array_equal(bo2.view(np.uint32), bo.view(np.uint32))
    assert np.array_equal(mx.view(np.uint32), np.abs(bo[:, :, 256:]).max(axis=2).view(np.uint32))
    # reference property tests: zero in -> zero out (atrac3denc_ut.cpp:96-123); no-gain call leaves new half alone
    s0, b0 = enc.mdct(np.zeros((1, 4, 512), np.float32))
    assert not s0.any() and not b0.any()
    s1, b1 = enc.mdct(b[:1])
    assert np.array_equal(b1[0, :, 256:], b[0, :, 256:])
    with pytest.raises(hip.At3HipError):
        enc.mdct(b[:1], np.full((1, 4), 9, np.int32), level[:1], loc[:1])
    enc.close()


def test_gain_energy_scale_api(hip, oracle):
    """at3hip_gain_energy_scale = TAtrac3MDCT::CalcGainEnergyScale (atrac3denc.cpp:175-224), against the oracle's stage
    function (pinned against the reference by test_oracle_golden / test_oracle_vs_ref) - bit patterns."""
    rng = np.random.RandomState(8)
    n = 200
    prev = (rng.uniform(-0.5, 0.5, (n, 256)) * rng.choice([1.0, 1e-3, 0.0, 1e-12], (n, 1))).astype(np.float32)
    cur = (rng.uniform(-0.5, 0.5, (n, 256)) * rng.choice([1.0, 1e-2, 0.0, 1e-11], (n, 1))).astype(np.float32)
    npnts = rng.randint(0, 8, n).astype(np.int32)
    level = rng.randint(0, 16, (n, 8)).astype(np.int32)
    loc = np.sort(rng.randint(0, 32, (n, 8)), axis=1).astype(np.int32)
    ps = rng.choice([1.0, 0.5, 3.7, 0.0, -1.0, np.inf, np.nan], n).astype(np.float32)
    enc = hip.At3Hip(n_streams=1, max_blocks=2)
    got = enc.gain_energy_scale(prev, cur, ps, npnts, level, loc)
    none = enc.gain_energy_scale(prev[:5], cur[:5], ps[:5])
    enc.close()
    for i in range(n):
        exp = oracle.gain_energy_scale(prev[i], cur[i], level[i, :npnts[i]], loc[i, :npnts[i]], ps[i])
        assert np.array_equal(got[i].view(np.uint32), exp.view(np.uint32)), (i, got[i], exp)
    for i in range(5):
        exp = oracle.gain_energy_scale(prev[i], cur[i], np.zeros(0, np.int32), np.zeros(0, np.int32), ps[i])
        assert np.array_equal(none[i].view(np.uint32), exp.view(np.uint32)), i


def _oracle_spectra(oracle, pcm):
    """[nb,1024,2] -> [nb-1, 2, 1024] spectra via the oracle's QMF + MDCT stage functions."""
    nb = pcm.shape[0]
    out = np.zeros((nb - 1, 2, 1024), np.float32)
    for ch in range(2):
        sub = oracle.qmf(np.ascontiguousarray(pcm[:, :, ch]).reshape(-1) * np.float32(0.25))
        bands = np.zeros((4, 512), np.float32)
        for f in range(nb - 1):
            bands[:, 256:] = sub[:, f * 256:(f + 1) * 256]
            specs, bands = oracle.mdct(bands)
            out[f, ch] = specs
    return out


@pytest.mark.parametrize("nb,runs,chain", [(9, 0, 0), (9, 3, 1), (9, 4, 2), (9, 8, 2), (18, 4, 2), (18, 16, 2), (18, 0, 2), (6, 4, 2), (2, 0, 0)])
def test_fused_qmf_mdct_kernel(hip, oracle, nb, runs, chain):
    """k_qmf_mdct8 alone (at3hip_qmf_mdct) against the oracle's QMF tree + MDCT, spectra as bit patterns - for the cut the library picks
    and for forced cuts: unchained runs (every run primes its own overlap) and CHAINED ones (AT3HIP_OPT_CHAIN: the runs of a workgroup
    hand the overlap on and finish their first frame after the group's rendezvous)."""
    torch = pytest.importorskip("torch")
    from atracdenc_amd import binding as B
    S = 3
    pcm = np.stack([SIGNALS["noise"](nb, seed=4), SIGNALS["mix"](nb), SIGNALS["tones"](nb)])
    enc = hip.At3Hip(n_streams=S, max_blocks=nb, no_gain=True)
    if runs: enc.set_option(B.OPT_RUNS, runs)
    if chain: enc.set_option(B.OPT_CHAIN, chain)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_specs = torch.full((S, nb - 1, 2, 1024), float("nan"), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()   # (the library's streams do not wait for torch's: its copies and fills first)
    enc.qmf_mdct_device(d_pcm.data_ptr(), nb, d_specs.data_ptr())
    got = d_specs.cpu().numpy()
    enc.close()
    for i in range(S):
        exp = _oracle_spectra(oracle, pcm[i])
        assert np.array_equal(got[i].view(np.uint32), exp.view(np.uint32)), i


def test_full_batch_config_properties(hip, oracle):
    """BASELINE config[1]: LP2 stereo, 4096 frames = 64 streams x 64 frames (+1 look-ahead block)."""
    S, nb = 64, 65
    base = [SIGNALS["noise"](nb, seed=100 + i) for i in range(4)] + [SIGNALS["mix"](nb, seed=i) for i in range(4)]
    pcm = np.stack([base[i % 8] for i in range(S)])
    enc = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=LP2)
    got = enc.encode(pcm)
    again_enc = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=LP2)
    again = again_enc.encode(pcm)
    enc.close(); again_enc.close()
    assert got.shape == (S, 64, 384)
    assert np.array_equal(got, again)                        # deterministic
    for i in range(8, S):
        assert np.array_equal(got[i], got[i % 8])             # identical streams -> identical frames (no cross-talk)
    assert (got[:, :, 0] == 0xA3).all()                       # sound-unit id 0x28 << 2 | (numQmf - 1)
    for i in range(8):                                        # every distinct stream against the oracle
        assert np.array_equal(got[i], oracle.encode(pcm[i], LP2)[0]), i


@pytest.mark.parametrize("br", [LP2, LP4])
def test_configs2_shard_properties(hip, oracle, br):
    """The per-GPU shard of BASELINE configs[2]/[3] (1M frames on 8 GPUs): 1024 streams x 128 frames, LP2 and LP4.
    Long workgroup runs (32 frames), many grid rounds; checked through size-independent properties and spot streams."""
    S, nb = 1024, 129
    base = [SIGNALS["noise"](nb, seed=7), SIGNALS["mix"](nb, seed=8), SIGNALS["burst"](nb), SIGNALS["tones"](nb),
            SIGNALS["noise"](nb, seed=11), SIGNALS["mix"](nb, seed=12), SIGNALS["silence"](nb), SIGNALS["burst"](nb, period=2500, phase=700)]
    pcm = np.stack([base[i % 8] for i in range(S)])
    enc = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=br)
    got = enc.encode(pcm)
    enc.close()
    fsz = 384 if br == LP2 else 192
    assert got.shape == (S, 128, fsz)
    for i in range(8, S, 37):
        assert np.array_equal(got[i], got[i % 8]), i           # identical streams -> identical frames, wherever they run
    for i in range(8):                                         # every distinct stream against the oracle
        assert np.array_equal(got[i], oracle.encode(pcm[i], br)[0]), i


@pytest.mark.parametrize("br", [LP2, LP4])
def test_async_pipelined_calls(hip, oracle, br):
    """AT3HIP_ASYNC: calls are only queued; the front half of a call runs beside the back half of the previous one on a
    second stream, with double-buffered spectra / curves / energy scales. Every call writes its own output buffer;
    after at3hip_sync() the concatenation must be what the synchronous path (and the oracle) produces."""
    import torch
    nb, piece = 49, 6
    names = ["burst", "mix", "tones", "noise"]
    pcm = np.stack([SIGNALS[n](nb) for n in names])
    enc = hip.At3Hip(n_streams=len(names), max_blocks=piece, bitrate=br)
    d_in, d_out, counts = [], [], []
    for pos in range(0, nb, piece):
        x = torch.from_numpy(np.ascontiguousarray(pcm[:, pos:pos + piece])).cuda()
        y = torch.zeros((len(names), piece, enc.frame_size), dtype=torch.uint8, device="cuda")
        d_in.append(x)
        d_out.append(y)
    torch.cuda.synchronize()
    for x, y in zip(d_in, d_out):
        counts.append(enc.encode_device(x.data_ptr(), x.shape[1], y.data_ptr(), asynchronous=True))
    enc.sync()
    fs, S = enc.frame_size, len(names)
    got = np.concatenate([y.cpu().numpy().reshape(-1)[: S * n * fs].reshape(S, n, fs) for y, n in zip(d_out, counts)], axis=1)   # [S][n][fs], compact
    tm = enc.timings_ago(0)
    enc.close()
    assert sum(counts) == nb - 1 and tm["qmf_mdct_ms"] > 0
    assert np.array_equal(got, oracle_frames(oracle, pcm, br))


def test_mono_input_lp2(hip, oracle):
    # SourceChannels = 1 (atrac3.h:260-277): [stream][block][1024][1] in, the one sound unit twice per frame out
    nb = 24
    names = ["burst", "mix", "tones", "noise"]
    pcm = np.stack([np.ascontiguousarray(SIGNALS[n](nb)[:, :, :1]) for n in names])
    enc = hip.At3Hip(n_streams=len(names), max_blocks=nb, bitrate=LP2, channels=1)
    got = enc.encode(pcm)
    enc.close()
    exp = np.stack([oracle.encode(pcm[i], LP2)[0] for i in range(len(names))])
    assert np.array_equal(got, exp)
    assert np.array_equal(got[:, :, :192], got[:, :, 192:])


def test_error_handling(hip):
    with pytest.raises(hip.At3HipError):
        hip.At3Hip(n_streams=0)
    enc = hip.At3Hip(n_streams=1, max_blocks=2)
    with pytest.raises(hip.At3HipError):
        enc.encode(np.zeros((1, 3, 1024, 2), np.float32))     # more blocks than max_blocks
    enc.close()


def test_host_cpp_shim(hip, oracle, tmp_path):
    """The C++ mirror of TAtrac3Encoder / TAtrac3MDCT (atracdenc_amd/host/at3hip_host.hpp) over the C ABI."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_host_shim")
    libdir = os.path.join(root, "atracdenc_amd")
    odir = os.path.join(root, "oracle")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(root, "tests", "host", "test_host_shim.cpp"), "-o", exe,
                           f"-L{libdir}", "-lat3hip", f"-L{odir}", "-lat3oracle", f"-Wl,-rpath,{libdir}",
                           f"-Wl,-rpath,{odir}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "HOST SHIM TEST OK" in out.stdout


@pytest.mark.parametrize("nsamp,ext,opts", [(20000, "oma", []), (12288, "at3", ["--bitrate", "64"]), (9000, "raw", ["--nogaincontrol"]),
                                            (8192, "oma", ["--notonal", "--batch", "3"]), (100, "at3", []), (4096, "oma", ["--mono"]),
                                            (7000, "at3", ["--mono", "--batch", "2"])])
def test_cli_file_parity(oracle, tmp_path, nsamp, ext, opts):
    """at3hipenc (WAV -> container) against: the reference's container writer (oracle/_ref) fed with the oracle's
    frames for the block sequence the reference's frame schedule produces (look-ahead call, short-read tail, drain call)."""
    import ctypes
    import os
    import struct
    import subprocess
    from at3_testlib import REF_SO, have_ref
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "atracdenc_amd", "at3hipenc")
    if not os.path.exists(exe):
        pytest.skip("at3hipenc not built")
    mono = "--mono" in opts          # test-only marker: write a one-channel WAV (the tool takes the channel count from the file)
    opts = [x for x in opts if x != "--mono"]
    nch = 1 if mono else 2
    s16 = (SIGNALS["mix"]((nsamp + 1023) // 1024 + 1, seed=11)[: (nsamp + 1023) // 1024 + 1].reshape(-1, 2)[:nsamp, :nch] * 32768).astype("<i2")
    body = np.ascontiguousarray(s16).tobytes()
    wav = str(tmp_path / "in.wav")
    fmt = struct.pack("<HHIIHH", 1, nch, 44100, 44100 * 2 * nch, 2 * nch, 16)
    open(wav, "wb").write(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt +
                          b"data" + struct.pack("<I", len(body)) + body)
    out = str(tmp_path / ("out." + ext))
    r = subprocess.run([exe, "-e", "atrac3", "-i", wav, "-o", out, "--nostdout"] + opts, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = open(out, "rb").read()

    # expected: the engine's block sequence (host IO layer, pinned against TPCMEngine in tests/test_host_io.py) ...
    so = str(tmp_path / "libhostio.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I", os.path.join(root, "include"), "-o", so,
                           os.path.join(root, "tests", "host", "host_io_capi.cpp")])
    host = ctypes.CDLL(so)
    blocks = np.zeros((64, 1024, nch), np.float32)
    info = (ctypes.c_uint64 * 3)()
    nb = host.at3host_wav_blocks(wav.encode(), blocks.ctypes.data_as(ctypes.c_void_p), 64, info)
    assert nb >= 2 and info[2] == nsamp and info[0] == nch
    br = 65536 if "--bitrate" in opts else 0
    frames = oracle.encode(blocks[:nb], br if br else LP2, int("--nogaincontrol" in opts), int("--notonal" in opts))[0]
    assert frames.shape[0] == nb - 1
    fsz = frames.shape[1]
    js = int(fsz == 192)
    kind = {"oma": 0, "at3": 1, "raw": 2}[ext]
    if have_ref():   # ... wrapped by the reference's own container code
        ref = ctypes.CDLL(REF_SO)
        exp_path = str(tmp_path / "exp.bin")
        buf = np.ascontiguousarray(frames)
        assert ref.at3ref_write_container(kind, exp_path.encode(), buf.ctypes.data_as(ctypes.c_void_p), buf.shape[0], fsz, js, nsamp // 1024, nch) == 0
        exp = open(exp_path, "rb").read()
        assert got == exp
    else:
        assert got[-frames.size:] == frames.tobytes()


@pytest.mark.parametrize("nsamp,ext,nch", [(20000, "oma", 2), (12288, "at3", 2), (9000, "raw", 1), (2048, "oma", 1), (30000, "wav", 2)])
def test_cli_atrac3plus_file_parity(oracle, tmp_path, nsamp, ext, nch):
    """at3hipenc -e atrac3plus (WAV -> container) against the reference's container writer (oracle/_ref) fed with the
    oracle's frames for the block sequence the reference's frame schedule produces: look-ahead call, then a silent frame,
    then input frame k - 2 at call k (at3p.cpp:113-160), the drain call after the last read."""
    import ctypes
    import os
    import struct
    import subprocess
    from at3_testlib import REF_SO, at3p_mdct, at3p_pqf, at3p_write_frames, have_ref
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "atracdenc_amd", "at3hipenc")
    if not os.path.exists(exe):
        pytest.skip("at3hipenc not built")
    nb1k = (nsamp + 1023) // 1024 + 1
    s16 = (SIGNALS["mix"](nb1k, seed=13)[:nb1k].reshape(-1, 2)[:nsamp, :nch] * 32768).astype("<i2")
    body = np.ascontiguousarray(s16).tobytes()
    wav = str(tmp_path / "in.wav")
    fmt = struct.pack("<HHIIHH", 1, nch, 44100, 44100 * 2 * nch, 2 * nch, 16)
    open(wav, "wb").write(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt +
                          b"data" + struct.pack("<I", len(body)) + body)
    out = str(tmp_path / ("out." + ext))
    r = subprocess.run([exe, "-e", "atrac3plus", "-i", wav, "-o", out, "--nostdout", "--batch", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = open(out, "rb").read()

    so = str(tmp_path / "libhostio.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I", os.path.join(root, "include"), "-o", so,
                           os.path.join(root, "tests", "host", "host_io_capi.cpp")])
    host = ctypes.CDLL(so)
    blocks = np.zeros((32, 2048, nch), np.float32)
    info = (ctypes.c_uint64 * 3)()
    ncalls = host.at3host_wav_blocks_step(wav.encode(), blocks.ctypes.data_as(ctypes.c_void_p), 32, info, 2048, 1)
    assert ncalls >= 1 and info[2] == nsamp and info[0] == nch
    # call k (k >= 1) writes: silence for k == 1, the frame of call k - 2's input otherwise
    specs = np.zeros((ncalls, nch, 2048), np.float32)          # row k + 1: input of call k; row 0: silence
    for c in range(nch):
        bands = at3p_pqf(np.ascontiguousarray(blocks[:ncalls - 1, :, c])) if ncalls > 1 else np.zeros((0, 16, 128), np.float32)
        if ncalls > 1:
            specs[1:, c] = at3p_mdct((bands.astype(np.float64) / (32768.0 / 1.122018)).astype(np.float32))
    frames = at3p_write_frames(specs)[: ncalls - 1]
    kind = {"oma": 5, "at3": 6, "wav": 6, "raw": 7}[ext]
    if have_ref():
        ref = ctypes.CDLL(REF_SO)
        exp_path = str(tmp_path / "exp.bin")
        buf = np.ascontiguousarray(frames)
        assert ref.at3ref_write_container(kind, exp_path.encode(), buf.ctypes.data_as(ctypes.c_void_p), buf.shape[0], 2048, 0, nsamp // 2048, nch) == 0
        assert got == open(exp_path, "rb").read()
    else:
        assert got[len(got) - frames.size:] == frames.tobytes()


def test_caller_stream_and_async_pipeline(hip, oracle):
    """at3hip_set_stream: PCM produced on the caller's HIP stream (no host synchronisation in between), eight asynchronous
    calls in flight across the context's three internal streams, one at3hip_sync at the end; then back to the context's
    own stream. Every call's frames equal the oracle's for the same PCM sequence."""
    import torch
    S, nb, calls = 6, 5, 8
    names = ["noise", "burst", "tones", "mix", "silence", "noise"]
    whole = np.stack([SIGNALS[n](calls * nb) if n != "mix" else SIGNALS["mix"](calls * nb, seed=21) for n in names])   # [S, calls * nb, 1024, 2]
    enc = hip.At3Hip(n_streams=S, max_blocks=nb, bitrate=LP2)
    side = torch.cuda.Stream()
    enc.set_stream(side.cuda_stream)
    host = torch.from_numpy(whole)
    outs = [torch.zeros((S, nb, enc.frame_size), dtype=torch.uint8, device="cuda") for _ in range(calls)]
    pcms = []
    counts = []
    with torch.cuda.stream(side):
        for k in range(calls):
            # device-side production of the call's PCM on the SAME stream the encoder will read it from
            x = host[:, k * nb:(k + 1) * nb].to("cuda", non_blocking=True)
            x = (x * 2.0 - x).contiguous()          # exact: a real kernel between the copy and the encoder
            pcms.append(x)
            counts.append(enc.encode_device(x.data_ptr(), nb, outs[k].data_ptr(), asynchronous=True))
    enc.sync()
    # a call's frames are packed [S][frames of this call][frame_size] (the first call yields one frame less: look-ahead)
    got = np.concatenate([outs[k].cpu().numpy().reshape(-1)[: S * counts[k] * enc.frame_size].reshape(S, counts[k], enc.frame_size)
                          for k in range(calls)], axis=1)
    exp = np.stack([oracle.encode(whole[s], LP2)[0] for s in range(S)])
    assert got.shape == exp.shape
    assert np.array_equal(got, exp)
    enc.set_stream(None)
    more = enc.encode(whole[:, :nb] * 0)     # the context still works on its own stream (state continues: silence after the signal)
    assert more.shape[1] == nb
    enc.close()


def test_bench_contract_and_two_context_paths(tmp_path):
    """bench.py prints ONE JSON line with the contract's fields (N = 1, a short run), and both multi-device code paths -
    one process driving two contexts from two host threads, and two torch.distributed.run ranks with a gloo barrier - run
    to completion on this box's single GPU through the --device-map test aid (their lines say so)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")

    def last_json(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=root, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    d = last_json([sys.executable, bench, "--steps", "4", "--warmup", "1", "--no-side-workloads", "--no-cpu-baseline"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True and d["value"] > 1e6
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert "workload" in d["config"]

    d2 = last_json([sys.executable, bench, "--gpus", "2", "--device-map", "0,0", "--steps", "2", "--warmup", "1", "--streams", "64",
                    "--frames", "16", "--no-side-workloads", "--no-cpu-baseline"])
    assert d2["n_gpus"] == 2 and "TEST AID" in d2["config"]["launch"] and d2["value"] > 1e5

    d3 = last_json([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29533", bench, "--gpus", "2", "--device-map", "0,0", "--steps", "2", "--warmup", "1", "--streams", "64",
                    "--frames", "16", "--no-side-workloads", "--no-cpu-baseline"])
    assert d3["n_gpus"] == 2 and "ranks" in d3["config"]["launch"] and d3["value"] > 1e5


@pytest.mark.parametrize("launch", ["one_process", "ranks"])
def test_eight_context_readiness(launch):
    """What runs the day an 8-GPU node is available, on this box's single GPU through --device-map 0,0,0,0,0,0,0,0: bench.py
    --gpus 8 as ONE process (eight contexts, eight host threads, the one_gpu_same_workload regions, the replay of every context's
    call sequence) and as eight torch.distributed.run ranks (gloo barrier, MAX of the elapsed time, checksums gathered). Every
    context checks its own shard against the oracle and replays its own sequence; eight seeds give eight distinct checksums."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    tail = [bench, "--gpus", "8", "--device-map", "0,0,0,0,0,0,0,0", "--steps", "3", "--warmup", "1", "--streams", "64", "--frames", "16",
            "--regions", "1", "--region-ms", "1", "--no-side-workloads", "--no-cpu-baseline"]
    cmd = [sys.executable] + tail if launch == "one_process" else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29541"] + tail
    def run_once(port):
        c = [a if a != "29541" else str(port) for a in cmd]
        r = subprocess.run(c, capture_output=True, text=True, cwd=root, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])
    d = run_once(29541)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["frames_per_step_all_gpus"] == 8 * 64 * 16
    assert d["parity_in_run"] is True
    ctx = d["contexts"]
    assert len(ctx) == 8 and all(c["ok"] and c["parity_check"]["mismatching_frames"] == 0 for c in ctx)
    assert all(c["parity_check"]["timed_sequence_replayed_identically"] is True for c in ctx)
    assert len({c["seed"] for c in ctx}) == 8 and len({c["checksum"] for c in ctx}) == 8
    if launch == "one_process":
        assert d["one_gpu_same_workload"]["value"] > 1e5 and sorted(c["rank"] for c in ctx) == [0] * 8
    else:
        assert sorted(c["rank"] for c in ctx) == list(range(8))


def wild_spread_pcm(nb, seed):
    """Full-scale noise below 3.5 kHz and float rounding dust above it: half of the BFUs scale near the top of the table, the
    other half at its bottom - a spread of the scale-factor indices that no ordinary material has."""
    rng = np.random.RandomState(seed)
    n = nb * 1024
    x = rng.uniform(-1, 1, size=(n, 2))
    X = np.fft.rfft(x, axis=0)
    X[int(3500.0 / 22050.0 * (n // 2)):] = 0
    y = np.fft.irfft(X, n=n, axis=0)
    return (0.9 * y / np.abs(y).max()).astype(np.float32).reshape(nb, 1024, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("ng", [0, 1])
def test_wild_scale_factor_spread(hip, oracle, ng):
    """k_alloc_pack takes TConfigure's spread from two integer reductions while the reference's float sums are provably
    exact (S2 = sum (32 sfi - m)^2 < 2^24) and runs the literal sequential loops otherwise. Ordinary signals never leave the
    fast side; this one does (checked on the PSY tap), and the frames still have to match."""
    from atracdenc_amd import binding as B
    nb = 9
    pcm = np.stack([wild_spread_pcm(nb, 5), wild_spread_pcm(nb, 6)])
    enc = hip.At3Hip(n_streams=2, max_blocks=nb, bitrate=LP2, no_gain=bool(ng))
    got = enc.encode(pcm)
    psy = enc.read_tap(B.TAP_PSY, B.At3Hip.PSY_DTYPE, (2, nb - 1, 2))
    enc.close()
    sfi = psy["sfi"].astype(np.int64).reshape(-1, 32)
    d = 32 * sfi - sfi.sum(1, keepdims=True)
    assert ((d * d).sum(1) >= (1 << 24)).any()
    for i in range(2):
        assert np.array_equal(got[i], oracle.encode(pcm[i], LP2, ng, 0)[0]), i


@pytest.mark.parametrize("br", [LP2, LP4])
@pytest.mark.parametrize("depth,s16", [(2, False), (3, False), (3, True), (4, True)])
def test_host_buffer_pipeline(hip, oracle, br, depth, s16):
    """Host PCM through the copy stream and the parity-double-buffered device staging (include/at3hip.h, "Host-buffer
    pipeline"): page-locked buffers from at3hip_host_alloc, asynchronous calls alternating between `depth` input and `depth`
    output buffers that are REFILLED / read as soon as at3hip_wait_input / at3hip_wait_frames allow (they reach three calls back)
    - any missing ordering between the copy stream, the three compute streams and the host shows up as a wrong frame. Float and
    16-bit samples."""
    nb, piece, S = 41, 5, 3
    pcm = np.stack([SIGNALS["mix"](nb, seed=31), SIGNALS["burst"](nb, phase=300), SIGNALS["noise"](nb, seed=32)])
    if s16:
        p16 = np.round(np.clip(pcm, -1.0, 32767.0 / 32768.0) * 32768.0).astype(np.int16)
        pcm = (p16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    src = p16 if s16 else pcm
    enc = hip.At3Hip(n_streams=S, max_blocks=piece, bitrate=br)
    ins = [enc.host_alloc((S, piece, 1024, 2), np.int16 if s16 else np.float32) for _ in range(depth)]
    outs = [enc.host_alloc((S, piece, enc.frame_size), np.uint8) for _ in range(depth)]
    got, counts = [], []
    calls = [(pos, min(piece, nb - pos)) for pos in range(0, nb, piece)]

    def take(k):   # the frames of call k out of its buffer
        c = counts[k]
        got.append(outs[k % depth].reshape(-1)[: S * c * enc.frame_size].reshape(S, c, enc.frame_size).copy())
        outs[k % depth][...] = 0xEE

    for k, (pos, n) in enumerate(calls):
        q = k % depth
        if k >= depth:
            enc.wait_input(depth - 1)               # call k - depth has left ins[q]
        ins[q][:, :n] = src[:, pos:pos + n]
        ins[q][:, n:] = -1 if s16 else np.nan       # (never read)
        counts.append(enc.encode_host_async(ins[q][:, :n] if n == piece else np.ascontiguousarray(ins[q][:, :n]), outs[q]))
        if k >= depth - 1:
            enc.wait_frames(depth - 1)              # the oldest call in flight: its frames are in its buffer
            take(k - (depth - 1))
    enc.sync()
    for k in range(max(0, len(calls) - (depth - 1)), len(calls)):
        take(k)
    with pytest.raises(hip.At3HipError):
        enc.wait_frames(4)                          # (the ring is four calls deep)
    for a in ins + outs:
        enc.host_free(a)
    enc.close()
    assert sum(counts) == nb - 1
    assert np.array_equal(np.concatenate(got, axis=1), oracle_frames(oracle, pcm, br))


def test_options_and_quant_tap(hip, oracle):
    """at3hip_set_option: the QUANT tap is off by default (at3hip_read_tap refuses it), on request it holds the unit cache of
    the rate loop - every (wordlen, BFU < 10) unit (the seventy small units are always quantised) with a finite energy error and
    a CLC | VLC cost - and switching it on, off or asking for another work partitioning never changes a frame."""
    from atracdenc_amd import binding as B
    nb = 9
    pcm = np.stack([SIGNALS["mix"](nb, seed=2), SIGNALS["noise"](nb, seed=3)])
    exp = oracle_frames(oracle, pcm, LP2)
    enc = hip.At3Hip(n_streams=2, max_blocks=nb, bitrate=LP2)
    assert np.array_equal(enc.encode(pcm), exp)
    with pytest.raises(hip.At3HipError):
        enc.read_tap(B.TAP_QUANT, B.At3Hip.QUANT_DTYPE, (2, nb - 1, 2))
    enc.reset()
    enc.set_option(B.OPT_QUANT_TAP, 1)
    enc.set_option(B.OPT_RUNS, 3)
    assert np.array_equal(enc.encode(pcm), exp)
    q = enc.read_tap(B.TAP_QUANT, B.At3Hip.QUANT_DTYPE, (2, nb - 1, 2))
    err, cost = q["err"][..., :10], q["cost"][..., :10]
    assert np.isfinite(err).all() and (err > 0).all()
    lines = np.array([8] * 8 + [16] * 2)
    clc = np.array([2, 3, 3, 4, 4, 5, 6])[:, None] * lines[None, :]          # CLC bits per unit: clc_len(wordlen) x lines (pairs of 4 bits at wordlen 1)
    assert ((cost & 0x1fff) == clc).all() and ((cost >> 13) > 0).all()
    # BFUs 10 .. 31: a unit the rate loop asked for has its cost (CLC bits of its wordlen and line count | VLC bits << 13) but its energy
    # error only if the energy-adaptive pass had to run for it - cost != 0 with err == 0 is a unit whose bits unit_bounds alone supplied -, and a unit
    # never asked for is zero in both
    err_hi, cost_hi = q["err"][..., 10:], q["cost"][..., 10:]
    asked = cost_hi != 0
    assert asked.any() and (~asked).any()
    assert np.isfinite(err_hi).all() and (err_hi >= 0).all() and (err_hi[~asked] == 0).all()
    lines_hi = np.array([16] * 6 + [32] * 10 + [64] * 4 + [128] * 2)
    clc_hi = np.array([2, 3, 3, 4, 4, 5, 6])[:, None] * lines_hi[None, :]
    assert ((cost_hi & 0x1fff)[asked] == np.broadcast_to(clc_hi, cost_hi.shape)[asked]).all()
    assert (asked & (err_hi == 0)).any()      # the bounds-only units exist on this material: the contract above is exercised
    enc.reset()
    enc.set_option(B.OPT_QUANT_TAP, 0)
    enc.set_option(B.OPT_RUNS, 0)
    assert np.array_equal(enc.encode(pcm), exp)
    with pytest.raises(hip.At3HipError):
        enc.set_option(99, 1)
    enc.close()


@pytest.mark.parametrize("br,channels", [(LP2, 2), (LP4, 2), (LP2, 1), (LP4, 1)])
def test_overflow_counters(hip, oracle, br, channels):
    """at3hip_get_counters: TScaler::Scale's stderr diagnostics as counters (atrac_scale.cpp:150-153 "Scale error" per block,
    :163-167 "clipping" per value). Input above full scale: frames equal the oracle's and the counters equal
    what the oracle counts - and, where the real reference is present, the lines it prints. They accumulate over calls and
    streams, a one-channel stream counts like the reference's single channel, ordinary material counts nothing, reset clears."""
    nb = 14
    pcm = np.stack([pcm_hot(nb, seed=5), pcm_hot(nb, seed=9, gain=12.0), SIGNALS["mix"](nb)])[..., :channels]
    pcm = np.ascontiguousarray(pcm)
    oracle_diag_counts(reset=True)
    exp = oracle_frames(oracle, pcm, br)
    want = oracle_diag_counts(reset=True)
    assert want[0] > 100 and want[1] >= want[0]
    if have_ref():
        _, n_scale, n_clip = capture_ref_diagnostics(lambda: [ref().encode(pcm[i], br) for i in range(pcm.shape[0])])
        assert (n_scale, n_clip) == want
    enc = hip.At3Hip(n_streams=3, max_blocks=nb, bitrate=br, channels=channels)
    assert enc.counters() == {"scale_overflow": 0, "clipped_values": 0}
    got = np.concatenate([enc.encode(pcm[:, :5]), enc.encode(pcm[:, 5:])], axis=1)     # two calls: the counters accumulate
    assert np.array_equal(got, exp)
    c = enc.counters(reset=True)
    assert (c["scale_overflow"], c["clipped_values"]) == want
    assert enc.counters() == {"scale_overflow": 0, "clipped_values": 0}
    enc.reset()
    quiet = np.ascontiguousarray(np.stack([SIGNALS[n](6) for n in ("mix", "burst", "noise")])[..., :channels])
    enc.encode(quiet)
    assert enc.counters() == {"scale_overflow": 0, "clipped_values": 0}
    enc.encode(pcm[:, :4])
    assert enc.counters()["scale_overflow"] > 0
    enc.reset()                                                                          # a fresh TAtrac3Encoder has printed nothing
    assert enc.counters() == {"scale_overflow": 0, "clipped_values": 0}
    enc.close()


def test_timing_every_samples_calls_and_changes_no_byte(hip):
    """AT3HIP_OPT_TIMING_EVERY (ABI 1.5): with N = 3 every third call with frames carries stage timings, the others report zeros and
    qmf_mdct_launches == 0; with 0 none does. The frames are the same bytes as with every call timed (the hand-overs between a call's
    streams are their own untimed events)."""
    import torch
    from atracdenc_amd import binding as B
    dev = torch.device("cuda", 0)
    S, nb, calls = 3, 8, 7
    g = torch.Generator(device="cpu").manual_seed(77)
    pcm = [((torch.rand((S, nb, 2, 1024), generator=g) - 0.5) * 1.6).to(dev) for _ in range(calls)]
    outs = {}
    for every in (1, 3, 0):
        enc = hip.At3Hip(n_streams=S, max_blocks=nb)
        enc.set_option(B.OPT_TIMING_EVERY, every)
        got, timed = [], []
        for i in range(calls):
            out = torch.zeros(S * nb * 384, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()   # (the library's streams do not wait for torch's: its copies and fills first)
            n = enc.encode_device(pcm[i].data_ptr(), nb, out.data_ptr(), asynchronous=False)
            got.append(out[: S * n * 384].cpu())
            if n:
                tm = enc.timings()
                timed.append(tm["qmf_mdct_launches"] > 0)
                assert (tm["total_ms"] > 0) == timed[-1]
        enc.close()
        outs[every] = torch.cat(got)
        if every == 1: assert all(timed)
        if every == 0: assert not any(timed)
        if every == 3: assert timed == [i % 3 == 0 for i in range(len(timed))]
    assert torch.equal(outs[1], outs[3]) and torch.equal(outs[1], outs[0])


def test_option_values_are_validated_and_version(hip):
    """at3hip_set_option rejects values outside an option's range and stores nothing; a 16-bit device pointer that
    the conversion kernel cannot read sixteen bytes at a time is refused; the library reports the ABI the binding was written for."""
    import torch
    from atracdenc_amd import binding as B
    lib = B.load_library()
    assert lib.at3hip_version() == B.AT3HIP_VERSION and B.AT3HIP_VERSION >> 16 == 1
    enc = hip.At3Hip(n_streams=1, max_blocks=4)
    for opt, bad in ((B.OPT_RUNS, -1), (B.OPT_LITERAL_FORMS, 2), (B.OPT_LITERAL_FORMS, -1), (B.OPT_QUANT_TAP, 2), (B.OPT_GAIN_FORM, 3),
                     (B.OPT_GAIN_FORM, -1), (B.OPT_GAIN_WGS_PER_CU, 17), (B.OPT_GAIN_WGS_PER_CU, 100000), (B.OPT_CHAIN, 3), (B.OPT_CHAIN, -1), (B.OPT_TIMING_EVERY, -1), (0, 0), (8, 0)):
        with pytest.raises(hip.At3HipError):
            enc.set_option(opt, bad)
    for opt, good in ((B.OPT_RUNS, 2), (B.OPT_RUNS, 0), (B.OPT_LITERAL_FORMS, 1), (B.OPT_LITERAL_FORMS, 0), (B.OPT_GAIN_FORM, B.GAIN_FORM_ONE_WAVE),
                      (B.OPT_GAIN_FORM, 2), (B.OPT_GAIN_FORM, B.GAIN_FORM_TWO_WAVES), (B.OPT_GAIN_WGS_PER_CU, 6), (B.OPT_GAIN_WGS_PER_CU, 0), (B.OPT_CHAIN, 2), (B.OPT_CHAIN, 1),
                      (B.OPT_CHAIN, 0), (B.OPT_TIMING_EVERY, 0), (B.OPT_TIMING_EVERY, 8), (B.OPT_TIMING_EVERY, 1)):
        enc.set_option(opt, good)
    assert B.OPT_FLATNESS_LITERAL == B.OPT_LITERAL_FORMS
    dev = torch.device("cuda", 0)
    raw = torch.zeros(4 * 2048 + 16, dtype=torch.int16, device=dev)
    out = torch.zeros(4 * 384, dtype=torch.uint8, device=dev)
    assert raw.data_ptr() % 16 == 0
    torch.cuda.synchronize()   # (the library's streams do not wait for torch's: its copies and fills first)
    with pytest.raises(hip.At3HipError):
        enc.encode_device_s16(raw.data_ptr() + 2, 4, out.data_ptr())      # misaligned by one sample
    assert enc.encode_device_s16(raw.data_ptr(), 4, out.data_ptr()) == 3
    enc.close()


def test_device_numa_node_query(hip):
    """at3hip_device_numa_node (ABI 1.4): the host NUMA node of a device's PCIe link from sysfs, -1 where the platform does not say;
    an ordinal the runtime does not know is -1 too (never an error: a missing pin only costs bandwidth)."""
    import os
    from atracdenc_amd import binding as B
    lib = B.load_library()
    node = lib.at3hip_device_numa_node(0)
    assert -1 <= node < 64
    if node >= 0:
        assert os.path.exists(f"/sys/devices/system/node/node{node}/cpulist")
    assert lib.at3hip_device_numa_node(1000) == -1
