"""ATRAC3plus front end on the GPU (include/at3phip.h) against the oracle and the golden vectors.
Float results compared as bit patterns."""
import os

import numpy as np
import pytest

from at3_testlib import at3p_mdct, at3p_pqf, at3p_signal

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "at3p_frontend.npz"))
NAMES = sorted(k[:-8] for k in GOLD.files if k.endswith("_pcm_s16"))
SIGS = [("mix", 32768.0), ("noise", 1.0), ("burst", 32768.0), ("tones", 1.0), ("silence", 1.0), ("stress", 1.0)]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _pcm(nf, nch):
    """[S, F, 2048, C] from the test signals, one stream per signal."""
    return np.stack([np.stack([at3p_signal(n, nf, channel=c, scale=sc) for c in range(nch)], axis=-1) for n, sc in SIGS])


@pytest.mark.parametrize("nch", [1, 2])
@pytest.mark.parametrize("mode", ["sine", "steep", "random", "random_residual"])
def test_fused_in_pieces(oracle, nch, mode):
    """at3phip_pqf_mdct fed in 3 + 1 + 8 frame pieces: subband samples and spectra equal the oracle's one-shot run."""
    from atracdenc_amd import At3pHip
    nf = 12
    pcm = _pcm(nf, nch)
    S = pcm.shape[0]
    rng = np.random.RandomState(5)
    flags = {"sine": None, "steep": np.full((S, nf, nch), 0xFFFF, np.uint16)}.get(mode, rng.randint(0, 65536, (S, nf, nch)).astype(np.uint16))
    if mode == "sine":
        flags = None
    rs = mode == "random_residual"
    enc = At3pHip(n_streams=S, max_frames=8, channels=nch)
    parts = [enc.pqf_mdct(pcm[:, a:b], None if flags is None else flags[:, a:b], rs) for a, b in ((0, 3), (3, 4), (4, 12))]
    enc.close()
    bands = np.concatenate([p[0] for p in parts], axis=1)
    specs = np.concatenate([p[1] for p in parts], axis=1)
    for s in range(S):
        for c in range(nch):
            eb = at3p_pqf(pcm[s, :, :, c])
            x = eb if not rs else (eb.astype(np.float64) / (32768.0 / 1.122018)).astype(np.float32)
            es = at3p_mdct(x, None if flags is None else flags[s, :, c])
            assert np.array_equal(bits(bands[s, :, c]), bits(eb)), (SIGS[s][0], c)
            assert np.array_equal(bits(specs[s, :, c]), bits(es)), (SIGS[s][0], c)


def test_separate_entry_points_and_reset(oracle):
    """at3phip_pqf_analyse and at3phip_mdct on their own (host buffers), state carried per entry point, reset."""
    from atracdenc_amd import At3pHip
    pcm = _pcm(6, 2)
    S = pcm.shape[0]
    flags = np.random.RandomState(9).randint(0, 65536, (S, 6, 2)).astype(np.uint16)
    enc = At3pHip(n_streams=S, max_frames=6, channels=2)
    for _ in range(2):
        bands = np.concatenate([enc.pqf(pcm[:, :2]), enc.pqf(pcm[:, 2:])], axis=1)
        specs = np.concatenate([enc.mdct(bands[:, :5], flags[:, :5]), enc.mdct(bands[:, 5:], flags[:, 5:])], axis=1)
        for s in range(S):
            for c in range(2):
                eb = at3p_pqf(pcm[s, :, :, c])
                assert np.array_equal(bits(bands[s, :, c]), bits(eb))
                assert np.array_equal(bits(specs[s, :, c]), bits(at3p_mdct(eb, flags[s, :, c])))
        enc.reset()
    enc.close()


@pytest.mark.parametrize("name", NAMES)
def test_golden(name):
    from atracdenc_amd import At3pHip
    pcm = (GOLD[f"{name}_pcm_s16"].astype(np.float32) / np.float32(32768.0) * GOLD[f"{name}_scale"]).astype(np.float32)
    nf = pcm.shape[0]
    enc = At3pHip(n_streams=1, max_frames=nf, channels=1)
    bands, specs = enc.pqf_mdct(pcm[None, :, :, None])
    enc.reset()
    _, specs_mixed = enc.pqf_mdct(pcm[None, :, :, None], GOLD[f"{name}_flags"][None, :, None])
    enc.close()
    assert np.array_equal(bits(bands[0, :, 0]), bits(GOLD[f"{name}_bands"]))
    assert np.array_equal(bits(specs[0, :, 0]), bits(GOLD[f"{name}_specs_sine"]))
    assert np.array_equal(bits(specs_mixed[0, :, 0]), bits(GOLD[f"{name}_specs_mixed"]))


def test_wide_batch_device_pointers(oracle):
    """64 streams x 32 frames (the audio of BASELINE configs[1]) with device-resident buffers; spot streams against the
    oracle, zero input -> zero output, and power-of-two linearity over the whole batch."""
    import torch
    from atracdenc_amd import At3pHip
    S, nf = 64, 32
    rng = np.random.RandomState(11)
    pcm = (rng.randint(-20000, 20000, size=(S, nf, 2048, 2)).astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    pcm[5] = 0.0
    d_pcm = torch.from_numpy(pcm).cuda()
    d_specs = torch.zeros((S, nf, 2, 2048), dtype=torch.float32, device="cuda")
    enc = At3pHip(n_streams=S, max_frames=nf, channels=2)
    enc.pqf_mdct_device(d_pcm.data_ptr(), nf, d_specs.data_ptr())
    torch.cuda.synchronize()
    specs = d_specs.cpu().numpy()
    assert not specs[5].any()
    for s in (0, 17, 63):
        for c in range(2):
            assert np.array_equal(bits(specs[s, :, c]), bits(at3p_mdct(at3p_pqf(pcm[s, :, :, c]))))
    enc.reset()
    d_half = torch.from_numpy(pcm * np.float32(0.5)).cuda()
    enc.pqf_mdct_device(d_half.data_ptr(), nf, d_specs.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_specs.cpu().numpy()), bits(specs * np.float32(0.5)))
    enc.close()


def test_bad_arguments():
    from atracdenc_amd import At3HipError, At3pHip
    with pytest.raises(At3HipError):
        At3pHip(n_streams=1, channels=3)
    enc = At3pHip(n_streams=1, max_frames=2, channels=1)
    with pytest.raises(At3HipError):
        enc.pqf(np.zeros((1, 3, 2048, 1), np.float32))
    enc.close()


# ---- frame writer without tonal block (at3phip_write_frames / at3phip_encode_frames) ---------------------------------
from at3_testlib import at3p_specs, at3p_write_frames  # noqa: E402

FRAMES_GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "at3p_frames.npz"))
FRAME_NAMES = sorted(k[:-6] for k in FRAMES_GOLD.files if k.endswith("_specs"))


@pytest.mark.parametrize("name", FRAME_NAMES)
def test_write_frames_golden(name):
    """Frames written by the real reference (tools/gen_golden_at3p_frames.py), sine windows and mixed window flags."""
    from atracdenc_amd import At3pHip
    sp, fl = FRAMES_GOLD[f"{name}_specs"], FRAMES_GOLD[f"{name}_flags"]
    nf, nch, _ = sp.shape
    enc = At3pHip(n_streams=1, max_frames=nf, channels=nch)
    assert np.array_equal(enc.write_frames(sp[None])[0], FRAMES_GOLD[f"{name}_frames_sine"])
    assert np.array_equal(enc.write_frames(sp[None], fl[None])[0], FRAMES_GOLD[f"{name}_frames_flags"])
    enc.close()


@pytest.mark.parametrize("nch", [1, 2])
def test_write_frames_vs_oracle(oracle, nch):
    """One stream per test signal plus white spectra from silence to clipping; random window flags."""
    from atracdenc_amd import At3pHip
    nf = 8
    rng = np.random.RandomState(21 + nch)
    streams = [at3p_specs(n, nf, nch) for n in ("noise", "tones", "burst", "mix", "silence", "stress")]
    streams += [(lvl * rng.standard_normal((nf, nch, 2048))).astype(np.float32) for lvl in (1e-7, 1e-3, 0.1, 1.0, 4.0)]
    specs = np.stack(streams)
    S = specs.shape[0]
    flags = rng.randint(0, 65536, (S, nf, nch)).astype(np.uint16)
    flags[:, 0] = 0
    flags[:, 1] = 0xFFFF
    flags[:, 2] = 0x00FF
    enc = At3pHip(n_streams=S, max_frames=nf, channels=nch)
    got_sine = enc.write_frames(specs)
    got_flags = enc.write_frames(specs, flags)
    enc.close()
    for s in range(S):
        assert np.array_equal(got_sine[s], at3p_write_frames(specs[s])), s
        assert np.array_equal(got_flags[s], at3p_write_frames(specs[s], flags[s])), s


@pytest.mark.parametrize("nch", [1, 2])
def test_encode_frames_in_pieces(oracle, nch):
    """at3phip_encode_frames fed 3 + 1 + 8 frames: PCM to frames equals the oracle's PQF, residual scale, MDCT, writer."""
    from atracdenc_amd import At3pHip
    nf = 12
    names = ("mix", "noise", "burst", "tones")
    pcm = np.stack([np.stack([at3p_signal(n, nf, channel=c) for c in range(nch)], axis=-1) for n in names])
    enc = At3pHip(n_streams=len(names), max_frames=8, channels=nch)
    got = np.concatenate([enc.encode_frames(pcm[:, a:b]) for a, b in ((0, 3), (3, 4), (4, 12))], axis=1)
    t = enc.timings()
    enc.close()
    assert t["write_ms"] > 0.0
    for s, n in enumerate(names):
        assert np.array_equal(got[s], at3p_write_frames(at3p_specs(n, nf, nch))), n


def test_write_frames_full_batch(oracle):
    """A 64-stream x 32-frame batch of stereo spectra at mixed levels: every frame equals the oracle's; every frame starts
    with the zero bit and the channel-block type, and codes at most 32 quant units."""
    from atracdenc_amd import At3pHip
    S, nf = 64, 32
    rng = np.random.RandomState(77)
    level = np.exp(rng.uniform(np.log(1e-6), np.log(2.0), size=(S, nf, 1, 1))).astype(np.float32)
    specs = (level * rng.standard_normal((S, nf, 2, 2048))).astype(np.float32)
    tilt = np.exp(-np.arange(2048, dtype=np.float32) / rng.uniform(100, 3000, size=(S, 1, 1, 1)).astype(np.float32))
    specs = (specs * tilt).astype(np.float32)
    enc = At3pHip(n_streams=S, max_frames=nf, channels=2)
    got = enc.write_frames(specs)
    enc.close()
    exp = at3p_write_frames(specs.reshape(S * nf, 2, 2048)).reshape(S, nf, 2048)
    bad = (got != exp).any(axis=2)
    assert not bad.any(), np.argwhere(bad)[:8].tolist()
    assert np.all(got[:, :, 0] >> 5 == 1)          # 0, then channels - 1 = 1 in two bits
    units = (got[:, :, 0] & 0x1F) + 1
    assert units.max() <= 32 and units.min() >= 1 and len(np.unique(units)) > 2


def test_encode_frames_async(oracle):
    """Six asynchronous at3phip_encode_frames calls in flight (the writer of a call on its own stream beside the next
    call's filter bank and transform), one at3phip_sync: the frames equal the oracle's for the whole PCM sequence; a
    synchronous entry point right after an asynchronous call waits for it."""
    import torch
    from atracdenc_amd import At3pHip
    nf_call, calls, nch = 4, 6, 2
    names = ("mix", "noise", "burst")
    nf = nf_call * calls
    pcm = np.stack([np.stack([at3p_signal(n, nf, channel=c) for c in range(nch)], axis=-1) for n in names])
    enc = At3pHip(n_streams=len(names), max_frames=nf_call, channels=nch)
    d_pcm = torch.from_numpy(pcm).cuda()
    outs = [torch.zeros((len(names), nf_call, 2048), dtype=torch.uint8, device="cuda") for _ in range(calls)]
    parts = [d_pcm[:, k * nf_call:(k + 1) * nf_call].contiguous() for k in range(calls)]
    torch.cuda.synchronize()
    for k in range(calls):
        enc.encode_frames_device(parts[k].data_ptr(), nf_call, outs[k].data_ptr(), asynchronous=True)
    enc.sync()
    got = np.concatenate([o.cpu().numpy() for o in outs], axis=1)
    for s, n in enumerate(names):
        assert np.array_equal(got[s], at3p_write_frames(at3p_specs(n, nf, nch))), n
    # asynchronous call, then a synchronous entry point without an explicit sync in between
    enc.reset()
    enc.encode_frames_device(parts[0].data_ptr(), nf_call, outs[0].data_ptr(), asynchronous=True)
    again = enc.write_frames(np.zeros((len(names), 1, nch, 2048), np.float32))
    assert np.array_equal(again[0, 0], at3p_write_frames(np.zeros((1, nch, 2048), np.float32))[0])
    enc.sync()
    assert np.array_equal(outs[0].cpu().numpy(), got[:, :nf_call])
    enc.close()
