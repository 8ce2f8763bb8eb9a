"""ATRAC1 HIP path (include/at1hip.h) against the oracle and the golden sound units.
Bit-exact: bytes equal; float taps equal as bit patterns."""
import os

import numpy as np
import pytest

from at3_testlib import AT1_MODES, SIGNALS, at1_blocks, at1_oracle_encode, pcm_mix, pcm_stress

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "at1_encode.npz"))


def _enc(**kw):
    from atracdenc_amd import At1Hip
    return At1Hip(**kw)


def _mode_kw(mode):
    auto, mask, bfu = AT1_MODES[mode]
    return dict(window_auto=auto, window_mask=mask, bfu_idx_const=bfu)


def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("mode", sorted(AT1_MODES))
@pytest.mark.parametrize("nch", [1, 2])
def test_signals_in_pieces_with_taps(oracle, mode, nch):
    """Five streams side by side, fed in pieces of 7 + 1 + 32 blocks: frames, spectra, window masks and the tracked
    loudness of every piece equal the oracle's one-shot encode."""
    from atracdenc_amd import At1Hip
    names = sorted(SIGNALS)
    pcm = np.stack([at1_blocks(SIGNALS[n](20), nch) for n in names])
    exp = [at1_oracle_encode(pcm[i], mode, taps=True) for i in range(len(names))]
    enc = _enc(n_streams=len(names), max_blocks=32, channels=nch, **_mode_kw(mode))
    pos = 0
    for n in (7, 1, 32):
        got = enc.encode(pcm[:, pos:pos + n])
        specs = enc.read_tap(At1Hip.TAP_SPECTRA, np.float32, (len(names), n, nch, 512))
        masks = enc.read_tap(At1Hip.TAP_MASKS, np.int32, (len(names), n, nch))
        loud = enc.read_tap(At1Hip.TAP_LOUDNESS, np.float32, (len(names), n))
        for i, name in enumerate(names):
            f, s, m, l = (e[pos:pos + n] for e in exp[i])
            assert np.array_equal(masks[i], m), (name, pos)
            assert np.array_equal(bits(specs[i]), bits(s)), (name, pos)
            assert np.array_equal(bits(loud[i]), bits(l)), (name, pos)
            assert np.array_equal(got[i], f), (name, pos)
        pos += n
    enc.close()


@pytest.mark.parametrize("key", sorted(k for k in GOLD.files if "_ch" in k))
def test_golden(key):
    name, ch, mode = key.split("_", 2)
    nch = int(ch[2:])
    pcm = (GOLD[f"{name}_pcm_s16"].astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    blocks = at1_blocks(pcm, nch)
    enc = _enc(n_streams=1, max_blocks=blocks.shape[0], channels=nch, **_mode_kw(mode))
    got = enc.encode(blocks[None])[0]
    enc.close()
    assert np.array_equal(got, GOLD[key])


@pytest.mark.parametrize("mode", ["auto", "short", "auto_bfu3"])
def test_stress_signal(oracle, mode):
    blocks = at1_blocks(pcm_stress(66))
    enc = _enc(n_streams=1, max_blocks=blocks.shape[0], **_mode_kw(mode))
    got = enc.encode(blocks[None])[0]
    enc.close()
    assert np.array_equal(got, at1_oracle_encode(blocks, mode))


def test_reset_and_device_pointers(oracle):
    """Device-resident PCM / output (the layout bench-style callers use) and at1hip_reset."""
    import torch
    blocks = np.stack([at1_blocks(pcm_mix(16, seed=s)) for s in (1, 2, 3)])
    exp = np.stack([at1_oracle_encode(b, "auto") for b in blocks])
    enc = _enc(n_streams=3, max_blocks=32)
    d_pcm = torch.from_numpy(blocks).cuda()
    d_out = torch.zeros((3, 32, 2, 212), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()   # (the library's streams do not wait for torch's: its copies and fills first)
    for _ in range(2):
        enc.encode_device(d_pcm.data_ptr(), 32, d_out.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), exp)
        enc.reset()
    enc.close()


def test_queued_calls(oracle):
    """AT3HIP_ASYNC + at1hip_sync: four calls of eight blocks queued back to back on device-resident buffers give the bytes of one
    synchronous call of 32 blocks (carried state across queued calls), queued calls carry no timing events, a synchronous call does."""
    import torch
    blocks = np.stack([at1_blocks(pcm_mix(16, seed=s)) for s in (4, 5)])
    exp = np.stack([at1_oracle_encode(b, "auto") for b in blocks])
    enc = _enc(n_streams=2, max_blocks=32)
    pieces = [torch.from_numpy(np.ascontiguousarray(blocks[:, 8 * i: 8 * i + 8])).cuda() for i in range(4)]
    outs = [torch.zeros((2, 8, 2, 212), dtype=torch.uint8, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()   # (the library's streams do not wait for torch's: its copies and fills first)
    for p, o in zip(pieces, outs):
        enc.encode_device(p.data_ptr(), 8, o.data_ptr(), asynchronous=True)
    enc.sync()
    got = np.concatenate([o.cpu().numpy() for o in outs], axis=1)
    assert np.array_equal(got, exp)
    assert enc.timings()["total_ms"] == 0          # queued calls carry no timing events
    enc.reset()
    o = torch.zeros((2, 32, 2, 212), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()   # (the library's streams do not wait for torch's: its copies and fills first)
    enc.encode_device(torch.from_numpy(blocks).cuda().data_ptr(), 32, o.data_ptr())
    tm = enc.timings()
    assert tm["total_ms"] > 0 and tm["front_ms"] > 0 and np.array_equal(o.cpu().numpy(), exp)
    enc.close()


def test_wide_batch(oracle):
    """96 streams x 64 blocks (the batch shape of BASELINE configs[1], in ATRAC1 sound units): every stream equals the
    oracle's encode of that stream alone - no cross-stream leakage, grid-size independent results."""
    S, nb = 96, 64
    blocks = np.stack([at1_blocks(pcm_mix(nb // 2, seed=100 + s)) for s in range(S)])
    enc = _enc(n_streams=S, max_blocks=nb)
    got = enc.encode(blocks)
    enc.close()
    for s in range(0, S, 7):
        assert np.array_equal(got[s], at1_oracle_encode(blocks[s], "auto")), s
    # determinism / size-independent property over all streams: a second context gives the same bytes
    enc = _enc(n_streams=S, max_blocks=nb)
    assert np.array_equal(enc.encode(blocks), got)
    enc.close()


def test_bad_arguments():
    from atracdenc_amd import At3HipError
    with pytest.raises(At3HipError):
        _enc(n_streams=1, channels=3)
    with pytest.raises(At3HipError):
        _enc(n_streams=1, bfu_idx_const=9)
    enc = _enc(n_streams=1, max_blocks=4)
    with pytest.raises(At3HipError):
        enc.encode(np.zeros((1, 5, 512, 2), np.float32))
    enc.close()


@pytest.mark.parametrize("nsamp,ext,nch,opts,mode", [
    (20000, "aea", 2, [], (1, 0, 0)),
    (12288, "aea", 1, ["--bfuidxconst", "3"], (1, 0, 3)),
    (9000, "raw", 2, ["--notransient=5", "--batch", "3"], (0, 5, 0)),
    (300, "aea", 2, ["--notransient"], (0, 0, 0)),
    (70000, "aea", 2, ["--batch", "7"], (1, 0, 0)),
])
def test_cli_file_parity(oracle, tmp_path, nsamp, ext, nch, opts, mode):
    """at3hipenc -e atrac1 (WAV -> AEA / raw) against the reference's container writer (oracle/_ref) fed with the
    oracle's sound units for the block sequence the reference's frame schedule produces (short-read tail included)."""
    import ctypes
    import struct
    import subprocess
    from at3_testlib import REF_SO, have_ref
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "atracdenc_amd", "at3hipenc")
    if not os.path.exists(exe):
        pytest.skip("at3hipenc not built")
    nb0 = (nsamp + 1023) // 1024 + 1
    s16 = (SIGNALS["mix"](nb0, seed=11).reshape(-1, 2)[:nsamp, :nch] * 32768).astype("<i2")
    body = np.ascontiguousarray(s16).tobytes()
    wav = str(tmp_path / "in.wav")
    fmt = struct.pack("<HHIIHH", 1, nch, 44100, 44100 * 2 * nch, 2 * nch, 16)
    open(wav, "wb").write(b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt +
                          b"data" + struct.pack("<I", len(body)) + body)
    out = str(tmp_path / ("out." + ext))
    r = subprocess.run([exe, "-e", "atrac1", "-i", wav, "-o", out, "--nostdout"] + opts, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = open(out, "rb").read()

    so = str(tmp_path / "libhostio.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I", os.path.join(root, "include"), "-o", so,
                           os.path.join(root, "tests", "host", "host_io_capi.cpp")])
    host = ctypes.CDLL(so)
    blocks = np.zeros((160, 512, nch), np.float32)
    info = (ctypes.c_uint64 * 3)()
    nb = host.at3host_wav_blocks_step(wav.encode(), blocks.ctypes.data_as(ctypes.c_void_p), 160, info, 512, 0)
    assert 1 <= nb <= 160 and info[2] == nsamp and info[0] == nch
    units = at1_oracle_encode(np.ascontiguousarray(blocks[:nb]), mode).reshape(-1, 212)
    kind = 3 if ext == "aea" else 4
    if have_ref():
        ref = ctypes.CDLL(REF_SO)
        exp_path = str(tmp_path / "exp.bin")
        buf = np.ascontiguousarray(units)
        assert ref.at3ref_write_container(kind, exp_path.encode(), buf.ctypes.data_as(ctypes.c_void_p), buf.shape[0], 212, 0,
                                          nch * nsamp // 512, nch) == 0
        assert got == open(exp_path, "rb").read()
    else:
        tail = units[1:] if ext == "aea" else units
        assert got[len(got) - tail.size:] == tail.tobytes()


def test_fuzz_slice(oracle):
    """One small round of tools/fuzz_at1_gpu.py (the long runs - 24 M sound units clean - are done by hand on the GPU box)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_at1_gpu.py"), "1", "48", "6"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "FUZZ CLEAN" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_long_call_crosses_scan_chunks(oracle):
    """600 blocks in ONE call: the loudness scan walks the units in chunks of 256 (k_at1_loud_scan), and the front
    kernel's grid grows with the block count - results must not depend on either."""
    blocks = np.stack([at1_blocks(pcm_mix(300, seed=s)) for s in (21, 22)])
    enc = _enc(n_streams=2, max_blocks=600)
    got = enc.encode(blocks)
    enc.close()
    for s in range(2):
        assert np.array_equal(got[s], at1_oracle_encode(blocks[s], "auto")), s
