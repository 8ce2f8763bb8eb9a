"""Callers and data formats either side of the hot path: container writers, the reader-driven frame
schedule and the WAV reader of atracdenc_amd/host/at3hip_io.hpp against the reference's own classes
(oracle/_ref: TOma + liboma, CreateAt3Output, CreateRawOutput, TPCMEngine). CPU only."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

from at3_testlib import REF_SO, have_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hostio") / "libat3host_io_test.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), "-o", so,
                           os.path.join(ROOT, "tests", "host", "host_io_capi.cpp")])
    return ctypes.CDLL(so)


@pytest.fixture(scope="module")
def reflib():
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    return ctypes.CDLL(REF_SO)


def _write(lib, fn, kind, path, frames, frame_sz, js, hint, nch):
    buf = np.ascontiguousarray(frames, dtype=np.uint8)
    rc = getattr(lib, fn)(kind, path.encode(), buf.ctypes.data_as(ctypes.c_void_p), buf.shape[0], frame_sz, js, hint, nch)
    assert rc == 0
    return open(path, "rb").read()


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("frame_sz,js", [(384, 0), (192, 1), (272, 0)])
@pytest.mark.parametrize("n,hint", [(7, 7), (8, 7), (3, 0), (0, 5)])
def test_container_bytes(host, reflib, tmp_path, kind, frame_sz, js, n, hint):
    frames = np.random.RandomState(kind * 100 + frame_sz + n).randint(0, 256, size=(n, frame_sz)).astype(np.uint8)
    a = _write(reflib, "at3ref_write_container", kind, str(tmp_path / "ref.bin"), frames, frame_sz, js, hint, 2)
    b = _write(host, "at3host_write_container", kind, str(tmp_path / "own.bin"), frames, frame_sz, js, hint, 2)
    assert a == b
    assert len(a) == {0: 96, 1: 76, 2: 0}[kind] + n * frame_sz


@pytest.mark.parametrize("kind", [5, 6, 7])
@pytest.mark.parametrize("nch", [1, 2])
@pytest.mark.parametrize("n,hint", [(5, 5), (6, 5), (2, 0), (0, 3)])
def test_atrac3plus_container_bytes(host, reflib, tmp_path, kind, nch, n, hint):
    """ATRAC3plus sinks of main.cpp:451-463: TOma with OMAC_ID_ATRAC3PLUS, CreateAt3POutput (WAVE_FORMAT_EXTENSIBLE + GUID,
    lengths back-filled on close), raw; 2048-byte frames."""
    frames = np.random.RandomState(kind * 10 + nch + n).randint(0, 256, size=(n, 2048)).astype(np.uint8)
    a = _write(reflib, "at3ref_write_container", kind, str(tmp_path / "ref.bin"), frames, 2048, 0, hint, nch)
    b = _write(host, "at3host_write_container", kind, str(tmp_path / "own.bin"), frames, 2048, 0, hint, nch)
    assert a == b
    assert len(a) == {5: 96, 6: 80, 7: 0}[kind] + n * 2048


def test_container_selection(host):
    for name, want in (("x.oma", 0), ("x.OMA", 0), ("x.at3", 1), ("a.b.WAV", 1), ("x.raw", 2), ("x.dat", 2), ("noext", 0), ("x.aa3", 0),
                       ("x.rm", -1)):
        assert host.at3host_select_container(name.encode()) == want, name


def _trace(lib, fn, total, nch):
    n = 64
    first = np.zeros(n, np.float32)
    last = np.zeros(n, np.float32)
    tail = np.zeros(1024 * nch, np.float32)
    processed = ctypes.c_uint64()
    f = getattr(lib, fn)
    f.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    calls = f(total, nch, first.ctypes.data, last.ctypes.data, n, ctypes.byref(processed), tail.ctypes.data)
    return calls, processed.value, first, last, tail


@pytest.mark.parametrize("nch", [1, 2])
@pytest.mark.parametrize("total", [1, 1000, 1024, 3071, 3072, 3073, 4096, 5000, 7168, 7169, 8192, 10000, 12288, 12289, 16384, 20001])
def test_frame_schedule(host, reflib, total, nch):
    """Number of lambda calls, which samples each call sees (incl. the partially cleared tail after a short read and
    the stale block of the drain call) and the final processed count equal TPCMEngine::ApplyProcess driven as in
    main.cpp:697-705."""
    a = _trace(reflib, "at3ref_engine_trace", total, nch)
    b = _trace(host, "at3host_engine_trace", total, nch)
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32))
    assert np.array_equal(a[4].view(np.uint32), b[4].view(np.uint32))


def _wav_bytes(fmt, bits, nch, data, extensible=False, junk=True):
    body = data.tobytes()
    if extensible:
        guid = struct.pack("<H", fmt) + bytes.fromhex("000000001000800000aa00389b71")
        fmtck = struct.pack("<HHIIHHHHI", 0xFFFE, nch, 44100, 44100 * nch * bits // 8, nch * bits // 8, bits, 22, bits, 3) + guid
    else:
        fmtck = struct.pack("<HHIIHH", fmt, nch, 44100, 44100 * nch * bits // 8, nch * bits // 8, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmtck)) + fmtck
    if junk:
        chunks += b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\0"      # odd-sized chunk + pad byte
    chunks += b"data" + struct.pack("<I", len(body)) + body
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


@pytest.mark.parametrize("case", ["s16", "s16_mono", "s24", "s32", "f32", "u8", "s16_ext"])
def test_wav_reader(host, tmp_path, case):
    """Sample values as sf_readf_float delivers them (integer PCM / 2^(bits-1); libsndfile is not in this image, the
    rule is its documented normalisation) and the block sequence of the engine on a real file."""
    rng = np.random.RandomState(7)
    nfr = 5000
    nch = 1 if case == "s16_mono" else 2
    if case.startswith("s16"):
        raw = rng.randint(-32768, 32768, size=(nfr, nch)).astype("<i2")
        exp = raw.astype(np.float32) / np.float32(32768)
        blob = _wav_bytes(1, 16, nch, raw, extensible=case.endswith("ext"))
    elif case == "s24":
        v = rng.randint(-(1 << 23), 1 << 23, size=(nfr, nch))
        raw = np.zeros((nfr, nch, 3), np.uint8)
        for k in range(3):
            raw[..., k] = (v >> (8 * k)) & 0xFF
        exp = (v.astype(np.float64) / (1 << 23)).astype(np.float32)
        blob = _wav_bytes(1, 24, nch, raw)
    elif case == "s32":
        raw = rng.randint(-(1 << 31), 1 << 31, size=(nfr, nch)).astype("<i4")
        exp = (raw.astype(np.float32) / np.float32(2147483648.0)).astype(np.float32)
        blob = _wav_bytes(1, 32, nch, raw)
    elif case == "f32":
        raw = rng.uniform(-1, 1, size=(nfr, nch)).astype("<f4")
        exp = raw
        blob = _wav_bytes(3, 32, nch, raw)
    else:
        raw = rng.randint(0, 256, size=(nfr, nch)).astype(np.uint8)
        exp = (raw.astype(np.float32) - 128) / np.float32(128)
        blob = _wav_bytes(1, 8, nch, raw)
    path = str(tmp_path / "in.wav")
    open(path, "wb").write(blob)
    blocks = np.zeros((16, 1024, nch), np.float32)
    info = (ctypes.c_uint64 * 3)()
    n = host.at3host_wav_blocks(path.encode(), blocks.ctypes.data_as(ctypes.c_void_p), 16, info)
    assert list(info) == [nch, 44100, nfr]
    assert n == 8                                   # 5000 samples: 4 + 4 lambda calls (first = LOOK_AHEAD), no drain
    flat = blocks[:8].reshape(-1, nch)
    assert np.array_equal(flat[:nfr].view(np.uint32), exp.view(np.uint32))
    # short second read: 904 frames, then (4096 - 904) * nch BYTES of zeros, then the first read's samples again
    nz = (4096 - 904) * nch // 4                    # floats cleared
    tail = flat[4096 + 904:].reshape(-1)
    assert not tail[:nz].any()
    stale = flat[:4096].reshape(-1)[904 * nch + nz:]
    assert np.array_equal(tail[nz:], stale)


# ---- ATRAC1 side: AEA container, 512-sample schedule without a look-ahead call ------------------
@pytest.mark.parametrize("kind", [3, 4])
@pytest.mark.parametrize("nch", [1, 2])
@pytest.mark.parametrize("n,hint", [(9, 9), (1, 1), (0, 4), (5, 70000)])
def test_aea_container_bytes(host, reflib, tmp_path, kind, nch, n, hint):
    """TAeaOutput (2048-byte header, dummy unit, first WriteFrame swallowed) and ATRAC1's raw output."""
    frames = np.random.RandomState(kind * 10 + n).randint(0, 256, size=(n, 212)).astype(np.uint8)
    a = _write(reflib, "at3ref_write_container", kind, str(tmp_path / "ref.bin"), frames, 212, 0, hint, nch)
    b = _write(host, "at3host_write_container", kind, str(tmp_path / "own.bin"), frames, 212, 0, hint, nch)
    assert a == b
    assert len(a) == (2048 + 212 + max(n - 1, 0) * 212 if kind == 3 else n * 212)


def _trace_step(lib, fn, total, nch, step, look_ahead):
    n = 96
    first = np.zeros(n, np.float32)
    last = np.zeros(n, np.float32)
    tail = np.zeros(step * nch, np.float32)
    processed = ctypes.c_uint64()
    f = getattr(lib, fn)
    f.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                  ctypes.c_void_p, ctypes.c_void_p]
    calls = f(total, nch, step, look_ahead, first.ctypes.data, last.ctypes.data, n, ctypes.byref(processed), tail.ctypes.data)
    return calls, processed.value, first, last, tail


@pytest.mark.parametrize("nch", [1, 2])
@pytest.mark.parametrize("total", [1, 511, 512, 513, 4095, 4096, 4097, 6000, 8192, 12800, 20001])
def test_frame_schedule_atrac1(host, reflib, total, nch):
    """ApplyProcess(512, ...) with a lambda that always answers PROCESSED (main.cpp:646, atrac1denc.cpp:253)."""
    a = _trace_step(reflib, "at3ref_engine_trace_step", total, nch, 512, 0)
    b = _trace_step(host, "at3host_engine_trace_step", total, nch, 512, 0)
    assert a[0] == b[0] and a[1] == b[1]
    for i in (2, 3, 4):
        assert np.array_equal(a[i].view(np.uint32), b[i].view(np.uint32))


@pytest.mark.parametrize("nch", [1, 2])
@pytest.mark.parametrize("total", [1, 2047, 2048, 2049, 4095, 4096, 4097, 6144, 8192, 10000, 12288, 20001])
def test_frame_schedule_atrac3plus(host, reflib, total, nch):
    """ApplyProcess(2048, ...) with one LOOK_AHEAD answer first (main.cpp:679-686, at3p.cpp:113-115): two calls per read."""
    a = _trace_step(reflib, "at3ref_engine_trace_step", total, nch, 2048, 1)
    b = _trace_step(host, "at3host_engine_trace_step", total, nch, 2048, 1)
    assert a[0] == b[0] and a[1] == b[1]
    for i in (2, 3, 4):
        assert np.array_equal(a[i].view(np.uint32), b[i].view(np.uint32))
