"""Shared helpers for the tests, bench.py's cpu_baseline leg and tools/: ctypes bindings of the
CPU oracle (oracle/libat3oracle.so), the optional real-reference build (oracle/_ref/libat3ref.so,
only buildable where /root/reference exists) and seeded synthetic PCM generators.

TEST INFRASTRUCTURE: nothing under atracdenc_amd/ imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libat3oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libat3ref.so")

LP2 = 132300
LP4 = 66150


class Tap(ctypes.Structure):
    _fields_ = [
        ("n_points", ctypes.c_int32 * 4),
        ("level", (ctypes.c_int32 * 8) * 4),
        ("loc", (ctypes.c_int32 * 8) * 4),
        ("ges_frame", ctypes.c_float * 4),
        ("loudness_ch", ctypes.c_float),
        ("loudness_track", ctypes.c_float),
        ("sfi", ctypes.c_int32 * 32),
        ("energy", ctypes.c_float * 32),
        ("values", ctypes.c_float * 1024),
        ("n_tonal", ctypes.c_int32),
        ("tonal_pos", ctypes.c_int32 * 64),
        ("tonal_len", ctypes.c_int32 * 64),
        ("tonal_sfi", ctypes.c_int32 * 64),
        ("tonal_values", (ctypes.c_float * 8) * 64),
    ]


TAP_DTYPE = np.dtype([
    ("n_points", "<i4", (4,)), ("level", "<i4", (4, 8)), ("loc", "<i4", (4, 8)), ("ges_frame", "<f4", (4,)),
    ("loudness_ch", "<f4"), ("loudness_track", "<f4"), ("sfi", "<i4", (32,)), ("energy", "<f4", (32,)),
    ("values", "<f4", (1024,)), ("n_tonal", "<i4"), ("tonal_pos", "<i4", (64,)), ("tonal_len", "<i4", (64,)),
    ("tonal_sfi", "<i4", (64,)), ("tonal_values", "<f4", (64, 8)),
])
assert TAP_DTYPE.itemsize == ctypes.sizeof(Tap)


def build_oracle():
    """Compile oracle/libat3oracle.so (and oracle/_ref when the reference sources exist)."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    if os.path.isdir("/root/reference/src") and not os.path.exists(REF_SO):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class CpuCodec:
    """ctypes view of one of the two CPU implementations (prefix 'at3o' or 'ref')."""

    def __init__(self, path, prefix):
        self.lib = ctypes.CDLL(path)
        self.prefix = prefix
        f = self._f
        f("encode").restype = ctypes.c_int
        f("quant_mantisas").restype = ctypes.c_float
        f("quant_mantisas").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
        f("log2f").restype = ctypes.c_float
        f("log2f").argtypes = [ctypes.c_float]
        f("calc_curve").restype = ctypes.c_int
        f("calc_curve").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p]
        f("relation_to_idx_hdr").restype = ctypes.c_int
        f("relation_to_idx_hdr").argtypes = [ctypes.c_float]
        f("gain_energy_scale").argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]

    def _f(self, name):
        return getattr(self.lib, f"{self.prefix}_{name}")

    def encode(self, pcm, bitrate=LP2, no_gain=False, no_tonal=False, bfu_idx_const=0, taps=False):
        """pcm: float32 [nblocks, 1024, nch] -> (frames uint8 [nblocks-1, frame_sz], taps or None)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        nb, _, nch = pcm.shape
        out = np.zeros(max(nb - 1, 1) * 1024, dtype=np.uint8)
        fsz = ctypes.c_int()
        tap = np.zeros((max(nb - 1, 1), nch), dtype=TAP_DTYPE) if taps else None
        n = self._f("encode")(int(bitrate), nch, int(no_gain), int(no_tonal), int(bfu_idx_const), _vp(pcm), nb,
                              _vp(out), ctypes.byref(fsz), _vp(tap) if taps else None)
        frames = out[: n * fsz.value].reshape(n, fsz.value).copy()
        return frames, (tap[:n] if taps else None)

    def qmf(self, pcm_mono):
        pcm_mono = np.ascontiguousarray(pcm_mono, dtype=np.float32)
        nb = pcm_mono.size // 1024
        sub = np.zeros((4, nb * 256), dtype=np.float32)
        self._f("qmf")(_vp(pcm_mono), nb, _vp(sub))
        return sub

    def mdct(self, bands, n_points=None, level=None, loc=None):
        """bands float32 [4,512] ([overlap|new]); returns (specs[1024], mutated bands)."""
        bands = np.ascontiguousarray(bands, dtype=np.float32).copy()
        specs = np.zeros(1024, dtype=np.float32)
        if n_points is None:
            self._f("mdct")(_vp(specs), _vp(bands), None, None, None)
        else:
            n_points = np.ascontiguousarray(n_points, dtype=np.int32)
            level = np.ascontiguousarray(level, dtype=np.int32)
            loc = np.ascontiguousarray(loc, dtype=np.int32)
            self._f("mdct")(_vp(specs), _vp(bands), _vp(n_points), _vp(level), _vp(loc))
        return specs, bands

    def mdct512(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.zeros(256, dtype=np.float32)
        self._f("mdct512")(_vp(x), _vp(out))
        return out

    def gain_energy_scale(self, prev, cur, level, loc, prev_scale):
        prev = np.ascontiguousarray(prev, dtype=np.float32)
        cur = np.ascontiguousarray(cur, dtype=np.float32)
        level = np.ascontiguousarray(level, dtype=np.int32)
        loc = np.ascontiguousarray(loc, dtype=np.int32)
        out = np.zeros(4, dtype=np.float32)
        self._f("gain_energy_scale")(_vp(prev), _vp(cur), len(level), _vp(level), _vp(loc), prev_scale, _vp(out))
        return out

    def upsample(self, x512):
        x512 = np.ascontiguousarray(x512, dtype=np.float32)
        out = np.zeros(4096, dtype=np.float32)
        hfr = ctypes.c_float()
        self._f("upsample")(_vp(x512), _vp(out), ctypes.byref(hfr))
        return out, np.float32(hfr.value)

    def analyze_gain(self, x, n_points=32):
        x = np.ascontiguousarray(x, dtype=np.float32)
        g = np.zeros(n_points, dtype=np.float32)
        lo = np.zeros(n_points, dtype=np.float32)
        hi = np.zeros(n_points, dtype=np.float32)
        self._f("analyze_gain")(_vp(x), x.size, n_points, _vp(g), _vp(lo), _vp(hi))
        return g, lo, hi

    def calc_curve(self, gain, ctx, min_score, lo, hi):
        gain = np.ascontiguousarray(gain, dtype=np.float32)
        lo = np.ascontiguousarray(lo, dtype=np.float32)
        hi = np.ascontiguousarray(hi, dtype=np.float32)
        ctx = np.ascontiguousarray(ctx, dtype=np.float32).copy()
        level = np.zeros(8, dtype=np.int32)
        loc = np.zeros(8, dtype=np.int32)
        n = self._f("calc_curve")(_vp(gain), _vp(ctx), float(min_score), _vp(lo), _vp(hi), _vp(level), _vp(loc))
        return level[:n].copy(), loc[:n].copy(), ctx

    def relation_to_idx_hdr(self, x):
        return self._f("relation_to_idx_hdr")(float(x))

    def quant_mantisas(self, values, mul, ea):
        values = np.ascontiguousarray(values, dtype=np.float32)
        mant = np.zeros(values.size, dtype=np.int32)
        e = self._f("quant_mantisas")(_vp(values), values.size, float(mul), int(ea), _vp(mant))
        return mant, np.float32(e)

    def scale_frame(self, specs):
        specs = np.ascontiguousarray(specs, dtype=np.float32)
        sfi = np.zeros(32, dtype=np.int32)
        en = np.zeros(32, dtype=np.float32)
        vals = np.zeros(1024, dtype=np.float32)
        self._f("scale_frame")(_vp(specs), _vp(sfi), _vp(en), _vp(vals))
        return sfi, en, vals

    def flatness(self, energy):
        energy = np.ascontiguousarray(energy, dtype=np.float32)
        out = np.zeros(32, dtype=np.float32)
        self._f("flatness")(_vp(energy), _vp(out))
        return out

    def log2f(self, x):
        return np.float32(self._f("log2f")(float(x)))

    def tables(self):
        names = [("scale", 64), ("encwin", 256), ("gainlevel", 16), ("gaininterp", 31), ("qmfwin", 48),
                 ("loud", 1024), ("ath", 1024)]
        arrs = [np.zeros(n, dtype=np.float32) for _, n in names]
        self._f("tables")(*[_vp(a) for a in arrs])
        return {k: a for (k, _), a in zip(names, arrs)}


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        _oracle = CpuCodec(ORACLE_SO, "at3o")
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        _ref = CpuCodec(REF_SO, "ref")
    return _ref


# ----------------------------------------------------------------------------------------------
# Seeded synthetic PCM: float32 = s16 / 32768, shape [nblocks, 1024, 2]
# ----------------------------------------------------------------------------------------------
def _quant16(x):
    return (np.round(np.clip(x, -1.0, 32767.0 / 32768.0) * 32768.0) / 32768.0).astype(np.float32)


def pcm_noise(nblocks, seed=1, amp=8192):
    rng = np.random.RandomState(seed)
    return (rng.randint(-amp, amp, size=(nblocks, 1024, 2)).astype(np.float32) / np.float32(32768.0)).astype(np.float32)


def pcm_burst(nblocks, period=3000, lo=0.02, hi=0.6, freq=3000.0, right=0.5, phase=0):
    n = nblocks * 1024
    t = np.arange(n, dtype=np.float64)
    amp = np.where(((t + phase) // period) % 2 == 0, lo, hi)
    left = amp * np.sin(2 * np.pi * freq * t / 44100.0)
    x = np.stack([left, right * left], axis=-1)
    return _quant16(x).reshape(nblocks, 1024, 2)


def pcm_tones(nblocks, freqs=(440.0, 1000.0, 3000.0, 7000.0, 11000.0), amp=0.1):
    n = nblocks * 1024
    t = np.arange(n, dtype=np.float64)
    left = sum(amp * np.sin(2 * np.pi * f * t / 44100.0) for f in freqs)
    rightv = sum(amp * np.sin(2 * np.pi * f * t / 44100.0 + 0.3 * i) for i, f in enumerate(freqs))
    return _quant16(np.stack([left, rightv], axis=-1)).reshape(nblocks, 1024, 2)


def pcm_silence(nblocks):
    return np.zeros((nblocks, 1024, 2), dtype=np.float32)


def pcm_mix(nblocks, seed=7):
    """Noise bed + tones + percussive bursts: exercises gain curves, tonal extraction and allocation."""
    rng = np.random.RandomState(seed)
    n = nblocks * 1024
    t = np.arange(n, dtype=np.float64)
    bed = rng.randn(n, 2) * 0.01
    tone = 0.15 * np.sin(2 * np.pi * 2500.0 * t / 44100.0)[:, None] * np.array([1.0, 0.7])
    env = np.zeros(n)
    for start in rng.randint(0, n, size=max(1, nblocks // 2)):
        ln = rng.randint(200, 3000)
        seg = np.arange(min(ln, n - start))
        env[start:start + seg.size] += rng.uniform(0.2, 0.8) * np.exp(-seg / (0.3 * ln))
    hit = env[:, None] * rng.randn(n, 2) * 0.5
    return _quant16(bed + tone + hit).reshape(nblocks, 1024, 2)


def pcm_dense_tonal(nb, seed, amp=0.02):
    """Every BFU 8..28 tonal: clusters of eight adjacent spectral lines across ten BFU boundaries (the runs of two BFUs are consecutive
    positions and merge into components of 7 + 3) and single lines elsewhere."""
    rng = np.random.RandomState(seed)
    n = nb * 1024
    t = np.arange(n, dtype=np.float64)
    x = np.zeros((n, 2))
    lines = [k for bnd in (80, 96, 128, 176, 192, 256, 320, 384, 512, 576) for k in range(bnd - 4, bnd + 4)]
    lines += [70, 105, 118, 150, 165, 210, 240, 290, 350, 420, 460, 490, 540, 610, 680, 740, 800, 860]
    for k in lines:
        f = (k + 0.5) * 22050.0 / 1024
        a = amp * rng.uniform(0.6, 1.0)
        ph = rng.uniform(0, 6.28)
        x[:, 0] += a * np.sin(2 * np.pi * f * t / 44100 + ph)
        x[:, 1] += a * np.sin(2 * np.pi * f * t / 44100 + 1.3 * ph)
    return x.reshape(nb, 1024, 2).astype(np.float32)


def pcm_stress(nblocks, seed=3):
    """Corner cases of the arithmetic, one after another in 4-block segments: full-scale noise (scale-factor clamp at
    1.0 and the +-0.99999 clip), full-scale square wave, isolated unit impulses, DC with a tiny dither (denormal-range
    energies), a linear chirp to Nyquist, hard-gated full-scale tone bursts (largest gain-curve swings), one silent
    channel, and sign-alternating full scale."""
    rng = np.random.RandomState(seed)
    n = nblocks * 1024
    out = np.zeros((n, 2), dtype=np.float64)
    seg = 4 * 1024
    t = np.arange(n, dtype=np.float64)
    for k in range(0, n, seg):
        sl = slice(k, min(n, k + seg))
        m = sl.stop - sl.start
        kind = (k // seg) % 8
        if kind == 0:
            out[sl] = rng.randint(-32768, 32768, size=(m, 2)) / 32768.0
        elif kind == 1:
            sq = np.where((t[sl] // 37) % 2 == 0, 32767, -32768) / 32768.0
            out[sl, 0] = sq
            out[sl, 1] = -sq
        elif kind == 2:
            imp = np.zeros(m)
            imp[::701] = 1.0 - 1.0 / 32768.0
            out[sl, 0] = imp
            out[sl, 1] = -imp[::-1]
        elif kind == 3:
            out[sl, 0] = 0.25 + rng.randint(-1, 2, size=m) / 32768.0
            out[sl, 1] = rng.randint(-1, 2, size=m) / 32768.0
        elif kind == 4:
            ph = np.pi * (np.arange(m) ** 2) / (2.0 * m)       # 0 .. fs/2
            out[sl, 0] = 0.9 * np.sin(ph)
            out[sl, 1] = 0.9 * np.cos(ph)
        elif kind == 5:
            gate = ((np.arange(m) // 300) % 3 == 0).astype(np.float64)
            out[sl, 0] = gate * np.sin(2 * np.pi * 5000.0 * t[sl] / 44100.0) * (32767 / 32768.0)
            out[sl, 1] = (1 - gate) * np.sin(2 * np.pi * 900.0 * t[sl] / 44100.0) * 0.7
        elif kind == 6:
            out[sl, 0] = rng.randint(-20000, 20000, size=m) / 32768.0
        else:
            alt = np.where(np.arange(m) % 2 == 0, 32767, -32768) / 32768.0
            out[sl, 0] = alt
            out[sl, 1] = alt
    s16 = np.clip(np.round(out * 32768.0), -32768, 32767)
    return (s16.astype(np.float32) / np.float32(32768.0)).reshape(nblocks, 1024, 2).astype(np.float32)


def pcm_hot(nblocks, seed=5, gain=40.0):
    """Input ABOVE full scale (what a float WAV may hold): the stress signal and a few loud sines, times `gain`, unquantised -
    spectra beyond MAX_SCALE = 1.0, so TScaler::Scale clamps scale factors and clips values (atrac_scale.cpp:150-167: the
    reference's "Scale error" / "clipping" diagnostics), in residual BFUs and in tonal components alike."""
    x = pcm_stress(nblocks, seed=seed).astype(np.float64).reshape(-1, 2)
    t = np.arange(x.shape[0], dtype=np.float64)
    for i, f in enumerate((700.0, 2500.0, 6100.0, 9000.0)):
        x[:, i & 1] += 0.4 * np.sin(2 * np.pi * f * t / 44100.0 + i)
    return (x * gain).astype(np.float32).reshape(nblocks, 1024, 2)


def capture_ref_diagnostics(fn):
    """Runs fn() with file descriptor 2 redirected and returns (fn's result, "Scale error" lines, "clipping" lines): the
    reference reports TScaler::Scale's overflow conditions on std::cerr only (atrac_scale.cpp:150-153, 163-167)."""
    import sys
    import tempfile
    sys.stderr.flush()
    saved = os.dup(2)
    with tempfile.TemporaryFile() as tmp:
        os.dup2(tmp.fileno(), 2)
        try:
            res = fn()
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        tmp.seek(0)
        text = tmp.read().decode(errors="replace")
    return res, text.count("Scale error: absSpec > MAX_SCALE"), text.count("clipping, scaled value")


def oracle_diag_counts(reset=False):
    """(scale errors, clipped values) the oracle has counted since the last reset (at3o_diag_counts; process-wide)."""
    o = oracle()
    out = (ctypes.c_ulonglong * 2)()
    o.lib.at3o_diag_counts(out, int(bool(reset)))
    return int(out[0]), int(out[1])


SIGNALS = {
    "noise": pcm_noise,
    "burst": pcm_burst,
    "tones": pcm_tones,
    "silence": pcm_silence,
    "mix": pcm_mix,
}


def flatness_threshold_walk(oracle, n_iter=48):
    """Stationary tones + g x white noise: as g grows, the BFUs around the tones stop being "tonal" one by one, each when
    its spectral flatness crosses 0.01 (atrac3denc.cpp:606, atrac_psy_common.cpp:158-199). Bisecting g on the number of
    tonal blocks of one frame converges onto such a crossing from both sides until two neighbouring f32 PCM inputs
    decide differently. Returns every PCM stream visited ([n, nb, 1024, 2]) and, for the last pair of each walk, the
    distance |flat - 0.01| of the deciding BFU."""
    nb, k = 6, 3
    n = nb * 1024
    t = np.arange(n, dtype=np.float64)
    tone = sum(0.1 * np.sin(2 * np.pi * f * t / 44100.0 + 0.3 * i) for i, f in enumerate((440.0, 1000.0, 3000.0, 7000.0, 11000.0)))
    noise = np.random.RandomState(11).randn(n)

    def pcm_of(g):
        return np.repeat((tone + g * noise)[:, None], 2, axis=1).astype(np.float32).reshape(nb, 1024, 2)

    def n_tonal(pcm):
        return int(oracle.encode(pcm, LP2, 1, 0, taps=True)[1]["n_tonal"][k, 0])

    def flat_of(pcm):
        sub = oracle.qmf(np.ascontiguousarray(pcm[:, :, 0]).reshape(-1) * np.float32(0.25))
        bands = np.zeros((4, 512), np.float32)
        for f in range(k + 1):      # frame k = the (k+1)-th MDCT of the stream (no gain control in this walk)
            bands[:, 256:] = sub[:, f * 256:(f + 1) * 256]
            specs, bands = oracle.mdct(bands)
        return oracle.flatness(specs * specs)

    visited, closest = [], []
    top = n_tonal(pcm_of(0.01))
    assert top >= 2 and n_tonal(pcm_of(0.1)) == 0
    for target in range(1, top + 1):
        lo, hi = 0.01, 0.1                       # n_tonal(lo) >= target > n_tonal(hi)
        p_lo, p_hi = pcm_of(lo), pcm_of(hi)
        for _ in range(n_iter):
            mid = 0.5 * (lo + hi)
            p_mid = pcm_of(mid)
            if np.array_equal(p_mid, p_lo) or np.array_equal(p_mid, p_hi):
                break
            visited.append(p_mid)
            if n_tonal(p_mid) >= target:
                lo, p_lo = mid, p_mid
            else:
                hi, p_hi = mid, p_mid
        f_lo, f_hi = flat_of(p_lo), flat_of(p_hi)
        flip = np.nonzero((f_lo < np.float32(0.01)) != (f_hi < np.float32(0.01)))[0]
        flip = flip[(flip >= 8) & (flip < 29)]
        assert flip.size >= 1
        closest.append(min(float(min(abs(np.float64(f_lo[b]) - 0.01), abs(np.float64(f_hi[b]) - 0.01))) for b in flip))
    return np.stack(visited), closest



# ----------------------------------------------------------------------------------------------
# ATRAC1: oracle/at1_oracle.c and the reference's TAtrac1Encoder (oracle/_ref)
# ----------------------------------------------------------------------------------------------
AT1_FRAME = 212
AT1_BLOCK = 512
# (window_auto, window_mask, bfu_idx_const) as TAtrac1EncodeSettings takes them
AT1_MODES = {"auto": (1, 0, 0), "long": (0, 0, 0), "short": (0, 7, 0), "mask5": (0, 5, 0), "auto_bfu8": (1, 0, 8),
             "auto_bfu3": (1, 0, 3), "mask2_bfu1": (0, 2, 1)}


def at1_blocks(pcm, nch=2):
    """[nblocks, 1024, 2] ATRAC3-shaped test PCM -> contiguous [2*nblocks, 512, nch]."""
    return np.ascontiguousarray(pcm.reshape(-1, AT1_BLOCK, 2)[:, :, :nch])


def at1_oracle_encode(pcm, mode="auto", taps=False):
    """pcm [n_blocks, 512, nch] float32 -> sound units [n_blocks, nch, 212] (+ specs, window masks, loudness)."""
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    lib = ctypes.CDLL(ORACLE_SO)
    nb, _, nch = pcm.shape
    auto, mask, bfu = AT1_MODES[mode] if isinstance(mode, str) else mode
    out = np.zeros((nb, nch, AT1_FRAME), np.uint8)
    specs = np.zeros((nb, nch, 512), np.float32)
    masks = np.zeros((nb, nch), np.int32)
    loud = np.zeros((nb,), np.float32)
    lib.at1o_encode.restype = ctypes.c_int
    n = lib.at1o_encode(_vp(pcm), nch, nb, auto, mask, bfu, _vp(out), _vp(specs), _vp(masks), _vp(loud))
    assert n == out.size
    return (out, specs, masks, loud) if taps else out


def at1_ref_encode(pcm, mode="auto"):
    lib = ctypes.CDLL(REF_SO)
    nb, _, nch = pcm.shape
    auto, mask, bfu = AT1_MODES[mode] if isinstance(mode, str) else mode
    out = np.zeros((nb, nch, AT1_FRAME), np.uint8)
    lib.at1ref_encode.restype = ctypes.c_int
    n = lib.at1ref_encode(_vp(pcm), nch, nb, auto, mask, bfu, _vp(out))
    assert n == out.size
    return out


# ----------------------------------------------------------------------------------------------
# ATRAC3plus front end: PQF analysis + TAt3pMDCT, one channel at a time
# ----------------------------------------------------------------------------------------------
def _at3p_lib(which):
    if which == "oracle" and not os.path.exists(ORACLE_SO):
        build_oracle()
    return ctypes.CDLL(ORACLE_SO if which == "oracle" else REF_SO), ("at3po_" if which == "oracle" else "at3pref_")


def at3p_pqf(x, which="oracle"):
    """x float32 [n_frames, 2048] (one channel) -> subbands [n_frames, 16, 128], start-of-stream state first."""
    lib, pre = _at3p_lib(which)
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros((x.shape[0], 16, 128), np.float32)
    getattr(lib, pre + "pqf_analyse")(_vp(x), x.shape[0], _vp(out))
    return out


def at3p_mdct(bands, flags=None, which="oracle"):
    """bands float32 [n_frames, 16, 128], flags uint16 [n_frames] (bit b = steep window in subband b) -> specs [n_frames, 2048]."""
    lib, pre = _at3p_lib(which)
    bands = np.ascontiguousarray(bands, np.float32)
    specs = np.zeros((bands.shape[0], 2048), np.float32)
    fl = None if flags is None else np.ascontiguousarray(flags, np.uint16)
    getattr(lib, pre + "mdct")(_vp(bands), None if fl is None else _vp(fl), bands.shape[0], _vp(specs))
    return specs


def at3p_ipqf_ref(bands):
    """The synthesis filter of the reference's unit test (decoder side), for round trips. Needs oracle/_ref."""
    bands = np.ascontiguousarray(bands, np.float32)
    out = np.zeros((bands.shape[0], 2048), np.float32)
    ctypes.CDLL(REF_SO).at3pref_ipqf(_vp(bands), bands.shape[0], _vp(out))
    return out


def at3p_signal(name, n_frames, channel=0, scale=1.0):
    """One channel of the ATRAC3 test signals cut into 2048-sample ATRAC3plus frames."""
    gens = dict(SIGNALS)
    gens["stress"] = pcm_stress
    x = gens[name](2 * n_frames + (32 if name == "stress" else 0)).reshape(-1, 2)[: n_frames * 2048, channel]
    return np.ascontiguousarray((x * np.float32(scale)).astype(np.float32).reshape(n_frames, 2048))


AT3P_INFO_DTYPE = np.dtype([("num_quant_units", "<i4"), ("bits_used", "<i4"), ("sfi", "u1", (2, 32)), ("tab", "u1", (2, 32)),
                            ("qu_bits", "<u2", (2, 32))])


def at3p_write_frames(specs, flags=None, which="oracle", info=False):
    """ScaleFrame + WriteFrame(channels, nullptr, sces): specs float32 [n_frames, channels, 2048], flags uint16
    [n_frames, channels] (bit b = steep window in subband b) -> frames uint8 [n_frames, 2048] (and the oracle's per-frame
    record when info=True)."""
    lib, pre = _at3p_lib(which)
    specs = np.ascontiguousarray(specs, np.float32)
    nf, nch, _ = specs.shape
    out = np.zeros((nf, 2048), np.uint8)
    fl = None if flags is None else np.ascontiguousarray(flags, np.uint16)
    fn = getattr(lib, pre + "write_frames")
    fn.restype = ctypes.c_int
    if which == "oracle":
        assert lib.at3po_frame_info_size() == AT3P_INFO_DTYPE.itemsize
        rec = np.zeros(nf, AT3P_INFO_DTYPE) if info else None
        rc = fn(_vp(specs), None if fl is None else _vp(fl), nch, nf, _vp(out), _vp(rec) if info else None)
    else:
        rec = None
        rc = fn(_vp(specs), None if fl is None else _vp(fl), nch, nf, _vp(out))
    assert rc == nf, rc
    return (out, rec) if info else out


def at3p_specs(name, n_frames, channels=2, scale=1.0):
    """The residual spectra TAt3PEnc::EncodeFrame would scale and pack for a test signal with the tonal analysis out of
    the way: PQF analysis, division by 32768 / 1.122018 (at3p.cpp:143-147), MDCT with sine windows. [n_frames, channels, 2048]"""
    out = np.zeros((n_frames, channels, 2048), np.float32)
    for ch in range(channels):
        bands = at3p_pqf(at3p_signal(name, n_frames, channel=ch, scale=scale))
        bands = (bands.astype(np.float64) / (32768.0 / 1.122018)).astype(np.float32)   # float / double, narrowed on the store
        out[:, ch] = at3p_mdct(bands)
    return out
