"""ctypes binding of include/at3hip.h (the same stub a maintainer would write for any FFI)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libat3hip.so")
CSRC = os.path.join(HERE, "csrc")

AT3HIP_PCM_ON_DEVICE = 1
AT3HIP_OUT_ON_DEVICE = 2
AT3HIP_ASYNC = 4
OPT_RUNS, OPT_LITERAL_FORMS, OPT_QUANT_TAP, OPT_GAIN_FORM, OPT_GAIN_WGS_PER_CU, OPT_CHAIN, OPT_TIMING_EVERY = 1, 2, 3, 4, 5, 6, 7
OPT_FLATNESS_LITERAL = OPT_LITERAL_FORMS           # (former name, same number)
GAIN_FORM_TWO_WAVES, GAIN_FORM_ONE_WAVE = 0, 1
AT3HIP_VERSION = (1 << 16) | 5                     # include/at3hip.h this stub mirrors: load_library refuses an older library
TAP_SPECTRA, TAP_CURVES, TAP_ENERGY_SCALE, TAP_PSY, TAP_LOUDNESS, TAP_QUANT, TAP_CLOCK, TAP_GAIN_ANALYSIS = 1, 2, 3, 4, 5, 6, 7, 8
LP2 = 132300
LP4 = 66150


class At3HipError(RuntimeError):
    pass


class Config(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("bitrate", "channels", "no_gain_control", "no_tonal", "bfu_idx_const",
                                               "n_streams", "max_blocks", "device_id")]


class Counters(ctypes.Structure):
    _fields_ = [("scale_overflow", ctypes.c_uint64), ("clipped_values", ctypes.c_uint64)]


class Timings(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("total_ms", "qmf_ms", "gain_ms", "curve_ms", "qmf_mdct_ms", "psy_ms",
                                               "alloc_ms")] + [("qmf_mdct_launches", ctypes.c_int32)]


SYMBOLS = ["at3hip_encode_s16", "at3hip_create", "at3hip_destroy", "at3hip_frame_size", "at3hip_joint_stereo", "at3hip_last_error",
           "at3hip_encode", "at3hip_reset", "at3hip_mdct", "at3hip_qmf_mdct", "at3hip_get_timings",
           "at3hip_set_stream", "at3hip_version", "at3hip_sync", "at3hip_get_timings_ago", "at3hip_read_tap",
           "at3hip_mdct_levels", "at3hip_gain_energy_scale", "at3hip_set_option", "at3hip_host_tables", "at3hip_host_alloc",
           "at3hip_host_free", "at3hip_wait_input", "at3hip_wait_frames", "at3hip_get_counters", "at3hip_device_numa_node"]
# include/at1hip.h
AT1_SYMBOLS = ["at1hip_create", "at1hip_destroy", "at1hip_last_error", "at1hip_encode", "at1hip_reset", "at1hip_get_timings",
               "at1hip_read_tap", "at1hip_host_tables", "at1hip_sync"]
# include/at3phip.h
AT3P_SYMBOLS = ["at3phip_create", "at3phip_destroy", "at3phip_last_error", "at3phip_reset", "at3phip_pqf_analyse", "at3phip_mdct",
                "at3phip_pqf_mdct", "at3phip_get_timings", "at3phip_host_tables", "at3phip_write_frames", "at3phip_encode_frames",
                "at3phip_get_write_timing", "at3phip_host_write_tables", "at3phip_sync"]


class At1Config(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("channels", "window_auto", "window_mask", "bfu_idx_const", "n_streams", "max_blocks",
                                              "device_id")]


class At3pConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("channels", "n_streams", "max_frames", "device_id")]


class At1Timings(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("total_ms", "front_ms", "scan_ms", "pack_ms")]


def build_library(verbose=False):
    """Compile libat3hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fvisibility=hidden", "-fPIC", "-shared",
           "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", LIB_PATH, os.path.join(CSRC, "at3hip.hip"), os.path.join(CSRC, "at1hip.hip"), os.path.join(CSRC, "at3phip.hip"),
           os.path.join(CSRC, "at3_tables.cpp")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


_lib_cache = {}


def load_library(path=None):
    path = path or os.environ.get("AT3HIP_LIB") or LIB_PATH
    if path in _lib_cache:
        return _lib_cache[path]
    # PyTorch wheels bundle their own HIP runtime; when torch shares the process (bench.py, tests) it has
    # to be loaded first so that libat3hip.so binds to the same runtime instance instead of a second copy.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(path):
        raise At3HipError(f"{path} not found: build it with atracdenc_amd.build_library() / __graft_entry__.build(); "
                          "there is no CPU fallback")
    lib = ctypes.CDLL(path)
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    lib.at3hip_version.restype = ctypes.c_uint32
    have = lib.at3hip_version()
    # same major number, and every entry point / option / wait depth this stub relies on (at3hip.h lists them per minor number)
    # (AT3HIP_MIN_MINOR=<n>: same-box A/B runs against a build of an earlier round that lacks only options the run does not set)
    need = (AT3HIP_VERSION & ~0xffff) | int(os.environ["AT3HIP_MIN_MINOR"]) if "AT3HIP_MIN_MINOR" in os.environ else AT3HIP_VERSION
    if have >> 16 != AT3HIP_VERSION >> 16 or have < need:
        raise At3HipError(f"{path} implements at3hip ABI {have >> 16}.{have & 0xffff}, this binding needs {AT3HIP_VERSION >> 16}.{AT3HIP_VERSION & 0xffff}: rebuild it")
    lib.at3hip_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    lib.at3hip_create.restype = ctypes.c_int
    lib.at3hip_destroy.argtypes = [vp]
    lib.at3hip_destroy.restype = None
    lib.at3hip_frame_size.argtypes = [vp]
    lib.at3hip_joint_stereo.argtypes = [vp]
    lib.at3hip_last_error.argtypes = [vp]
    lib.at3hip_last_error.restype = ctypes.c_char_p
    lib.at3hip_encode.argtypes = [vp, vp, i32, vp, ctypes.POINTER(i32), ctypes.c_uint32]
    lib.at3hip_encode_s16.argtypes = [vp, vp, i32, vp, ctypes.POINTER(i32), ctypes.c_uint32]
    lib.at3hip_reset.argtypes = [vp]
    lib.at3hip_mdct.argtypes = [vp, vp, vp, vp, vp, vp, i32, ctypes.c_uint32]
    lib.at3hip_mdct_levels.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, ctypes.c_uint32]
    lib.at3hip_gain_energy_scale.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, ctypes.c_uint32]
    lib.at3hip_qmf_mdct.argtypes = [vp, vp, i32, vp, ctypes.c_uint32]
    lib.at3hip_get_timings.argtypes = [vp, ctypes.POINTER(Timings)]
    lib.at3hip_set_stream.argtypes = [vp, vp]
    lib.at3hip_sync.argtypes = [vp]
    lib.at3hip_set_option.argtypes = [vp, i32, i32]
    lib.at3hip_host_tables.argtypes = [vp, ctypes.c_size_t]
    lib.at3hip_device_numa_node.argtypes = [i32]
    lib.at3hip_device_numa_node.restype = ctypes.c_int
    lib.at3hip_host_alloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.at3hip_host_free.argtypes = [vp, vp]
    lib.at3hip_wait_input.argtypes = [vp, i32]
    lib.at3hip_wait_frames.argtypes = [vp, i32]
    lib.at3hip_read_tap.argtypes = [vp, i32, vp, ctypes.c_size_t]
    lib.at3hip_get_timings_ago.argtypes = [vp, i32, ctypes.POINTER(Timings)]
    lib.at3hip_get_counters.argtypes = [vp, ctypes.POINTER(Counters), i32]
    lib.at3hip_version.restype = ctypes.c_uint32
    lib.at1hip_create.argtypes = [ctypes.POINTER(At1Config), ctypes.POINTER(vp)]
    lib.at1hip_destroy.argtypes = [vp]
    lib.at1hip_destroy.restype = None
    lib.at1hip_last_error.argtypes = [vp]
    lib.at1hip_last_error.restype = ctypes.c_char_p
    lib.at1hip_encode.argtypes = [vp, vp, i32, vp, ctypes.c_uint32]
    lib.at1hip_reset.argtypes = [vp]
    lib.at1hip_sync.argtypes = [vp]
    lib.at1hip_get_timings.argtypes = [vp, ctypes.POINTER(At1Timings)]
    lib.at1hip_read_tap.argtypes = [vp, i32, vp, ctypes.c_size_t]
    lib.at1hip_host_tables.argtypes = [vp, ctypes.c_size_t]
    lib.at3phip_create.argtypes = [ctypes.POINTER(At3pConfig), ctypes.POINTER(vp)]
    lib.at3phip_destroy.argtypes = [vp]
    lib.at3phip_destroy.restype = None
    lib.at3phip_last_error.argtypes = [vp]
    lib.at3phip_last_error.restype = ctypes.c_char_p
    lib.at3phip_reset.argtypes = [vp]
    lib.at3phip_pqf_analyse.argtypes = [vp, vp, i32, vp, ctypes.c_uint32]
    lib.at3phip_mdct.argtypes = [vp, vp, i32, vp, vp, ctypes.c_uint32]
    lib.at3phip_pqf_mdct.argtypes = [vp, vp, i32, vp, vp, vp, ctypes.c_uint32]
    lib.at3phip_get_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    lib.at3phip_host_tables.argtypes = [vp, ctypes.c_size_t]
    lib.at3phip_write_frames.argtypes = [vp, vp, i32, vp, vp, ctypes.c_uint32]
    lib.at3phip_encode_frames.argtypes = [vp, vp, i32, vp, ctypes.c_uint32]
    lib.at3phip_get_write_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    lib.at3phip_host_write_tables.argtypes = [vp, ctypes.c_size_t]
    lib.at3phip_sync.argtypes = [vp]
    _lib_cache[path] = lib
    return lib


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class At3Hip:
    """n_streams TAtrac3Encoder objects encoded side by side on one GPU."""

    def __init__(self, n_streams=1, max_blocks=64, bitrate=LP2, no_gain=False, no_tonal=False, bfu_idx_const=0,
                 device_id=0, lib_path=None, channels=2):
        self.lib = load_library(lib_path)
        self.channels = int(channels)
        self.cfg = Config(int(bitrate), int(channels), int(no_gain), int(no_tonal), int(bfu_idx_const), int(n_streams),
                          int(max_blocks), int(device_id))
        self.ctx = ctypes.c_void_p()
        rc = self.lib.at3hip_create(ctypes.byref(self.cfg), ctypes.byref(self.ctx))
        if rc != 0:
            self.ctx = None
            raise At3HipError(f"at3hip_create failed with {rc} (no usable MI355X / HIP runtime?)")
        self.n_streams = n_streams
        self.frame_size = self.lib.at3hip_frame_size(self.ctx)
        self.joint_stereo = bool(self.lib.at3hip_joint_stereo(self.ctx))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.at3hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise At3HipError(f"{what} failed ({rc}): {self.lib.at3hip_last_error(self.ctx).decode()}")

    def reset(self):
        self._check(self.lib.at3hip_reset(self.ctx), "at3hip_reset")

    def encode(self, pcm):
        """pcm float32 [n_streams, n_blocks, 1024, channels] (host) -> uint8 [n_streams, n_frames, frame_size]."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        assert pcm.ndim == 4 and pcm.shape[0] == self.n_streams and pcm.shape[2:] == (1024, self.channels), pcm.shape
        nb = pcm.shape[1]
        out = np.zeros((self.n_streams, nb, self.frame_size), dtype=np.uint8)
        nf = ctypes.c_int32()
        self._check(self.lib.at3hip_encode(self.ctx, _vp(pcm), nb, _vp(out), ctypes.byref(nf), 0), "at3hip_encode")
        n = nf.value
        return np.ascontiguousarray(out.reshape(-1)[: self.n_streams * n * self.frame_size].reshape(
            self.n_streams, n, self.frame_size))

    def encode_s16(self, pcm):
        """pcm int16 [n_streams, n_blocks, 1024, channels] (host) -> uint8 [n_streams, n_frames, frame_size] (at3hip_encode_s16)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        assert pcm.ndim == 4 and pcm.shape[0] == self.n_streams and pcm.shape[2:] == (1024, self.channels), pcm.shape
        nb = pcm.shape[1]
        out = np.zeros((self.n_streams, nb, self.frame_size), dtype=np.uint8)
        nf = ctypes.c_int32()
        self._check(self.lib.at3hip_encode_s16(self.ctx, _vp(pcm), nb, _vp(out), ctypes.byref(nf), 0), "at3hip_encode_s16")
        n = nf.value
        return np.ascontiguousarray(out.reshape(-1)[: self.n_streams * n * self.frame_size].reshape(self.n_streams, n, self.frame_size))

    def encode_device_s16(self, pcm_ptr, n_blocks, out_ptr, asynchronous=False):
        """Device-resident int16 PCM / out (raw pointers). Returns frames per stream."""
        nf = ctypes.c_int32()
        flags = AT3HIP_PCM_ON_DEVICE | AT3HIP_OUT_ON_DEVICE | (AT3HIP_ASYNC if asynchronous else 0)
        self._check(self.lib.at3hip_encode_s16(self.ctx, ctypes.c_void_p(pcm_ptr), n_blocks, ctypes.c_void_p(out_ptr),
                                               ctypes.byref(nf), flags), "at3hip_encode_s16")
        return nf.value

    def encode_device(self, pcm_ptr, n_blocks, out_ptr, asynchronous=False):
        """Device-resident PCM/out (raw pointers, e.g. torch tensor .data_ptr()). Returns frames per stream.
        asynchronous=True only queues the work (AT3HIP_ASYNC): call sync() before the frames are read."""
        nf = ctypes.c_int32()
        flags = AT3HIP_PCM_ON_DEVICE | AT3HIP_OUT_ON_DEVICE | (AT3HIP_ASYNC if asynchronous else 0)
        self._check(self.lib.at3hip_encode(self.ctx, ctypes.c_void_p(pcm_ptr), n_blocks, ctypes.c_void_p(out_ptr),
                                           ctypes.byref(nf), flags), "at3hip_encode")
        return nf.value

    PSY_DTYPE = np.dtype([("loud_ch", "<f4"), ("n_tonal", "<i4"), ("sfi", "u1", 32), ("energy", "<f4", 32),
                          ("tonal", [("pos", "<u2"), ("bfu", "u1"), ("len", "u1"), ("sfi", "u1"), ("pad", "u1", 3),
                                     ("values", "<f4", 7), ("pad2", "u1", 4)], 24), ("flat", "<f4", 32)])
    QUANT_DTYPE = np.dtype([("err", "<f4", (7, 32)), ("cost", "<u4", (7, 32))])

    def read_tap(self, kind, dtype, shape):
        """Stage tap of the most recent encode call (AT3HIP_TAP_*), as a numpy array of `dtype` and `shape`."""
        out = np.zeros(shape, dtype=dtype)
        self._check(self.lib.at3hip_read_tap(self.ctx, int(kind), _vp(out), out.nbytes), "at3hip_read_tap")
        return out

    def sclk_mhz(self):
        """Shader clock observed under the last call's rate loop (AT3HIP_TAP_CLOCK: s_memtime cycles against the 100 MHz
        s_memrealtime over the life of the allocation kernel's workgroup 0), or None before the first frames."""
        c = self.read_tap(TAP_CLOCK, np.uint64, (2,))
        return float(c[0]) / float(c[1]) * 100.0 if c[1] else None

    def sync(self):
        self._check(self.lib.at3hip_sync(self.ctx), "at3hip_sync")

    def counters(self, reset=False):
        """at3hip_get_counters: what TScaler::Scale would have printed since create / reset - {"scale_overflow", "clipped_values"}."""
        c = Counters()
        self._check(self.lib.at3hip_get_counters(self.ctx, ctypes.byref(c), int(bool(reset))), "at3hip_get_counters")
        return {"scale_overflow": int(c.scale_overflow), "clipped_values": int(c.clipped_values)}

    def host_alloc(self, shape, dtype):
        """Page-locked host array (at3hip_host_alloc); free it with host_free(array) before close()."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        self._check(self.lib.at3hip_host_alloc(self.ctx, n, ctypes.byref(p)), "at3hip_host_alloc")
        buf = (ctypes.c_char * n).from_address(p.value)
        a = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p
        return a

    def host_free(self, a):
        p = self._pinned.pop(a.ctypes.data)
        self._check(self.lib.at3hip_host_free(self.ctx, p), "at3hip_host_free")

    def encode_host_async(self, pcm, out):
        """Queues one call on host arrays (pinned ones overlap copies and kernels); returns frames per stream. `out` is valid
        after wait_frames(ago) / sync(), `pcm` may be refilled after wait_input(ago)."""
        nf = ctypes.c_int32()
        fn = self.lib.at3hip_encode_s16 if pcm.dtype == np.int16 else self.lib.at3hip_encode
        self._check(fn(self.ctx, _vp(pcm), pcm.shape[1], _vp(out), ctypes.byref(nf), AT3HIP_ASYNC), "at3hip_encode")
        return nf.value

    def wait_input(self, ago=0):
        self._check(self.lib.at3hip_wait_input(self.ctx, int(ago)), "at3hip_wait_input")

    def wait_frames(self, ago=0):
        self._check(self.lib.at3hip_wait_frames(self.ctx, int(ago)), "at3hip_wait_frames")

    def set_option(self, option, value):
        """AT3HIP_OPT_*: work partitioning / equivalent-form switches; results never change."""
        self._check(self.lib.at3hip_set_option(self.ctx, int(option), int(value)), "at3hip_set_option")

    def set_stream(self, hip_stream):
        """Queue the front half on the caller's HIP stream (a hipStream_t handle, e.g. torch.cuda.Stream().cuda_stream);
        0 / None goes back to the context's own stream."""
        self._check(self.lib.at3hip_set_stream(self.ctx, ctypes.c_void_p(int(hip_stream) if hip_stream else None)), "at3hip_set_stream")

    def timings_ago(self, ago):
        t = Timings()
        self._check(self.lib.at3hip_get_timings_ago(self.ctx, int(ago), ctypes.byref(t)), "at3hip_get_timings_ago")
        return {n: getattr(t, n) for n, _ in Timings._fields_}

    def qmf_mdct_device(self, pcm_ptr, n_blocks, specs_ptr):
        self._check(self.lib.at3hip_qmf_mdct(self.ctx, ctypes.c_void_p(pcm_ptr), n_blocks, ctypes.c_void_p(specs_ptr),
                                             AT3HIP_PCM_ON_DEVICE | AT3HIP_OUT_ON_DEVICE), "at3hip_qmf_mdct")

    def mdct(self, bands, n_points=None, level=None, loc=None, max_levels=False):
        """Batched TAtrac3MDCT::Mdct. bands float32 [n,4,512] -> (specs [n,1024], mutated bands[, max levels [n,4]])."""
        bands = np.ascontiguousarray(bands, dtype=np.float32).copy()
        n = bands.shape[0]
        specs = np.zeros((n, 1024), dtype=np.float32)
        curves = (None, None, None)
        if n_points is not None:
            n_points = np.ascontiguousarray(n_points, dtype=np.int32)
            level = np.ascontiguousarray(level, dtype=np.int32)
            loc = np.ascontiguousarray(loc, dtype=np.int32)
            curves = (_vp(n_points), _vp(level), _vp(loc))
        if max_levels:
            mx = np.zeros((n, 4), dtype=np.float32)
            self._check(self.lib.at3hip_mdct_levels(self.ctx, _vp(bands), _vp(specs), _vp(mx), *curves, n, 0), "at3hip_mdct_levels")
            return specs, bands, mx
        self._check(self.lib.at3hip_mdct(self.ctx, _vp(bands), _vp(specs), *curves, n, 0), "at3hip_mdct")
        return specs, bands

    def gain_energy_scale(self, prev_overlap, cur_input, prev_scale, n_points=None, level=None, loc=None):
        """Batched TAtrac3MDCT::CalcGainEnergyScale. prev_overlap / cur_input float32 [n,256], prev_scale [n], optional
        n_points [n], level / loc [n,8] -> float32 [n,4] (PrevHalf, CurHalf, Frame, NextOverlapScale)."""
        prev_overlap = np.ascontiguousarray(prev_overlap, dtype=np.float32)
        cur_input = np.ascontiguousarray(cur_input, dtype=np.float32)
        prev_scale = np.ascontiguousarray(prev_scale, dtype=np.float32)
        n = prev_overlap.shape[0]
        out = np.zeros((n, 4), dtype=np.float32)
        curves = (None, None, None)
        if n_points is not None:
            n_points = np.ascontiguousarray(n_points, dtype=np.int32)
            level = np.ascontiguousarray(level, dtype=np.int32)
            loc = np.ascontiguousarray(loc, dtype=np.int32)
            curves = (_vp(n_points), _vp(level), _vp(loc))
        self._check(self.lib.at3hip_gain_energy_scale(self.ctx, _vp(prev_overlap), _vp(cur_input), *curves, _vp(prev_scale),
                                                      _vp(out), n, 0), "at3hip_gain_energy_scale")
        return out

    def timings(self):
        t = Timings()
        self._check(self.lib.at3hip_get_timings(self.ctx, ctypes.byref(t)), "at3hip_get_timings")
        return {n: getattr(t, n) for n, _ in Timings._fields_}


# atracdenc_amd/csrc/at3_tables.hpp, struct Tables (cpx = two float32)
AT3_TABLES_DTYPE = np.dtype([("qmf_win", "<f4", 48), ("scale", "<f4", 64), ("enc_win", "<f4", 256), ("gain_level", "<f4", 16),
                             ("gain_interp", "<f4", 32), ("mdct_sincos", "<f4", 256), ("planck", "<f4", 512), ("hpf_w", "<f4", 4),
                             ("loud_curve", "<f4", 1024), ("ath_bfu", "<f4", 32), ("tw128", "<f4", (128, 2)), ("tw256", "<f4", (256, 2)),
                             ("stw256", "<f4", (128, 2)), ("tw2048", "<f4", (2048, 2)), ("stw2048", "<f4", (1024, 2)),
                             ("log2f_tab", "<f8", (16, 2)), ("log2f_poly", "<f8", 4), ("gain_tw", "<f4", (27, 128, 2)),
                             ("ga1_twb", "<f4", (2, 15, 16, 2)), ("ga1_twc", "<f4", (8, 3, 64, 2)),
                             ("mdct_tab", "<f4", (18, 16, 4)), ("spec16_win", "<f4", (16, 16, 2)), ("spec16_tw", "<f4", (15, 16, 2)), ("spec16_stw", "<f4", (9, 16, 2)), ("log_c", "<f8", 18),
                             ("log_tab", "<f8", (128, 2)), ("exp_c", "<f8", 8), ("exp_tab", "<u8", (128, 2))])


def at3_host_tables(lib_path=None):
    """The ATRAC3 constant tables as the library builds them on this host (no GPU involved)."""
    out = np.zeros((), dtype=AT3_TABLES_DTYPE)
    rc = load_library(lib_path).at3hip_host_tables(_vp(out), out.nbytes)
    if rc != 0:
        raise At3HipError(f"at3hip_host_tables failed ({rc}): table block is {out.nbytes} bytes here")
    return out


AT1_TABLES_DTYPE = np.dtype([("qmf_win", "<f4", 48), ("scale", "<f4", 64), ("sine", "<f4", 32), ("sc512", "<f4", 256),
                             ("sc256", "<f4", 128), ("sc64", "<f4", 32), ("tw128", "<f4", 256), ("tw64", "<f4", 128),
                             ("tw16", "<f4", 32), ("loud", "<f4", 512), ("ath_bfu", "<f4", 52), ("fir", "<f4", 10),
                             ("fix_long", "<f4", 52), ("fix_short", "<f4", 52), ("logf", "<f8", 36)])
assert AT1_TABLES_DTYPE.itemsize == 6904


def at1_host_tables(lib_path=None):
    """The ATRAC1 constant tables as the library builds them on the host (no GPU involved)."""
    out = np.zeros((), dtype=AT1_TABLES_DTYPE)
    rc = load_library(lib_path).at1hip_host_tables(_vp(out), out.nbytes)
    if rc != 0:
        raise At3HipError(f"at1hip_host_tables failed ({rc})")
    return out


class At1Hip:
    """n_streams TAtrac1Encoder objects encoded side by side on one GPU (include/at1hip.h)."""

    FRAME = 212
    TAP_SPECTRA, TAP_MASKS, TAP_LOUDNESS, TAP_TABLES = 1, 2, 3, 4

    def __init__(self, n_streams=1, max_blocks=64, channels=2, window_auto=True, window_mask=0, bfu_idx_const=0, device_id=0,
                 lib_path=None):
        self.lib = load_library(lib_path)
        self.channels, self.n_streams = int(channels), int(n_streams)
        self.cfg = At1Config(int(channels), int(bool(window_auto)), int(window_mask), int(bfu_idx_const), int(n_streams),
                             int(max_blocks), int(device_id))
        self.ctx = ctypes.c_void_p()
        rc = self.lib.at1hip_create(ctypes.byref(self.cfg), ctypes.byref(self.ctx))
        if rc != 0:
            self.ctx = None
            raise At3HipError(f"at1hip_create failed with {rc} (no usable MI355X / HIP runtime?)")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.at1hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise At3HipError(f"{what} failed ({rc}): {self.lib.at1hip_last_error(self.ctx).decode()}")

    def reset(self):
        self._check(self.lib.at1hip_reset(self.ctx), "at1hip_reset")

    def encode(self, pcm):
        """pcm float32 [n_streams, n_blocks, 512, channels] (host) -> uint8 [n_streams, n_blocks, channels, 212]."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        assert pcm.ndim == 4 and pcm.shape[0] == self.n_streams and pcm.shape[2:] == (512, self.channels), pcm.shape
        nb = pcm.shape[1]
        out = np.zeros((self.n_streams, nb, self.channels, self.FRAME), dtype=np.uint8)
        self._check(self.lib.at1hip_encode(self.ctx, _vp(pcm), nb, _vp(out), 0), "at1hip_encode")
        return out

    def encode_device(self, pcm_ptr, n_blocks, out_ptr, asynchronous=False):
        self._check(self.lib.at1hip_encode(self.ctx, ctypes.c_void_p(pcm_ptr), n_blocks, ctypes.c_void_p(out_ptr),
                                           AT3HIP_PCM_ON_DEVICE | AT3HIP_OUT_ON_DEVICE | (AT3HIP_ASYNC if asynchronous else 0)), "at1hip_encode")

    def sync(self):
        self._check(self.lib.at1hip_sync(self.ctx), "at1hip_sync")

    def read_tap(self, kind, dtype, shape):
        out = np.zeros(shape, dtype=dtype)
        self._check(self.lib.at1hip_read_tap(self.ctx, int(kind), _vp(out), out.nbytes), "at1hip_read_tap")
        return out

    def timings(self):
        t = At1Timings()
        self._check(self.lib.at1hip_get_timings(self.ctx, ctypes.byref(t)), "at1hip_get_timings")
        return {n: getattr(t, n) for n, _ in At1Timings._fields_}


AT3PHIP_RESIDUAL_SCALE = 16
AT3P_TABLES_DTYPE = np.dtype([("fir", "<f4", 384), ("sc32", "<f4", 16), ("sc256", "<f4", 128), ("tw8", "<f4", 16), ("tw64", "<f4", 128),
                              ("sine128", "<f4", 128), ("sine64", "<f4", 64)])
assert AT3P_TABLES_DTYPE.itemsize == 3456


def at3p_host_tables(lib_path=None):
    out = np.zeros((), dtype=AT3P_TABLES_DTYPE)
    rc = load_library(lib_path).at3phip_host_tables(_vp(out), out.nbytes)
    if rc != 0:
        raise At3HipError(f"at3phip_host_tables failed ({rc})")
    return out


class At3pHip:
    """ATRAC3plus front end (include/at3phip.h): PQF analysis and windowed MDCT-256 x 16 for n_streams streams."""

    def __init__(self, n_streams=1, max_frames=32, channels=2, device_id=0, lib_path=None):
        self.lib = load_library(lib_path)
        self.channels, self.n_streams = int(channels), int(n_streams)
        self.cfg = At3pConfig(int(channels), int(n_streams), int(max_frames), int(device_id))
        self.ctx = ctypes.c_void_p()
        rc = self.lib.at3phip_create(ctypes.byref(self.cfg), ctypes.byref(self.ctx))
        if rc != 0:
            self.ctx = None
            raise At3HipError(f"at3phip_create failed with {rc} (no usable MI355X / HIP runtime?)")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.at3phip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise At3HipError(f"{what} failed ({rc}): {self.lib.at3phip_last_error(self.ctx).decode()}")

    def reset(self):
        self._check(self.lib.at3phip_reset(self.ctx), "at3phip_reset")

    def _flags(self, win_flags, nf):
        if win_flags is None:
            return None, None
        fl = np.ascontiguousarray(win_flags, dtype=np.uint16)
        assert fl.shape == (self.n_streams, nf, self.channels), fl.shape
        return fl, _vp(fl)

    def pqf(self, pcm):
        """pcm float32 [S, F, 2048, C] -> subbands [S, F, C, 16, 128]."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        assert pcm.ndim == 4 and pcm.shape[0] == self.n_streams and pcm.shape[2:] == (2048, self.channels), pcm.shape
        out = np.zeros((self.n_streams, pcm.shape[1], self.channels, 16, 128), np.float32)
        self._check(self.lib.at3phip_pqf_analyse(self.ctx, _vp(pcm), pcm.shape[1], _vp(out), 0), "at3phip_pqf_analyse")
        return out

    def mdct(self, bands, win_flags=None, residual_scale=False):
        """bands [S, F, C, 16, 128], win_flags uint16 [S, F, C] or None -> specs [S, F, C, 2048]."""
        bands = np.ascontiguousarray(bands, dtype=np.float32)
        nf = bands.shape[1]
        fl, flp = self._flags(win_flags, nf)
        out = np.zeros((self.n_streams, nf, self.channels, 2048), np.float32)
        self._check(self.lib.at3phip_mdct(self.ctx, _vp(bands), nf, flp, _vp(out), AT3PHIP_RESIDUAL_SCALE if residual_scale else 0),
                    "at3phip_mdct")
        return out

    def pqf_mdct(self, pcm, win_flags=None, residual_scale=False, want_bands=True):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        nf = pcm.shape[1]
        fl, flp = self._flags(win_flags, nf)
        bands = np.zeros((self.n_streams, nf, self.channels, 16, 128), np.float32) if want_bands else None
        specs = np.zeros((self.n_streams, nf, self.channels, 2048), np.float32)
        self._check(self.lib.at3phip_pqf_mdct(self.ctx, _vp(pcm), nf, flp, _vp(bands) if want_bands else None, _vp(specs),
                                              AT3PHIP_RESIDUAL_SCALE if residual_scale else 0), "at3phip_pqf_mdct")
        return bands, specs

    def pqf_mdct_device(self, pcm_ptr, n_frames, specs_ptr):
        self._check(self.lib.at3phip_pqf_mdct(self.ctx, ctypes.c_void_p(pcm_ptr), n_frames, None, None, ctypes.c_void_p(specs_ptr),
                                              AT3HIP_PCM_ON_DEVICE | AT3HIP_OUT_ON_DEVICE), "at3phip_pqf_mdct")

    def write_frames(self, specs, win_flags=None):
        """ScaleFrame + WriteFrame without tonal block: specs [S, F, C, 2048], win_flags uint16 [S, F, C] or None
        -> frames uint8 [S, F, 2048]."""
        specs = np.ascontiguousarray(specs, dtype=np.float32)
        assert specs.ndim == 4 and specs.shape[0] == self.n_streams and specs.shape[2:] == (self.channels, 2048), specs.shape
        nf = specs.shape[1]
        fl, flp = self._flags(win_flags, nf)
        out = np.zeros((self.n_streams, nf, 2048), np.uint8)
        self._check(self.lib.at3phip_write_frames(self.ctx, _vp(specs), nf, flp, _vp(out), 0), "at3phip_write_frames")
        return out

    def encode_frames(self, pcm):
        """pcm float32 [S, F, 2048, C] -> frames uint8 [S, F, 2048] (tonal analysis finding nothing; no look-ahead delay)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        assert pcm.ndim == 4 and pcm.shape[0] == self.n_streams and pcm.shape[2:] == (2048, self.channels), pcm.shape
        nf = pcm.shape[1]
        out = np.zeros((self.n_streams, nf, 2048), np.uint8)
        self._check(self.lib.at3phip_encode_frames(self.ctx, _vp(pcm), nf, _vp(out), 0), "at3phip_encode_frames")
        return out

    def encode_frames_device(self, pcm_ptr, n_frames, frames_ptr, asynchronous=False):
        """asynchronous=True only queues the call (AT3HIP_ASYNC): sync() before the frames are read."""
        self._check(self.lib.at3phip_encode_frames(self.ctx, ctypes.c_void_p(pcm_ptr), n_frames, ctypes.c_void_p(frames_ptr),
                                                   AT3HIP_PCM_ON_DEVICE | AT3HIP_OUT_ON_DEVICE | (AT3HIP_ASYNC if asynchronous else 0)),
                    "at3phip_encode_frames")

    def sync(self):
        self._check(self.lib.at3phip_sync(self.ctx), "at3phip_sync")

    def timings(self):
        a, b, w = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        self._check(self.lib.at3phip_get_timings(self.ctx, ctypes.byref(a), ctypes.byref(b)), "at3phip_get_timings")
        self._check(self.lib.at3phip_get_write_timing(self.ctx, ctypes.byref(w)), "at3phip_get_write_timing")
        return {"pqf_ms": a.value, "mdct_ms": b.value, "write_ms": w.value}
