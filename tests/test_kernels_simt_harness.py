"""The kernel SOURCES of the product, compiled for the host and run lane by lane through the SIMT harness of tools/emu,
against the oracle - the parity gate that exists without a GPU (the `-m gpu` tests are the parity tests proper).

The harness is test infrastructure like the oracle: the product library never contains or loads it. It is built in its
strict form here (-O0 + EMU_STRICT): besides comparing results it aborts when the lanes of a wavefront reach a cross-lane
exchange (ballot, readlane, ds_bpermute, DPP, wave-level rendezvous) from two DIFFERENT calls, i.e. when such a read
sits inside divergent control flow - a class of mistake the GPU tolerates until the compiler or the data change.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs ROCm's clang++ to compile the kernel sources for the host")


def _run(script, *args, timeout=900):
    env = dict(os.environ, EMU_STRICT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "emu", script), *args], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    return out


def _assert_clean(out, min_cases):
    counts = re.findall(r"(?:mismatching frames|bad) (\d+)", out)
    assert len(counts) >= min_cases, out[-4000:]
    assert all(c == "0" for c in counts), out[-4000:]


@pytest.fixture(scope="module")
def harness():
    sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))
    import run_emu
    run_emu.build(strict=True)
    return run_emu.EMU


def test_atrac3_kernels(harness):
    """Six signals x LP2 / LP4 x (all tools, no gain, no gain + no tonal), two streams, two calls (carried state)."""
    _assert_clean(_run("run_emu.py", "--strict", "--nobuild"), 72)   # (frames and overflow counters per case)


def test_atrac3_overflow_counters(harness):
    """Input above full scale: frames and at3hip_get_counters (TScaler::Scale's "Scale error" / "clipping" diagnostics,
    atrac_scale.cpp:150-167, counted by k_psy) against the oracle, which tests/test_oracle_vs_ref.py pins to the lines the
    reference prints."""
    out = _run("run_emu.py", "--strict", "--nobuild", "hot")
    _assert_clean(out, 12)
    assert re.search(r"overflow counters [1-9]\d+, [1-9]\d+ ", out), out[-2000:]


def test_atrac3_dense_tonal_material(harness):
    """Every BFU tonal, runs continuing across BFU boundaries (at3_testlib.pcm_dense_tonal): k_psy's wavefront-parallel tonal
    extraction and mapping with both halves of its position list in use, under the emulator's rendezvous checks."""
    _assert_clean(_run("run_emu.py", "--strict", "--nobuild", "dense"), 12)


def test_gain_curve_select_walk(harness):
    """cell_divisors_packed - the gain curve's point list walked by selects, as k_gain_energy_scale, k_mdct_sub and k_gain_curve's score use it -
    against curve_divisor, the sample-by-sample restatement of TGainProcessor::Modulate (gain_processor.h:93-112), for 20 000 curves of
    ARBITRARY bytes per field (0 .. 7 points, levels 0 .. 15, locations 0 .. 31 in any order, repeated, adjacent): all 256 divisors, bit patterns."""
    import ctypes
    import numpy as np
    sys.path.insert(0, ROOT)
    from atracdenc_amd import binding as B
    lib = ctypes.CDLL(harness)
    enc = B.At3Hip(n_streams=1, max_blocks=2, lib_path=harness)
    rng = np.random.RandomState(5)
    n = 20000
    cv = np.zeros((n, 16), np.uint8)
    cv[:, 0] = rng.randint(0, 8, n)
    cv[:, 1:8] = rng.randint(0, 16, (n, 7))
    loc = rng.randint(0, 32, (n, 7))
    srt = rng.rand(n) < 0.6                                  # most as the encoder makes them: ascending
    loc[srt] = np.sort(loc[srt], axis=1)
    cv[:, 8:15] = loc
    cv[: n // 10, 1:8] = rng.randint(0, 256, (n // 10, 7))   # and bytes no encoder writes: the walk masks what it must
    a = np.zeros((n, 256), np.float32)
    b = np.zeros((n, 256), np.float32)
    fn = lib.at3hip_debug_cell_divisors
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    lv = cv[: n // 10, 1:8]
    lv[lv > 15] &= 15                                         # (levels are four bits in the bitstream and in both walks' tables)
    assert fn(enc.ctx, cv.ctypes.data, n, a.ctypes.data, b.ctypes.data) == 0
    enc.close()
    bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
    assert bad.size == 0, (bad[:5].tolist(), cv[bad[0, 0]].tolist())
    assert (a != 1.0).any(axis=1).sum() > n // 2


def test_atrac3_gain_analysis_one_wavefront_form(harness):
    """AT3HIP_OPT_GAIN_FORM = AT3HIP_GAIN_FORM_ONE_WAVE (k_gain_analysis1, incl. the restated v_permlane32/16_swap, which tools/ubench/permlane_check
    compares with the hardware): the signals with gain curves x LP2 / LP4 x three option sets."""
    _assert_clean(_run("run_emu.py", "--strict", "--nobuild", "--gain-form=1", "burst", "stress"), 24)


def test_atrac3_literal_forms(harness):
    """AT3HIP_OPT_LITERAL_FORMS: the flatness measure per line and k_gain_spec's energy sums as the reference's chains."""
    _assert_clean(_run("run_emu.py", "--strict", "--nobuild", "--literal", "mix", "stress"), 24)


def test_atrac3_s16_entry_point(harness):
    """at3hip_encode_s16 (k_s16_to_f32 + the unchanged pipeline), calls of both kinds alternating on one context."""
    _assert_clean(_run("run_emu.py", "--strict", "--nobuild", "--s16", "mix"), 12)


def test_atrac1_kernels(harness):
    _assert_clean(_run("run_emu_at1.py", "--nobuild"), 24 * 4)


def test_atrac3plus_front_kernels(harness):
    _assert_clean(_run("run_emu_at3p.py", "--nobuild"), 12)


def test_atrac3plus_frame_kernels(harness):
    _assert_clean(_run("run_emu_at3p_write.py", "--nobuild"), 26)
