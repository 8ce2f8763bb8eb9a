"""The C-ABI library loads and exports every symbol include/at3hip.h declares (no compute without a GPU)."""
import os
import re

import pytest


def test_library_exports_every_declared_symbol():
    import atracdenc_amd
    if not os.path.exists(atracdenc_amd.LIB_PATH):
        atracdenc_amd.build_library()
    lib = atracdenc_amd.load_library()
    header = open(os.path.join(os.path.dirname(atracdenc_amd.__file__), "..", "include", "at3hip.h")).read()
    declared = set(re.findall(r"\b(at3hip_[a-z0-9_]+)\s*\(", header))
    assert declared == set(atracdenc_amd.binding.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.at3hip_version() >> 16 == 1
    header1 = open(os.path.join(os.path.dirname(atracdenc_amd.__file__), "..", "include", "at1hip.h")).read()
    declared1 = set(re.findall(r"\b(at1hip_[a-z_]+)\s*\(", header1))
    assert declared1 == set(atracdenc_amd.binding.AT1_SYMBOLS)
    for name in declared1:
        assert hasattr(lib, name), name
    headerp = open(os.path.join(os.path.dirname(atracdenc_amd.__file__), "..", "include", "at3phip.h")).read()
    declaredp = set(re.findall(r"\b(at3phip_[a-z_]+)\s*\(", headerp))
    assert declaredp == set(atracdenc_amd.binding.AT3P_SYMBOLS)
    for name in declaredp:
        assert hasattr(lib, name), name


def test_library_exports_nothing_but_the_c_abi():
    """Built with -fvisibility=hidden and csrc/exports.map: `nm -D` shows the three C ABIs of include/ and nothing else
    (no kernel stubs, no table builders, no C++ symbols)."""
    import subprocess
    import atracdenc_amd
    if not os.path.exists(atracdenc_amd.LIB_PATH):
        atracdenc_amd.build_library()
    out = subprocess.check_output(["nm", "-D", "--defined-only", atracdenc_amd.LIB_PATH], text=True)
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    b = atracdenc_amd.binding
    assert names == set(b.SYMBOLS) | set(b.AT1_SYMBOLS) | set(b.AT3P_SYMBOLS), sorted(names ^ (set(b.SYMBOLS) | set(b.AT1_SYMBOLS) | set(b.AT3P_SYMBOLS)))


def test_no_cpu_fallback_when_library_missing(tmp_path):
    import atracdenc_amd
    with pytest.raises(atracdenc_amd.At3HipError):
        atracdenc_amd.load_library(str(tmp_path / "missing.so"))


def test_product_does_not_reference_the_oracle():
    root = os.path.join(os.path.dirname(__file__), "..", "atracdenc_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "at3o_" not in text and "at1o_" not in text and "at3po_" not in text and "libat3oracle" not in text and "at3_testlib" not in text, f


def test_vlc_length_constants():
    """at3_k_alloc.hpp holds the Huffman code LENGTHS as rows of 4-bit fields in 64-bit constants (indexed by |mantissa|,
    or by the pair index for selector 1). Re-derive them from the code table in at3_common.hpp (atrac3.h:96-176)."""
    root = os.path.join(os.path.dirname(__file__), "..", "atracdenc_amd", "csrc")
    common = open(os.path.join(root, "at3_common.hpp")).read()
    a = common.index("__device__ static const uint16_t c_huff[130] = {")
    bits = [int(x) for x in re.findall(r"HE\(0x[0-9A-Fa-f]+, (\d+)\)", common[a:common.index("};", a)])]
    assert len(bits) == 130
    off = {2: 9, 3: 14, 4: 0, 5: 21, 6: 36, 7: 67}
    maxabs = {2: 2, 3: 3, 4: 4, 5: 7, 6: 15, 7: 31}
    alloc = open(os.path.join(root, "at3_k_alloc.hpp")).read()
    # ONE definition of the rows: kVlcLo / kVlcHi7 / kVlcTop - vlc_row and unit_bounds' lb_row_of both read them, no second copy of a literal
    lo_txt = re.search(r"constexpr unsigned long long kVlcLo\[8\] = \{([^}]*)\}", alloc).group(1)
    k_lo = [int(x.strip().rstrip("ul"), 16) for x in lo_txt.split(",")]
    k_hi7 = int(re.search(r"constexpr unsigned long long kVlcHi7 = (0x[0-9a-fA-F]+)ull", alloc).group(1), 16)
    k_top = [int(x) for x in re.search(r"constexpr int kVlcTop\[8\] = \{([^}]*)\}", alloc).group(1).split(",")]
    assert len(k_lo) == 8 and len(k_top) == 8 and k_lo[0] == k_lo[1] == 0
    assert len(re.findall(r"0x4888888888877777", alloc)) == 1 and len(re.findall(r"0x7766666666555553", alloc)) == 1   # (no stray copies)
    for sel in range(2, 8):
        k = 0
        for m in range(maxabs[sel] + 1):
            if m == 0:
                ln = bits[off[sel]]
            else:
                ln = bits[off[sel] + (m << 1) - 1]
                assert ln == bits[off[sel] + (m << 1)]      # +m and -m: same length
            k |= ln << (4 * m)
        assert k_top[sel] == (1 if sel == 2 else maxabs[sel])   # (wordlen 2 rounds to |m| <= 1: its table's third length is never a neighbour)
        if sel < 7:
            assert k_lo[sel] == k, sel
        else:
            assert k_lo[7] == k & ((1 << 64) - 1) and k_hi7 == k >> 64
        assert f"case {sel}: k = kVlcLo[{sel}]" in alloc or sel == 7
        assert f"lb_row(kVlcLo[{sel}], kVlcTop[{sel}]" in alloc
        # the property unit_bounds' lower bound rests on: a code's length never shrinks with |m|, but for the top code of wordlens 5 .. 7
        lens = [(k >> (4 * m)) & 15 for m in range(maxabs[sel] + 1)]
        drops = [m for m in range(1, len(lens)) if lens[m] < lens[m - 1]]
        assert drops == ([maxabs[sel]] if sel >= 5 else []), (sel, lens)
    rt9 = [8, 4, 7, 2, 0, 1, 6, 3, 5]
    kp = sum(bits[rt9[i]] << (4 * i) for i in range(9))
    assert f"({hex(kp)}ull >> (4 * (3 * (m0 + 1) + (m1 + 1))))" in alloc
    ki = sum(rt9[i] << (4 * i) for i in range(9))
    assert f"({hex(ki)}ull >> (4 * (3 * (m0 + 1) + (m1 + 1))))" in alloc


def test_spread_integer_shortcut_is_exact():
    """k_alloc_pack (at3_k_alloc.hpp, TConfigure's spread) replaces the reference's two sequential float sums over the 32
    scale-factor indices (atrac3_bitstream.cpp:590-621) by integer reductions whenever S2 = sum (32 sfi - m)^2 < 2^24, m the
    indices' total: then every term and every partial sum of the float loops is exactly representable. Replay both here."""
    import numpy as np
    rng = np.random.RandomState(11)
    f32 = np.float32
    taken = 0
    for trial in range(4000):
        kind = trial % 4
        if kind == 0:
            sfi = rng.randint(0, 64, size=32)
        elif kind == 1:
            sfi = np.clip(rng.randint(20, 30) + rng.randint(-6, 7, size=32), 0, 63)
        elif kind == 2:
            sfi = np.where(rng.rand(32) < rng.rand(), 63, 0)
        else:
            sfi = np.clip((rng.randn(32) * rng.uniform(1, 25) + 30).astype(int), 0, 63)
        s = f32(0)
        for v in sfi:
            s = f32(s + f32(v))
        s = f32(s / f32(32))
        sigma = f32(0)
        for v in sfi:
            t = f32(f32(v) - s)
            t = f32(t * t)
            sigma = f32(sigma + t)
        m = int(sfi.sum())
        d = 32 * sfi.astype(np.int64) - m
        S2 = int((d * d).sum())
        if S2 < (1 << 24):
            taken += 1
            assert f32(f32(S2) / f32(1024.0)) == sigma, (sfi, S2)
    assert 1000 < taken < 4000    # both sides of the guard were visited
