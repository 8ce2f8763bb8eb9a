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
    declared = set(re.findall(r"\b(at3hip_[a-z_]+)\s*\(", header))
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
