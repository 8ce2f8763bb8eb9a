"""atracdenc_amd/csrc/at3_libm64.hpp (glibc 2.35's f64 log / exp restated for the device: the literal form of
CalcSpectralFlatnessPerBfu, atrac_psy_common.cpp:184,194) against this machine's libm, bit for bit. Host-only: the header
compiles as plain C++; the device executes the same IEEE operations (f64 add / mul / fma, no contraction)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_check(tmp_path, n, skip_ok=True):
    exe = str(tmp_path / "test_libm64")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", os.path.join(ROOT, "tests", "host", "test_libm64.cpp"),
                           "-o", exe, "-lm"])
    r = subprocess.run([exe, str(n)], capture_output=True, text=True)
    if r.returncode == 77:
        if skip_ok:
            pytest.skip(r.stdout.strip())
        pytest.fail("this host's libm does not run the variants at3_libm64.hpp restates (no FMA): frames are then bit-identical to the "
                    "reference as an FMA host runs it, not to this host's - see README, 'Supported reference platform'. " + r.stdout.strip())
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches; exp:" in r.stdout and r.stdout.rstrip().endswith(" 0 mismatches"), r.stdout
    return r.stdout


def test_restated_log_exp_equal_libm(tmp_path):
    run_check(tmp_path, 2000000)


def test_generated_data_is_current():
    """at3_libm64.inc is what tools/gen_libm_f64.py reads out of this image's libm.a."""
    import importlib.util
    if not os.path.exists("/usr/lib/x86_64-linux-gnu/libm-2.35.a"):
        pytest.skip("no static libm-2.35 here")
    spec = importlib.util.spec_from_file_location("gen_libm_f64", os.path.join(ROOT, "tools", "gen_libm_f64.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    before = open(gen.OUT).read()
    gen.main()
    assert open(gen.OUT).read() == before


@pytest.mark.gpu
def test_restated_log_exp_equal_libm_on_the_gpu_box(tmp_path):
    """The same check on the host that builds the product's tables and runs the reference for cpu_baseline."""
    run_check(tmp_path, 500000, skip_ok=False)   # on the measurement box a mismatch of platforms must be loud, not a skip
