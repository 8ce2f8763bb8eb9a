"""The claim k_alloc_pack's rate loop rests on since round 5, checked on the oracle alone (no GPU): a comparison of TAlloc::Encode
(atrac3_bitstream.cpp:621-659) that is decided from the CLC bits (upper bound) or from unit_bounds' lower bound of the VLC bits comes
out as the reference's own comparison does, and no such bound exceeds the bits QuantMantisas' energy-adaptive pass ends with
(atrac_scale.cpp:40-130). tools/oracle_bounds_count.py replays the scheme next to the real evaluation inside a patched temporary copy of
oracle/at3_oracle.c and exits non-zero on the first disagreement; the SIMT harness makes the same check on the kernel's own bounds."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bounds_decide_like_the_bits():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "oracle_bounds_count.py"), "9"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [ln for ln in r.stdout.splitlines() if "BAD" in ln]
    assert len(rows) == 4, r.stdout
    for ln in rows:
        assert re.search(r"BAD 0 ", ln), ln
        ref_lines = float(re.search(r"reference: [\d.]+ units, \d+ lines, (\d+) through the pass", ln).group(1))
        new_lines = float(re.search(r"quantised \((\d+) lines through the pass\)", ln).group(1))
        assert new_lines < 0.7 * ref_lines, ln   # the point of it: most units never need the pass
