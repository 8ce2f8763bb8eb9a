#!/usr/bin/env python3
"""DEVELOPMENT HARNESS (build container, no GPU): the fused QMF + MDCT kernel's SOURCE through the CPU SIMT emulator (tools/emu)
against the oracle's spectra, for a set of -D build flags.  usage: tools/k1/emu_check.py [-DK1_NW=8 ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from at3_testlib import SIGNALS, oracle
from atracdenc_amd.binding import At3Hip

flags = [a for a in sys.argv[1:] if a.startswith("-D")]
tag = "".join(c if c.isalnum() else "_" for c in "".join(flags))[:80]
EMU = os.path.join(ROOT, "build_ab", f"libemu_{tag}.so")
os.makedirs(os.path.dirname(EMU), exist_ok=True)
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-x", "c++", *flags,
                       "-I", os.path.join(ROOT, "tools", "emu"), "-include", os.path.join(ROOT, "tools", "emu", "at3_pk_emu.hpp"), "-o", EMU,
                       os.path.join(ROOT, "atracdenc_amd/csrc/at3hip.hip"), os.path.join(ROOT, "atracdenc_amd/csrc/at1hip.hip"),
                       os.path.join(ROOT, "atracdenc_amd/csrc/at3phip.hip"), os.path.join(ROOT, "atracdenc_amd/csrc/at3_tables.cpp"),
                       os.path.join(ROOT, "tools/emu/emu_runtime.cpp")])
o = oracle()

def oracle_spectra(pcm):
    nb = pcm.shape[0]
    out = np.zeros((nb - 1, 2, 1024), np.float32)
    for ch in range(2):
        sub = o.qmf(np.ascontiguousarray(pcm[:, :, ch]).reshape(-1) * np.float32(0.25))
        bands = np.zeros((4, 512), np.float32)
        for f in range(nb - 1):
            bands[:, 256:] = sub[:, f * 256:(f + 1) * 256]
            specs, bands = o.mdct(bands)
            out[f, ch] = specs
    return out

bad_total = 0
for nb, S, runs, chain in ((9, 3, 0, 0), (9, 3, 1, 0), (9, 3, 3, 1), (9, 3, 8, 1), (12, 5, 5, 0), (3, 2, 0, 0), (2, 1, 0, 0),
                            (9, 3, 4, 2), (9, 3, 8, 2), (17, 2, 4, 2), (17, 2, 8, 2), (17, 2, 16, 2), (12, 3, 0, 2), (6, 2, 4, 2), (5, 2, 4, 2)):
    names = ["noise", "mix", "tones", "burst", "stress"][:S] if S <= 3 else ["noise", "mix", "tones", "burst", "noise"]
    pcm = np.stack([SIGNALS[n](nb, seed=4 + i) if n == "noise" else SIGNALS[n](nb) for i, n in enumerate(names)]).astype(np.float32)
    enc = At3Hip(n_streams=S, max_blocks=nb, no_gain=True, lib_path=EMU)
    if runs: enc.set_option(1, runs)
    if chain: enc.set_option(6, chain)
    specs = np.full((S, nb - 1, 2, 1024), np.nan, np.float32)
    enc.qmf_mdct_device(pcm.ctypes.data, nb, specs.ctypes.data)
    enc.close()
    bad = 0
    for i in range(S):
        exp = oracle_spectra(pcm[i])
        bad += int((specs[i].view(np.uint32) != exp.view(np.uint32)).sum())
    bad_total += bad
    print(f"flags {' '.join(flags) or '(none)'}: {S} streams x {nb} blocks, runs {runs or 'auto'} chain {chain}: mismatching words {bad}")
sys.exit(1 if bad_total else 0)
