#!/usr/bin/env python3
"""DEVELOPMENT HARNESS: run the kernel sources through the CPU SIMT emulator (tools/emu) and diff
against the oracle. Not a test, not a benchmark - only a debugging aid for a GPU-less container."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from at3_testlib import SIGNALS, LP2, LP4, oracle
from atracdenc_amd.binding import At3Hip

EMU = os.path.join(ROOT, "tools", "emu", "libat3hip_emu.so")

def build():
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-x", "c++",
                           "-I", os.path.join(ROOT, "tools", "emu"), "-include", os.path.join(ROOT, "tools", "emu", "at3_pk_emu.hpp"), "-o", EMU,
                           os.path.join(ROOT, "atracdenc_amd/csrc/at3hip.hip"),
                           os.path.join(ROOT, "atracdenc_amd/csrc/at1hip.hip"),
                           os.path.join(ROOT, "atracdenc_amd/csrc/at3phip.hip"),
                           os.path.join(ROOT, "atracdenc_amd/csrc/at3_tables.cpp"),
                           os.path.join(ROOT, "tools/emu/emu_runtime.cpp")])

if __name__ == "__main__":
    if "--nobuild" not in sys.argv: build()
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["noise", "burst", "tones", "silence", "mix"]
    nb = 6
    o = oracle()
    for name in names:
        for br in (LP2, LP4):
            for ng, nt in ((1, 1), (1, 0), (0, 0)):
                pcm = np.stack([SIGNALS[name](nb), SIGNALS["mix"](nb, seed=3)])
                t = time.time()
                enc = At3Hip(n_streams=2, max_blocks=nb, bitrate=br, no_gain=ng, no_tonal=nt, lib_path=EMU)
                # feed in two pieces to exercise the carried state
                got = np.concatenate([enc.encode(pcm[:, :4]), enc.encode(pcm[:, 4:])], axis=1)
                enc.close()
                exp = np.stack([o.encode(pcm[i], br, ng, nt)[0] for i in range(2)])
                bad = (got != exp).any(axis=2)
                print(f"{name:8s} br={br} nogain={ng} notonal={nt}: shape {got.shape} mismatching frames "
                      f"{int(bad.sum())}/{bad.size} {np.argwhere(bad)[:6].tolist()} ({time.time()-t:.1f}s)")
