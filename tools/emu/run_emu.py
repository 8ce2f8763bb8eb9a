#!/usr/bin/env python3
"""DEVELOPMENT HARNESS: run the kernel sources through the CPU SIMT emulator (tools/emu) and diff
against the oracle. Not a test, not a benchmark - only a debugging aid for a GPU-less container."""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from at3_testlib import SIGNALS, LP2, LP4, oracle, oracle_diag_counts, pcm_dense_tonal, pcm_hot, pcm_stress
SIGNALS = dict(SIGNALS, stress=lambda nb: pcm_stress(nb, seed=7), hot=lambda nb: pcm_hot(nb),
               dense=lambda nb: pcm_dense_tonal(nb, 7))   # dense: every BFU tonal, runs across BFU boundaries (k_psy's parallel extraction)   # hot: above full scale (overflow counters)
from atracdenc_amd.binding import At3Hip

EMU = os.path.join(ROOT, "tools", "emu", "libat3hip_emu.so")

def build(strict=False):
    # --strict: -O0 (every cross-lane call keeps one address) + EMU_STRICT=1: the harness aborts when the lanes of a wavefront
    # meet in a rendezvous from two different calls, i.e. when a cross-lane read sits inside divergent control flow
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O0" if strict else "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-x", "c++",
                           "-I", os.path.join(ROOT, "tools", "emu"), "-include", os.path.join(ROOT, "tools", "emu", "at3_pk_emu.hpp"), "-o", EMU,
                           os.path.join(ROOT, "atracdenc_amd/csrc/at3hip.hip"),
                           os.path.join(ROOT, "atracdenc_amd/csrc/at1hip.hip"),
                           os.path.join(ROOT, "atracdenc_amd/csrc/at3phip.hip"),
                           os.path.join(ROOT, "atracdenc_amd/csrc/at3_tables.cpp"),
                           os.path.join(ROOT, "tools/emu/emu_runtime.cpp")])

if __name__ == "__main__":
    strict = "--strict" in sys.argv
    if strict: os.environ["EMU_STRICT"] = "1"
    if "--nobuild" not in sys.argv: build(strict)
    gain_form = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--gain-form=")), 0)   # AT3HIP_OPT_GAIN_FORM
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["noise", "burst", "tones", "silence", "mix", "stress"]
    nb = 6
    o = oracle()
    for name in names:
        for br in (LP2, LP4):
            for ng, nt in ((1, 1), (1, 0), (0, 0)):
                pcm = np.stack([SIGNALS[name](nb), SIGNALS["mix"](nb, seed=3)])
                t = time.time()
                enc = At3Hip(n_streams=2, max_blocks=nb, bitrate=br, no_gain=ng, no_tonal=nt, lib_path=EMU)
                if gain_form:
                    from atracdenc_amd.binding import OPT_GAIN_FORM
                    enc.set_option(OPT_GAIN_FORM, gain_form)
                if "--literal" in sys.argv:   # AT3HIP_OPT_LITERAL_FORMS: the guarded short forms (flatness, highFreqRatio) take their literal paths
                    from atracdenc_amd.binding import OPT_LITERAL_FORMS
                    enc.set_option(OPT_LITERAL_FORMS, 1)
                # feed in two pieces to exercise the carried state
                if "--s16" in sys.argv:   # at3hip_encode_s16: the same samples as 16-bit integers, converted on the "device"
                    p16 = np.round(np.clip(pcm, -1.0, 32767.0 / 32768.0) * 32768.0).astype(np.int16)
                    pcm = (p16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
                    got = np.concatenate([enc.encode_s16(p16[:, :4]), enc.encode(pcm[:, 4:5]), enc.encode_s16(p16[:, 5:])], axis=1)
                else:
                    got = np.concatenate([enc.encode(pcm[:, :4]), enc.encode(pcm[:, 4:])], axis=1)
                cnt = enc.counters()
                enc.close()
                oracle_diag_counts(reset=True)
                exp = np.stack([o.encode(pcm[i], br, ng, nt)[0] for i in range(2)])
                want = oracle_diag_counts(reset=True)
                # at3hip_get_counters against the oracle's count of TScaler::Scale's diagnostics (zero on everything but `hot`)
                print(f"{name:8s} overflow counters {cnt['scale_overflow']}, {cnt['clipped_values']} (oracle {want[0]}, {want[1]}): "
                      f"bad {int((cnt['scale_overflow'], cnt['clipped_values']) != want)}")
                # k_alloc_pack's own account of its rate loop (AT3_STAT): every lower bound that was later replaced by the bits must not exceed them
                st = (ctypes.c_ulonglong * 16).in_dll(ctypes.CDLL(EMU), "g_alloc_stats")
                cf = max(1, st[10])
                print(f"{name:8s} rate loop per channel-frame: {st[0] / cf:.1f} trips, {st[3] / cf:.1f} evaluations ({st[8] / cf:.1f} / {st[9] / cf:.1f} decided by the upper / "
                      f"lower bound), {st[5] / cf:.1f} units bounded, {st[7] / cf:.1f} quantised; {st[11]} bounds checked against the bits: bad {st[12]}")
                for k in range(16): st[k] = 0
                bad = (got != exp).any(axis=2)
                print(f"{name:8s} br={br} nogain={ng} notonal={nt}: shape {got.shape} mismatching frames "
                      f"{int(bad.sum())}/{bad.size} {np.argwhere(bad)[:6].tolist()} ({time.time()-t:.1f}s)")
