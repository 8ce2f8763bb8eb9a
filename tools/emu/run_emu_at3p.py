#!/usr/bin/env python3
"""DEVELOPMENT HARNESS: the ATRAC3plus front-end kernels through the CPU SIMT emulator against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))
import numpy as np
from at3_testlib import at3p_signal, at3p_pqf, at3p_mdct
from atracdenc_amd.binding import At3pHip
import run_emu

if __name__ == "__main__":
    if "--nobuild" not in sys.argv: run_emu.build()
    nf = 5
    rng = np.random.RandomState(2)
    for nch in (2, 1):
        chans = [[at3p_signal(n, nf, channel=c, scale=sc) for c in range(nch)] for n, sc in (("mix", 32768.0), ("stress", 1.0))]
        pcm = np.stack([np.stack(cs, axis=-1) for cs in chans])          # [S, F, 2048, C]
        flags = rng.randint(0, 65536, size=(2, nf, nch)).astype(np.uint16)
        for fl, rs in ((None, False), (flags, False), (flags, True)):
            t = time.time()
            enc = At3pHip(n_streams=2, max_frames=nf, channels=nch, lib_path=run_emu.EMU)
            parts = [enc.pqf_mdct(pcm[:, a:b], None if fl is None else fl[:, a:b], rs) for a, b in ((0, 2), (2, 5))]
            enc.close()
            bands = np.concatenate([p[0] for p in parts], axis=1); specs = np.concatenate([p[1] for p in parts], axis=1)
            bad_b = bad_s = 0
            for s in range(2):
                for c in range(nch):
                    eb = at3p_pqf(pcm[s, :, :, c])
                    x = eb if not rs else (eb.astype(np.float64) / (32768.0 / 1.122018)).astype(np.float32)
                    es = at3p_mdct(x, None if fl is None else fl[s, :, c])
                    bad_b += int((bands[s, :, c].view(np.uint32) != eb.view(np.uint32)).sum())
                    bad_s += int((specs[s, :, c].view(np.uint32) != es.view(np.uint32)).sum())
            print(f"nch={nch} flags={'none' if fl is None else 'rand'} residual={rs}: bands bad {bad_b} specs bad {bad_s} ({time.time()-t:.1f}s)")
