#!/usr/bin/env python3
"""DEVELOPMENT HARNESS: the ATRAC1 kernels through the CPU SIMT emulator (tools/emu) against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))
import numpy as np
from at3_testlib import SIGNALS, AT1_MODES, at1_blocks, at1_oracle_encode, pcm_stress
from atracdenc_amd.binding import At1Hip
import run_emu

if __name__ == "__main__":
    if "--nobuild" not in sys.argv: run_emu.build()
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["mix", "burst", "stress"]
    gens = dict(SIGNALS); gens["stress"] = pcm_stress
    nb = 6
    for name in names:
        for nch in (2, 1):
            for mode in ("auto", "short", "mask5", "auto_bfu3"):
                auto, mask, bfu = AT1_MODES[mode]
                pcm = np.stack([at1_blocks(gens[name](34 if name == "stress" else nb // 2 + 1), nch)[:nb],
                                at1_blocks(SIGNALS["mix"](nb // 2 + 1, seed=3), nch)[:nb]])
                t = time.time()
                enc = At1Hip(n_streams=2, max_blocks=nb, channels=nch, window_auto=auto, window_mask=mask, bfu_idx_const=bfu,
                             lib_path=run_emu.EMU)
                got = np.concatenate([enc.encode(pcm[:, :4]), enc.encode(pcm[:, 4:])], axis=1)
                specs = enc.read_tap(At1Hip.TAP_SPECTRA, np.float32, (2, nb - 4, nch, 512))
                masks = enc.read_tap(At1Hip.TAP_MASKS, np.int32, (2, nb - 4, nch))
                loud = enc.read_tap(At1Hip.TAP_LOUDNESS, np.float32, (2, nb - 4))
                enc.close()
                exp = [at1_oracle_encode(pcm[i], mode, taps=True) for i in range(2)]
                ef = np.stack([e[0] for e in exp])
                bad = (got != ef).any(axis=3)
                es = np.stack([e[1][4:] for e in exp]); em = np.stack([e[2][4:] for e in exp]); el = np.stack([e[3][4:] for e in exp])
                print(f"{name:7s} nch={nch} {mode:10s}: frames bad {int(bad.sum())}/{bad.size} {np.argwhere(bad)[:4].tolist()} "
                      f"specs bad {int((specs.view(np.uint32) != es.view(np.uint32)).sum())} masks bad {int((masks != em).sum())} "
                      f"loud bad {int((loud.view(np.uint32) != el.view(np.uint32)).sum())} ({time.time()-t:.1f}s)")
