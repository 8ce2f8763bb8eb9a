"""GPU box: the fused QMF + MDCT kernel alone (at3hip_qmf_mdct: HIP events round the one launch) for several builds of the library,
alternating on the same box, at configs[1]'s size and at the per-GPU shard of configs[2]; a checksum of the spectra says whether a
variant still produces the baseline's bits.   usage: python tools/k1/lab.py [--sizes 64x64,1024x128] [--runs R,..] lib1.so lib2.so ..."""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
import torch
from atracdenc_amd import binding as B
B.AT3HIP_VERSION = (1 << 16) | 3   # (the lab also loads builds of earlier rounds for same-box comparisons)

def main():
    a = sys.argv[1:]
    sizes = "64x64,1024x128"
    runs = [0]
    chains = [0]
    cold = False   # --cold: 768 MiB written between launches (the PCM comes out of HBM, not the Infinity Cache) and the device idles a moment: what a kernel of a synchronous pipeline step sees
    reps = 3
    libs = []
    i = 0
    while i < len(a):
        if a[i] == "--sizes": sizes = a[i + 1]; i += 2
        elif a[i] == "--runs": runs = [int(x) for x in a[i + 1].split(",")]; i += 2
        elif a[i] == "--reps": reps = int(a[i + 1]); i += 2
        elif a[i] == "--chain": chains = [int(x) for x in a[i + 1].split(",")]; i += 2
        elif a[i] == "--cold": cold = True; i += 1
        else: libs.append(a[i]); i += 1
    sizes = [tuple(int(x) for x in s.split("x")) for s in sizes.split(",")]
    res = {}
    sums = {}
    for S, F in sizes:
        nb = F + 1
        g = torch.Generator(device="cuda"); g.manual_seed(1)
        pcm = (torch.randint(-8192, 8192, (S, nb, 1024, 2), generator=g, device="cuda", dtype=torch.int32).to(torch.float32) / 32768.0).contiguous()
        specs = torch.zeros((S, F, 2, 1024), dtype=torch.float32, device="cuda")
        n_iter = 400 if S * F <= 8192 else 80   # (the first launches of a burst run in the power manager's transient: only the second half is kept)
        if cold:
            n_iter = 60 if S * F <= 8192 else 20
            scrub = torch.zeros(768 << 20, dtype=torch.uint8, device="cuda")
        for rep in range(reps):
            for lib in libs:
                for r, cm in [(r, cm) for r in runs for cm in chains]:
                    enc = B.At3Hip(n_streams=S, max_blocks=nb, no_gain=True, lib_path=os.path.abspath(lib))
                    if r: enc.set_option(1, r)
                    if cm:
                        try: enc.set_option(6, cm)
                        except Exception: pass
                    r = r * 10 + cm
                    specs.zero_()
                    ms = []
                    for it in range(n_iter):
                        if cold:
                            scrub.add_(1)
                            torch.cuda.synchronize()
                        enc.qmf_mdct_device(pcm.data_ptr(), nb, specs.data_ptr())
                        ms.append(enc.timings()["qmf_mdct_ms"])
                    torch.cuda.synchronize()
                    v = specs.view(torch.int32).to(torch.int64)
                    cs = int((v * (torch.arange(v.numel(), device="cuda").view(v.shape) % 1000003 + 1)).sum().item()) & 0xffffffffffff
                    if "stamps" in lib and rep == 0:
                        raw = enc.read_tap(B.TAP_CLOCK, np.uint64, (16 + 256 * 24,))[16:].reshape(256, 24)
                        inv = lambda v: int(v) ^ 0xffffffffffffffff
                        live = [r for r in raw if int(r[13]) != 0]
                        t0 = min(inv(r[12]) for r in live)
                        print("  last launch: wavefront starts span %.2f us; first end %.2f us, last end %.2f us after the first start" % (
                            (max(int(r[13]) for r in live) - t0) / 100.0, (min(inv(r[14]) for r in live) - t0) / 100.0, (max(int(r[15]) for r in live) - t0) / 100.0))
                        c = raw.astype(np.float64).sum(axis=0)
                        waves, blocks = c[8], c[9]
                        names = ["prologue", "hist/tile in, fetch", "stage 1", "stage 2 + subbands out", "gather (+ early tile)", "MDCT + store"]
                        print("  %s %dx%d runs=%d: %d wavefront-launches, %.1f blocks per wavefront, life %.0f cycles, sclk %.0f MHz" % (
                            os.path.basename(lib), S, F, r, waves, blocks / waves, c[6] / waves, 100.0 * c[6] / max(c[7], 1)))
                        names.append("hand-over + deferred frame")
                        ph = list(c[:6]) + [c[10]]
                        for k in range(7):
                            print("    %-26s %9.0f cycles per wavefront  %8.0f per block  %5.1f %%" % (names[k], ph[k] / waves, ph[k] / blocks, 100 * ph[k] / sum(ph)))
                    enc.close()
                    key = (S, F, os.path.basename(lib), r)
                    res.setdefault(key, []).extend(ms[len(ms) // 2:] if len(ms) > 1 else ms)
                    sums[key] = cs
    base = {}
    for (S, F, lib, r), ms in res.items():
        ms = np.array(ms) * 1e3
        cs = sums[(S, F, lib, r)]
        base.setdefault((S, F), cs)
        by = S * F * 16384
        print("%5dx%-4d %-34s runs*10+chain=%-4d min %8.2f us  med %8.2f us  hbm_frac(med) %.3f  %s" % (
            S, F, lib, r, ms.min(), np.median(ms), by / (np.median(ms) * 1e-6) / 8e12, "same-bits" if cs == base[(S, F)] else "BITS-DIFFER"))

main()
