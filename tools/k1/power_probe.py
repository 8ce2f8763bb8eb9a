"""GPU box: shader clock and board power while the fused kernel runs back to back (shard size), for white noise and for silence.
usage: python tools/k1/power_probe.py lib.so [lib2.so ...]"""
import os, sys, subprocess, threading, time, re
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
import torch
from atracdenc_amd import binding as B

def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:
        return str(e)
    keep = [l.strip() for l in o.splitlines() if re.search(r"sclk|Power|Temperature \(Sensor (junction|edge)", l)]
    return " | ".join(k.split("GPU[0]")[-1].strip(" :\t") for k in keep)

S, F = 1024, 128
nb = F + 1
specs = torch.zeros((S, F, 2, 1024), dtype=torch.float32, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
noise = (torch.randint(-8192, 8192, (S, nb, 1024, 2), generator=g, device="cuda", dtype=torch.int32).to(torch.float32) / 32768.0).contiguous()
silence = torch.zeros_like(noise)
small = (noise * 1e-3).contiguous()
print("idle:", smi())
for lib in sys.argv[1:]:
    for name, pcm in (("noise", noise), ("silence", silence), ("noise x 1e-3", small), ("noise", noise)):
        enc = B.At3Hip(n_streams=S, max_blocks=nb, no_gain=True, lib_path=os.path.abspath(lib))
        ms = []
        stop = [False]
        samples = []
        def sampler():
            while not stop[0]:
                samples.append(smi()); time.sleep(0.25)
        th = threading.Thread(target=sampler); th.start()
        t0 = time.time()
        while time.time() - t0 < 2.5:
            enc.qmf_mdct_device(pcm.data_ptr(), nb, specs.data_ptr())
            ms.append(enc.timings()["qmf_mdct_ms"])
        stop[0] = True; th.join()
        enc.close()
        ms = np.array(ms) * 1e3
        print("%s %-14s: %d launches, kernel min %.1f med %.1f us (first 5: %s)" % (os.path.basename(lib), name, len(ms), ms.min(), np.median(ms), np.round(ms[:5], 1)))
        for s_ in samples[-3:]: print("     ", s_)
