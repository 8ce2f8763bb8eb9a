"""GPU box: ONE context, the same PCM and output bytes in several tensor sets at different addresses (all alive): does the rate follow the
caller's tensors' placement? usage: python tools/ctx_place_probe.py [--input tones] [--sets 6] [--pad-mb 3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import torch, bench
a = sys.argv[1:]
kind = a[a.index("--input") + 1] if "--input" in a else "tones"
nsets = int(a[a.index("--sets") + 1]) if "--sets" in a else 6
pad_mb = float(a[a.index("--pad-mb") + 1]) if "--pad-mb" in a else 3
job = bench.DeviceJob(0, 64, 64, bench.LP2, False, kind, seed=1)
job.warmup(5)
def rate():
    best = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(150): job.step(True)
        job.enc.sync(); torch.cuda.synchronize()
        best.append(64 * 64 * 150 / (time.perf_counter() - t0))
    return sorted(best)[1] / 1e6
sets = [(job.d_batches, job.d_out)]
pads = []
for i in range(1, nsets):
    pads.append(torch.empty(int(pad_mb * (1 << 20) * i), dtype=torch.uint8, device=job.dev))   # shifts what follows
    sets.append(([b.clone() for b in sets[0][0]], sets[0][1].clone()))
def show(tag):
    p = [t.data_ptr() for t in job.d_batches] + [job.d_out.data_ptr()]
    print("%-22s %.2f M   pcm %x %x  out %x" % (tag, rate(), p[0], p[1], p[2]), flush=True)
for rnd in range(2):
    for i, (b, o) in enumerate(sets):
        job.d_batches, job.d_out = b, o
        show("set %d" % i)
# mixed: the first set's PCM with every set's output tensor, and the other way round
for i, (b, o) in enumerate(sets):
    job.d_batches, job.d_out = sets[0][0], o
    show("pcm 0 + out %d" % i)
for i, (b, o) in enumerate(sets):
    job.d_batches, job.d_out = b, sets[0][1]
    show("pcm %d + out 0" % i)
