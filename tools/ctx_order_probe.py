"""GPU box: frames/s of the SAME workload in the first, second, third ... context a process creates (later contexts ran
tonal material 10 - 25 % slower). usage: python tools/ctx_order_probe.py [--input tones] [--n 4] [--keep]   (--keep: earlier contexts stay alive)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import torch, bench
a = sys.argv[1:]
kind = a[a.index("--input") + 1] if "--input" in a else "tones"
n = int(a[a.index("--n") + 1]) if "--n" in a else 4
keep = "--keep" in a
alive = []
for i in range(n):
    job = bench.DeviceJob(0, 64, 64, bench.LP2, False, kind, seed=1)
    job.warmup(5)
    best = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(150): job.step(True)
        t_enq = time.perf_counter() - t0
        job.enc.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best.append(64 * 64 * 150 / dt)
        enq = t_enq / 150 * 1e6
    print("context %d of the process (%s): %.2f M frames/s median, %.2f max; host enqueue %.0f us per call (step %.0f us)" % (i + 1, kind, sorted(best)[len(best) // 2] / 1e6, max(best) / 1e6, enq, 4096 / (sorted(best)[len(best) // 2]) * 1e6), flush=True)
    if keep: alive.append(job)
    else: job.close()
