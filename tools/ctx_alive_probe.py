"""GPU box: three contexts of one process kept alive, the same workload measured on them in turn: is a context's speed its own property?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import torch, bench
a = sys.argv[1:]
kind = a[a.index("--input") + 1] if "--input" in a else "tones"
jobs = []
for i in range(3):
    j = bench.DeviceJob(0, 64, 64, bench.LP2, False, kind, seed=1)
    j.warmup(5)
    jobs.append(j)
    print("context %d: psy buffer %s" % (i + 1, hex(j.d_out.data_ptr())))
for rnd in range(3):
    for i, job in enumerate(jobs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        job.run_steps(150)
        dt = time.perf_counter() - t0
        print("round %d context %d (%s): %.2f M frames/s" % (rnd, i + 1, kind, 64 * 64 * 150 / dt / 1e6), flush=True)
