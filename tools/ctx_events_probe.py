"""GPU box: does a process that has used many timing events BEFORE creating its first context get a slow first context? (hypothesis test)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import torch, bench
a = sys.argv[1:]
n_ev = int(a[a.index("--events") + 1]) if "--events" in a else 0
n_streams = int(a[a.index("--streams") + 1]) if "--streams" in a else 0
keep = []
if n_ev:
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev)]
    for e in evs: e.record()
    torch.cuda.synchronize()
    keep.append(evs)
if n_streams:
    ss = [torch.cuda.Stream(priority=-1) for _ in range(n_streams)] + [torch.cuda.Stream(priority=0) for _ in range(n_streams)]
    x = torch.zeros(1024, device="cuda")
    for s in ss:
        with torch.cuda.stream(s): x += 1
    torch.cuda.synchronize()
    keep.append(ss)
job = bench.DeviceJob(0, 64, 64, bench.LP2, False, "tones", seed=1)
job.warmup(5)
r = []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); job.run_steps(150); r.append(64 * 64 * 150 / (time.perf_counter() - t0) / 1e6)
print("events before %d, streams before %d: first context %.2f M frames/s (max %.2f)" % (n_ev, 2 * n_streams, sorted(r)[2], max(r)))
