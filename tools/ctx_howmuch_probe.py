"""GPU box: how much must the FIRST context of a process have run for the second one to be slow on `tones`?  usage: ctx_howmuch_probe.py <steps of the first context> [S] [F]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import torch, bench
n0 = int(sys.argv[1]); S = int(sys.argv[2]) if len(sys.argv) > 2 else 64; F = int(sys.argv[3]) if len(sys.argv) > 3 else 64
kind0 = sys.argv[4] if len(sys.argv) > 4 else "tones"
j0 = bench.DeviceJob(0, S, F, bench.LP2, False, kind0, seed=1)
if n0: j0.run_steps(n0)
j0.close(); del j0
job = bench.DeviceJob(0, 64, 64, bench.LP2, False, "tones", seed=1)
job.warmup(5)
r = []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); job.run_steps(150); r.append(64 * 64 * 150 / (time.perf_counter() - t0) / 1e6)
print("first context: %d x %d %s, %d steps -> second context %.2f M frames/s" % (S, F, kind0, n0, sorted(r)[2]))
