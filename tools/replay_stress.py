"""GPU box: run K asynchronous steps, replay them from start of stream, compare the last frames' checksums - many times, optionally with
other processes loading the device. usage: python tools/replay_stress.py [--iters 200] [--frames 16] [--steps 7] [--procs 1]"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
a = sys.argv[1:]
arg = lambda n, d: int(a[a.index(n) + 1]) if n in a else d
iters, F, K, procs = arg("--iters", 200), arg("--frames", 16), arg("--steps", 7), arg("--procs", 1)
if procs > 1 and "--child" not in a:
    ps = [subprocess.Popen([sys.executable, __file__, "--iters", str(iters), "--frames", str(F), "--steps", str(K), "--child"] + (["--pattern", a[a.index("--pattern") + 1]] if "--pattern" in a else [])) for _ in range(procs)]
    sys.exit(max(p.wait() for p in ps))
import torch, bench
job = bench.DeviceJob(0, 64, F, bench.LP2, False, "noise", seed=1 + os.getpid() % 7)
bad = 0
sums = set()
pattern = a[a.index("--pattern") + 1] if "--pattern" in a else "replay"
for it in range(iters):
    if "fresh" in pattern:   # a NEW context per iteration: the first pass after at3hip_create is the one that was seen to go wrong
        job.close()
        job = bench.DeviceJob(0, 64, F, bench.LP2, False, "noise", seed=1 + os.getpid() % 7)
    else:
        job.enc.reset(); job.calls = 0
        job.enc.encode_device(job.d_prime.data_ptr(), 1, job.d_out.data_ptr())
    if pattern == "replay":
        job.run_steps(K)
    else:   # what bench.py's timed path does: warm-up, regions with waits between them, the clock tap, the timing getters
        job.warmup(1)
        job.run_steps(3)
        if "noclk" not in pattern: job.enc.sclk_mhz()
        job.run_steps(K - 4)
        if "notm" not in pattern: job.k1_stats(8, 0)
        if "iso" in pattern: job.isolated_k1()
    c = job.checksum()
    if "cmp" in pattern:   # ... and the same number of steps again as bench.py replays them
        c2 = job.replay(job.calls)
        if c2 != c:
            bad += 1
            print("iteration %d: %d steps, timed pattern %d, replay %d" % (it, job.calls, c, c2), flush=True)
        c = c2
    sums.add(c)
print(pattern, os.environ.get("AT3HIP_LIB", "")[-14:], "pid %d: %d iterations of %d async steps (64 x %d frames): %d distinct checksums %s" % (os.getpid(), iters, K, F, len(sums), "OK" if len(sums) == 1 else "NONDETERMINISTIC"), flush=True)
print("pattern mismatches:", bad)
sys.exit(0 if len(sums) == 1 and not bad else 1)
