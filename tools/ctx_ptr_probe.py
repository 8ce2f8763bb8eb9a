"""GPU box: the caller-side tensors' addresses of consecutive jobs of one process beside their rates (later-context slowdown)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import torch, bench
kind = "tones"
def rate(job):
    best = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(150): job.step(True)
        job.enc.sync(); torch.cuda.synchronize()
        best.append(64 * 64 * 150 / (time.perf_counter() - t0))
    return sorted(best)[1] / 1e6
for i in range(4):
    job = bench.DeviceJob(0, 64, 64, bench.LP2, False, kind, seed=1)
    job.warmup(5)
    ptrs = [t.data_ptr() for t in job.d_batches] + [job.d_out.data_ptr()]
    print("job %d: %.2f M  " % (i + 1, rate(job)) + "  ".join("%x (mod 2M %x)" % (p, p % (2 << 20)) for p in ptrs), "reserved %.0f MB" % (torch.cuda.memory_reserved() / 1e6), flush=True)
    try: clk = " sclk %.0f MHz" % job.enc.sclk_mhz()
    except Exception as ex: clk = " sclk ? (%r)" % ex
    print("   ", clk, flush=True)
    job.close()
    if "--sleep" in sys.argv: time.sleep(float(sys.argv[sys.argv.index("--sleep") + 1]))
    if "--empty" in sys.argv: torch.cuda.empty_cache()
