"""GPU box: second context after the first one RAN - with torch's caching allocator emptied in between, or the job's tensors allocated before the context."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import torch, bench
mode = sys.argv[1] if len(sys.argv) > 1 else "empty_cache"
def run(tag):
    job = bench.DeviceJob(0, 64, 64, bench.LP2, False, "tones", seed=1)
    job.warmup(5)
    r = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); job.run_steps(150); r.append(64 * 64 * 150 / (time.perf_counter() - t0) / 1e6)
    print("%s %s: %.2f M frames/s; pcm %s %s out %s" % (mode, tag, sorted(r)[2], hex(job.d_batches[0].data_ptr()), hex(job.d_batches[1].data_ptr()), hex(job.d_out.data_ptr())))
    return job
j = run("first")
if mode == "keep_first_tensors":
    keep = (j.d_batches, j.d_out, j.d_prime)
j.close(); del j
if mode == "empty_cache":
    import gc; gc.collect(); torch.cuda.empty_cache()
run("second")
