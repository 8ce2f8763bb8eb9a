"""GPU box: ONE context, one job: measure, at3hip_reset + look-ahead call, measure again, ... Is the 'later context' slowdown a 'later stream' slowdown?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import torch, bench
kind = sys.argv[1] if len(sys.argv) > 1 else "tones"
job = bench.DeviceJob(0, 64, 64, bench.LP2, False, kind, seed=1)
def measure(tag):
    job.warmup(5)
    r = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); job.run_steps(150); r.append(64 * 64 * 150 / (time.perf_counter() - t0) / 1e6)
    print("%s: %.2f M frames/s (calls so far %d)" % (tag, sorted(r)[1], job.calls), flush=True)
measure("fresh context")
for k in range(3):
    job.enc.reset(); job.calls = 0
    job.enc.encode_device(job.d_prime.data_ptr(), 1, job.d_out.data_ptr())
    measure("after reset %d" % (k + 1))
# odd number of calls before the measurement: the batches swap roles
job.enc.reset(); job.calls = 1
job.enc.encode_device(job.d_prime.data_ptr(), 1, job.d_out.data_ptr())
measure("after reset, batches in the other order")
# new caller-side tensors for the same context
keep_old = (job.d_prime, job.d_batches, job.d_out)
pcm = bench.synth_pcm_device(kind, 64, 2 * 64 + 1, 1, job.dev)
job.d_prime = pcm[:, :1].contiguous()
job.d_batches = [pcm[:, 1 + i * 64: 1 + (i + 1) * 64].contiguous() for i in range(2)]
del pcm
job.d_out = torch.zeros_like(job.d_out)
job.enc.reset(); job.calls = 0
job.enc.encode_device(job.d_prime.data_ptr(), 1, job.d_out.data_ptr())
measure("same context, NEW tensors (pcm %s out %s)" % (hex(job.d_batches[0].data_ptr()), hex(job.d_out.data_ptr())))
# a second context now, the first still alive
job2 = bench.DeviceJob(0, 64, 64, bench.LP2, False, kind, seed=1)
old = job; job = job2
measure("second context")
job = old
measure("first context again")

job.d_prime, job.d_batches, job.d_out = keep_old
job.enc.reset(); job.calls = 0
job.enc.encode_device(job.d_prime.data_ptr(), 1, job.d_out.data_ptr())
measure("first context, its FIRST tensors again")
job2.d_prime, job2.d_batches, job2.d_out = keep_old
job = job2
job.enc.reset(); job.calls = 0
job.enc.encode_device(job.d_prime.data_ptr(), 1, job.d_out.data_ptr())
measure("second context with the first tensors")
