"""GPU box: ONE context; the caller's three buffers (look-ahead block, PCM batches, frame output) of a FIRST and of a LATER generation in every combination."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import torch, bench
kind = sys.argv[1] if len(sys.argv) > 1 else "tones"
job = bench.DeviceJob(0, 64, 64, bench.LP2, False, kind, seed=1)
def measure(tag):
    job.enc.reset(); job.calls = 0
    job.enc.encode_device(job.d_prime.data_ptr(), 1, job.d_out.data_ptr())
    job.warmup(5)
    r = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); job.run_steps(150); r.append(64 * 64 * 150 / (time.perf_counter() - t0) / 1e6)
    print("%-34s %.2f M frames/s   prime %s pcm %s %s out %s" % (tag, sorted(r)[1], hex(job.d_prime.data_ptr()), hex(job.d_batches[0].data_ptr()), hex(job.d_batches[1].data_ptr()), hex(job.d_out.data_ptr())), flush=True)
measure("first generation")
old = (job.d_prime, job.d_batches, job.d_out)
job.run_steps(300)
pcm = bench.synth_pcm_device(kind, 64, 2 * 64 + 1, 1, job.dev)
new = (pcm[:, :1].contiguous(), [pcm[:, 1 + i * 64: 1 + (i + 1) * 64].contiguous() for i in range(2)], torch.zeros_like(job.d_out))
del pcm
for combo in itertools.product((0, 1), repeat=3):
    job.d_prime = (old, new)[combo[0]][0]
    job.d_batches = (old, new)[combo[1]][1]
    job.d_out = (old, new)[combo[2]][2]
    measure("prime/pcm/out generation %d%d%d" % combo)
