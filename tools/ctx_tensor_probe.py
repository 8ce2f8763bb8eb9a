"""GPU box: ONE context; which of the caller's buffers - PCM batches or the frame output - makes a later job slow on `tones`?"""
import os, sys, time
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
    print("%-46s %.2f M frames/s   pcm %s %s out %s" % (tag, sorted(r)[1], hex(job.d_batches[0].data_ptr()), hex(job.d_batches[1].data_ptr()), hex(job.d_out.data_ptr())), flush=True)
measure("first tensors")
old_out = job.d_out
job.d_out = torch.zeros_like(old_out)
measure("new output buffer")
job.d_out = old_out
measure("old output buffer again")
old_b = job.d_batches
job.d_batches = [b.clone() for b in old_b]
measure("cloned PCM batches")
job.d_batches = old_b
measure("old PCM batches again")
# explicit placements: the frame output at offsets inside one big allocation
big = torch.zeros(256 << 20, dtype=torch.uint8, device=job.dev)
n = old_out.numel()
for off in (0, 4096, 1 << 16, 1 << 20, (1 << 21) + 384, 33 << 20, 100 << 20):
    job.d_out = big[off:off + n].view(old_out.shape)
    measure("output at +%d of a 256 MiB block" % off)
job.d_out = old_out
pcm = bench.synth_pcm_device(kind, 64, 2 * 64 + 1, 1, job.dev)
new_b = [pcm[:, 1 + i * 64: 1 + (i + 1) * 64].contiguous() for i in range(2)]
new_p = pcm[:, :1].contiguous()
print("regenerated PCM equals the first:", bool(torch.equal(new_b[0], old_b[0])), bool(torch.equal(new_b[1], old_b[1])), bool(torch.equal(new_p, job.d_prime)))
del pcm
measure("after synth ran again, OLD tensors")
job.d_batches = new_b
measure("regenerated PCM batches")
job.d_batches = old_b
measure("old PCM batches once more")
torch.cuda.empty_cache()
measure("old tensors after empty_cache")
