"""GPU box: a fresh process runs the bench's call pattern once on a fresh context and reports whether the tracked loudness stayed sane
(the eight-rank failure: TrackLoudness state in the tens of thousands). usage: python tools/loud_detect.py (run several at once)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/.."); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import numpy as np, torch, bench
from atracdenc_amd import binding as B
if "RANK" in os.environ:   # under torch.distributed.run: the bench's own start (gloo group, a barrier in front of the job)
    import torch.distributed as dist
    dist.init_process_group(backend="gloo"); dist.barrier()
job = bench.DeviceJob(0, 64, 16, bench.LP2, False, "noise", seed=1 + int(os.environ.get("RANK", os.getpid() % 7)) * 64)
import time
def barrier(q=0.2):   # the ranks of the bench meet in a gloo barrier before every region: here all processes start on the same tick of the wall clock
    t = (int(time.time() / q) + 1) * q
    while time.time() < t: pass
barrier(2.0); job.warmup(1); barrier(); job.run_steps(3); job.enc.sclk_mhz(); barrier(); job.run_steps(3); job.k1_stats(8, 0); job.isolated_k1()
lo = job.enc.read_tap(B.TAP_LOUDNESS, np.float32, (64, 16))
print("LOUD", "BAD" if not (lo.max() < 1.0) else "ok", float(lo.max()), flush=True)
