"""GPU box: is the second context slow even when the first one never ran anything? And does a context created BEFORE torch touches the device differ?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import atracdenc_amd
mode = sys.argv[1] if len(sys.argv) > 1 else "second_unused_first"
pre = []
if mode == "second_unused_first":
    pre.append(atracdenc_amd.At3Hip(n_streams=64, max_blocks=65))          # created, never used, kept alive
elif mode == "second_closed_first":
    atracdenc_amd.At3Hip(n_streams=64, max_blocks=65).close()
elif mode == "small_first":
    pre.append(atracdenc_amd.At3Hip(n_streams=1, max_blocks=4))
elif mode == "at1_first":
    pre.append(atracdenc_amd.At1Hip(n_streams=1, max_blocks=8))
elif mode == "small_ran_first":
    import numpy as np
    e = atracdenc_amd.At3Hip(n_streams=1, max_blocks=4)
    e.encode(np.zeros((1, 4, 1024, 2), np.float32)); e.close()
elif mode == "small_nogain_ran_first":
    import numpy as np
    e = atracdenc_amd.At3Hip(n_streams=1, max_blocks=4, no_gain=True)
    e.encode(np.zeros((1, 4, 1024, 2), np.float32)); e.close()
elif mode == "torch_streams_first":
    import torch
    ss = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
    x = torch.zeros(1 << 20, device="cuda")
    for s_ in ss:
        with torch.cuda.stream(s_):
            for _ in range(50): x += 1
    torch.cuda.synchronize(); del ss
import torch, bench
job = bench.DeviceJob(0, 64, 64, bench.LP2, False, "tones", seed=1)
job.warmup(5)
r = []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); job.run_steps(150); r.append(64 * 64 * 150 / (time.perf_counter() - t0) / 1e6)
print("%s: %.2f M frames/s (max %.2f)" % (mode, sorted(r)[2], max(r)))
