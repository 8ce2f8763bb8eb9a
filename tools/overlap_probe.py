import sys, os, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import atracdenc_amd
from bench import synth_pcm
S, F, K = 64, 64, 20
def make(seed):
    enc = atracdenc_amd.At3Hip(n_streams=S, max_blocks=F + 1, bitrate=132300)
    host = synth_pcm(S, 2 * F + 1, seed=seed)
    prime = torch.from_numpy(host[:, :1].copy()).cuda()
    bat = [torch.from_numpy(host[:, 1 + i * F: 1 + (i + 1) * F].copy()).cuda() for i in range(2)]
    out = torch.zeros((S, F, 384), dtype=torch.uint8, device="cuda")
    enc.encode_device(prime.data_ptr(), 1, out.data_ptr())
    for i in range(2): enc.encode_device(bat[i].data_ptr(), F, out.data_ptr())
    return enc, bat, out
def run(ctx, n):
    enc, bat, out = ctx
    for i in range(n): enc.encode_device(bat[i % 2].data_ptr(), F, out.data_ptr())
for nctx in (1, 2, 3):
    ctxs = [make(10 + i) for i in range(nctx)]
    torch.cuda.synchronize()
    t = time.perf_counter()
    th = [threading.Thread(target=run, args=(c, K)) for c in ctxs]
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(nctx, "contexts:", round(nctx * S * F * K / dt), "frames/s aggregate,", round(dt / K * 1e3, 3), "ms per round")
    for c in ctxs: c[0].close()
