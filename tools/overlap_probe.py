"""Development probe: aggregate throughput of G contexts sharing one GPU (each S/G streams), calls queued with
AT3HIP_ASYNC. Usage: python tools/overlap_probe.py [total_streams] [frames]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atracdenc_amd
from bench import synth_pcm
S_total = int(sys.argv[1]) if len(sys.argv) > 1 else 64
F = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = 20
def make(S, seed):
    enc = atracdenc_amd.At3Hip(n_streams=S, max_blocks=F + 1, bitrate=132300)
    host = synth_pcm(S, 2 * F + 1, seed=seed)
    prime = torch.from_numpy(host[:, :1].copy()).cuda()
    bat = [torch.from_numpy(host[:, 1 + i * F: 1 + (i + 1) * F].copy()).cuda() for i in range(2)]
    out = torch.zeros((S, F, 384), dtype=torch.uint8, device="cuda")
    enc.encode_device(prime.data_ptr(), 1, out.data_ptr())
    for i in range(2): enc.encode_device(bat[i].data_ptr(), F, out.data_ptr())
    return enc, bat, out
for G in (1, 2, 4):
    ctxs = [make(S_total // G, 10 + i) for i in range(G)]
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(K):
        for enc, bat, out in ctxs:
            enc.encode_device(bat[i % 2].data_ptr(), F, out.data_ptr(), asynchronous=True)
    for enc, _, _ in ctxs: enc.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    k1 = np.mean([ctxs[0][0].timings_ago(a)["qmf_mdct_ms"] for a in range(5)])
    print(G, "contexts:", round(S_total * F * K / dt), "frames/s aggregate,", round(dt / K * 1e3, 3), "ms per step, K1 per launch", round(k1, 4), "ms")
    for c in ctxs: c[0].close()
