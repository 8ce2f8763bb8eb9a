#!/usr/bin/env python3
"""bench.py - ATRAC3 LP2 stereo encode throughput on MI355X (BASELINE.json metric).

A "step" = one pass of the whole hot path (QMF -> gain control -> MDCT -> psy -> bit allocation ->
quantisation -> sound-unit packing) over one batch of synthetic PCM that is already resident in HBM:
`--streams` independent streams x `--frames` new 1024-sample stereo frames each (default 64 x 64 = 4096
frames = BASELINE configs[1]). Streams continue across steps (the encoder carries its state), so every
step does identical, full work and emits streams*frames ATRAC3 frames into device memory.

N GPUs: one process per GPU (torch.distributed / RCCL only for the barrier and the MAX of the elapsed
time); streams are sharded, per-GPU work is fixed ("weak" scaling), there is no data-path collective.

One JSON line on rank 0. Extra objects:
  roofline     - fused QMF+MDCT kernel (k_qmf_mdct): algorithmic 16384 B/frame x frames per launch /
                 average launch duration measured with HIP events on the ctx stream inside the timed region
                 (where it co-runs with the previous step's back half); "isolated" = the same launch alone.
  cpu_baseline - the real reference (oracle/_ref, kind "reference") or the C port (oracle/, kind "port")
                 timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_FRAME_K1 = 16384   # 8192 B interleaved PCM in + 8192 B spectra out
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def synth_pcm(n_streams, n_blocks, seed):
    """Uniform white noise, s16 in [-8192, 8191] / 32768 (SURVEY.md 8(d) 'noise'), per-stream seeds."""
    rng = np.random.RandomState(seed)
    return (rng.randint(-8192, 8192, size=(n_streams, n_blocks, 1024, 2)).astype(np.float32)
            / np.float32(32768.0)).astype(np.float32)


def cpu_baseline(seconds_budget=12.0):
    """Reference encoder on host cores, bounded sample; returns the dict for the JSON line."""
    from concurrent.futures import ThreadPoolExecutor
    import at3_testlib as tl
    if tl.have_ref():
        codec, kind = tl.ref(), "reference"
    else:
        tl.build_oracle()
        codec, kind = tl.oracle(), "port"
    probe = synth_pcm(1, 201, 12345)[0]
    codec.encode(probe[:3])  # one-time table init, single threaded
    t = time.perf_counter()
    codec.encode(probe)
    per_frame = (time.perf_counter() - t) / 200.0
    nfr = int(max(200, min(20000, seconds_budget * 0.4 / per_frame)))
    pcm = synth_pcm(1, nfr + 1, 777)[0]
    t = time.perf_counter()
    frames, _ = codec.encode(pcm)
    dt1 = time.perf_counter() - t
    single = frames.shape[0] / dt1
    cores = os.cpu_count() or 1
    nthreads = max(1, min(cores, 64))
    nfr_mt = int(max(100, min(nfr, seconds_budget * 0.5 / per_frame)))
    pcms = [synth_pcm(1, nfr_mt + 1, 1000 + i)[0] for i in range(nthreads)]
    t = time.perf_counter()
    with ThreadPoolExecutor(nthreads) as ex:   # ctypes releases the GIL during the foreign call
        res = list(ex.map(lambda p: codec.encode(p)[0].shape[0], pcms))
    dtm = time.perf_counter() - t
    multi = sum(res) / dtm
    return {
        "value": round(single, 1), "unit": "frames/s", "cores": 1, "kind": kind,
        "sample": f"{frames.shape[0]} frames of one LP2 stereo white-noise stream, single thread",
        "all_cores_value": round(multi, 1), "all_cores": nthreads,
        "all_cores_sample": f"{nthreads} threads x {nfr_mt} frames (one stream per thread)",
        "x_realtime": round(single * 1024 / 44100.0, 1),
    }


def widened_rows(S):
    """SURVEY 8(f) rows f3 / f4 measured in the same process, after the headline (not part of `value`): the ATRAC1 encode
    path and the ATRAC3plus front end on the same audio shape (S stereo streams x 65536 samples, PCM resident in HBM)."""
    out = {}
    try:
        import torch
        import atracdenc_amd
        rng = np.random.RandomState(3)
        pcm = torch.from_numpy((rng.randint(-8192, 8192, size=(S, 65536, 2)).astype(np.float32) / np.float32(32768.0))).cuda()

        def timed(fn, steps=10):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps

        e1 = atracdenc_amd.At1Hip(n_streams=S, max_blocks=128)
        o1 = torch.zeros((S, 128, 2, 212), dtype=torch.uint8, device="cuda")
        dt = timed(lambda: e1.encode_device(pcm.data_ptr(), 128, o1.data_ptr()))
        out["atrac1_encode"] = {"value": round(S * 128 / dt, 1), "unit": "512-sample stereo sound-unit pairs/s", "ms_per_step": round(dt * 1e3, 4),
                                "x_realtime": round(S * 65536 / 44100.0 / dt, 1), "device_ms": {k: round(v, 4) for k, v in e1.timings().items()}}
        e1.close()
        ep = atracdenc_amd.At3pHip(n_streams=S, max_frames=32)
        op = torch.zeros((S, 32, 2, 2048), dtype=torch.float32, device="cuda")
        dt = timed(lambda: ep.pqf_mdct_device(pcm.data_ptr(), 32, op.data_ptr()))
        tm = ep.timings()
        out["atrac3plus_pqf_mdct"] = {"value": round(S * 32 / dt, 1), "unit": "2048-sample stereo frames/s", "ms_per_step": round(dt * 1e3, 4),
                                      "x_realtime": round(S * 65536 / 44100.0 / dt, 1), "device_ms": {k: round(v, 4) for k, v in tm.items()},
                                      "algorithmic_GBps": round(S * 32 * 2 * 2048 * 16 / ((tm["pqf_ms"] + tm["mdct_ms"]) * 1e-3) / 1e9, 1)}
        ep.close()
    except Exception as ex:   # the headline line must not depend on these
        out["error"] = repr(ex)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=64, help="streams per GPU")
    ap.add_argument("--frames", type=int, default=64, help="frames per stream per step")
    ap.add_argument("--bitrate", type=int, default=132300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gain", action="store_true")
    args = ap.parse_args()

    import torch
    import atracdenc_amd

    from atracdenc_amd import dist as at3dist
    rank, local_rank, world = at3dist.env_world()
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        dist = at3dist.init("nccl", local_rank)   # RCCL: barrier + MAX of elapsed time only, no data-path collective

    S, F = args.streams, args.frames
    enc = atracdenc_amd.At3Hip(n_streams=S, max_blocks=F + 1, bitrate=args.bitrate, no_gain=args.no_gain,
                               device_id=local_rank)
    fsz = enc.frame_size
    # synthetic PCM resident in HBM before timing: a priming look-ahead block + (warmup+steps) distinct batches
    # would be large; instead two alternating batches are kept resident and re-fed (the encoder does full work).
    host = synth_pcm(S, 2 * F + 1, seed=1 + rank)
    d_prime = torch.from_numpy(host[:, :1].copy()).cuda()
    d_batches = [torch.from_numpy(host[:, 1 + i * F: 1 + (i + 1) * F].copy()).cuda() for i in range(2)]
    d_out = torch.zeros((S, F, fsz), dtype=torch.uint8, device="cuda")
    enc.encode_device(d_prime.data_ptr(), 1, d_out.data_ptr())     # LOOK_AHEAD call, emits nothing
    for i in range(args.warmup):
        n = enc.encode_device(d_batches[i % 2].data_ptr(), F, d_out.data_ptr())
        assert n == F

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region: K steps queued back to back (AT3HIP_ASYNC) and completed by one sync. Inside the context the front
    # half of step i+1 (QMF, gain control, fused QMF+MDCT) runs beside the back half of step i (psychoacoustics,
    # quantisation, rate loop, packing) on a second HIP stream - every step still does its full work on its own batch.
    k1_ms, stage_ms = [], {}
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        enc.encode_device(d_batches[(args.warmup + i) % 2].data_ptr(), F, d_out.data_ptr(), asynchronous=True)
    enc.sync()
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = at3dist.max_over_ranks(elapsed, dist, device="cuda")
    checksum = int(d_out.to(torch.int64).sum().item())
    # after the timed region: the same launch without a co-running back half (synchronous calls), for reference
    iso_ms = []
    for i in range(3):
        enc.encode_device(d_batches[(args.warmup + args.steps + i) % 2].data_ptr(), F, d_out.data_ptr())
        iso_ms.append(enc.timings()["qmf_mdct_ms"])
    n_timed = min(args.steps, 28)           # per-step HIP-event timings of the timed region (history of 32 calls)
    first_ago = 3                           # the three reference calls above are the most recent ones
    for ago in range(first_ago, first_ago + n_timed):
        tm = enc.timings_ago(ago)
        k1_ms.append(tm["qmf_mdct_ms"] / max(1, tm["qmf_mdct_launches"]))
        for k, v in tm.items():
            if k.endswith("_ms"):
                stage_ms[k] = stage_ms.get(k, 0.0) + v

    if rank == 0:
        frames_total = world * S * F * args.steps
        value = frames_total / elapsed
        k1_avg_ms = float(np.mean(k1_ms))
        achieved = ALGO_BYTES_PER_FRAME_K1 * S * F / (k1_avg_ms * 1e-3) / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get("bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "ATRAC3 1024-sample stereo frames/sec", "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "x_realtime": round(value * 1024 / 44100.0, 1),
            "config": {"workload": f"ATRAC3 {'LP2 132 kbps' if fsz == 384 else str(fsz) + ' B/frame'} stereo, "
                                   f"{S} streams x {F} frames = {S * F} frames per GPU per step, white-noise PCM "
                                   f"resident in HBM, frames written to HBM (PCIe excluded)",
                       "streams_per_gpu": S, "frames_per_stream_per_step": F, "frame_bytes": fsz,
                       "gain_control": not args.no_gain, "tonal_components": True, "parallelism": f"streams/{world}"},
            "roofline": {"bound": "hbm", "kernel": "k_qmf_mdct (fused QMF + gain modulation + windowed MDCT-512)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME_K1 * S * F,
                         "avg_launch_ms": round(k1_avg_ms, 5),
                         "note": "launches of the timed region; they share the GPU with the previous step's back half "
                                 "(quantisation / rate loop) running on the context's second stream",
                         "isolated": {"avg_launch_ms": round(float(np.mean(iso_ms)), 5),
                                      "achieved": round(ALGO_BYTES_PER_FRAME_K1 * S * F / (float(np.mean(iso_ms)) * 1e-3) / 1e9, 2),
                                      "frac": round(ALGO_BYTES_PER_FRAME_K1 * S * F / (float(np.mean(iso_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                      "note": "same kernel, same batch, launched alone (3 synchronous steps after the timed region)"}},
            "stage_ms_per_step": {k: round(v / max(1, len(k1_ms)), 4) for k, v in sorted(stage_ms.items())},
            "pipelining": "front half of step i+1 overlaps the back half of step i (two HIP streams inside the context); "
                          "stage_ms are per-step HIP-event spans and overlap in time, total_ms is one step's latency",
            "checksum": checksum,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        if world == 1:
            line["widened_rows"] = widened_rows(S)
        print(json.dumps(line))
    enc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
