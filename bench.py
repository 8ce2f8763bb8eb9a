#!/usr/bin/env python3
"""bench.py - ATRAC3 stereo encode throughput on MI355X (BASELINE.json metric: 1024-sample stereo frames/s).

A "step" = one pass of the whole hot path (QMF -> gain control -> MDCT -> psy -> bit allocation -> quantisation ->
sound-unit packing) over one batch of synthetic PCM that is already resident in HBM: S independent streams x F new
1024-sample stereo frames each, per GPU. Streams continue across steps (the encoder carries its state), so every step
does identical, full work and emits S*F ATRAC3 frames into device memory.

Workloads (BASELINE.json configs):
  N = 1   configs[1]: LP2 132 kbps, 64 streams x 64 frames = 4096 frames per step
  N > 1   configs[2]: LP2, 8192 streams x 128 frames = 1 048 576 frames on 8 GPUs, i.e. 1024 streams x 128 frames PER GPU;
          the same per-GPU shard is used for N = 2 and 4 ("weak" scaling, per-GPU work fixed for N > 1). The N = 1 line
          carries the rate of that shard on one GPU (`other_workloads.shard_1024x128`), which is the like-for-like
          single-GPU reference for the N > 1 lines.

Multi-GPU (streams are independent: sharded, no data-path collective, NO RCCL):
  * `python bench.py --gpus N` alone: ONE process drives N devices - one at3hip context and one host thread per
    device, every thread queues its K asynchronous steps and waits for its device. Fails loudly when fewer than N
    devices are visible.
  * under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: one rank per GPU; the ranks only
    share a start/stop barrier and the MAX of the elapsed time over the CPU-side gloo backend.

One JSON line on rank 0. Extra objects:
  roofline        - the batched QMF + MDCT work (two kernels around the gain analysis: k_qmf_sub8 + k_mdct_sub; one fused kernel
                    without gain control): algorithmic 16384 B/frame x frames per step / its duration measured with HIP
                    events on the ctx stream inside the timed region (where it co-runs with the previous step's back half);
                    "isolated" = the same launches alone.
  cpu_baseline    - the real reference (oracle/_ref, kind "reference") or the C port (oracle/, kind "port") timed on this
                    box's host cores on a bounded sample of the same workload.
  other_workloads - (N = 1 only, after the headline, never part of `value`) SURVEY 8(d)'s other inputs and shapes.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_FRAME_K1 = 16384   # 8192 B interleaved PCM in + 8192 B spectra out
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LP2, LP4 = 132300, 66150


def synth_pcm(n_streams, n_blocks, seed):
    """Host-side uniform white noise, s16 in [-8192, 8191] / 32768 (SURVEY.md 8(d) 'noise') for the CPU baseline."""
    rng = np.random.RandomState(seed)
    return (rng.randint(-8192, 8192, size=(n_streams, n_blocks, 1024, 2)).astype(np.float32)
            / np.float32(32768.0)).astype(np.float32)


def synth_pcm_device(kind, n_streams, n_blocks, seed, device):
    """SURVEY.md 8(d) synthetic inputs generated in HBM: float32 [S][n_blocks][1024][2] = s16 / 32768.
    noise: uniform s16 in [-8192, 8191]; burst: 3 kHz sine whose amplitude toggles 0.02 / 0.6 every 3000 samples,
    right = 0.5 left (drives the gain-control path); tones: five stationary sines at 0.1 (drives tonal extraction).
    Streams differ by seed (noise) or by a per-stream phase (burst, tones)."""
    import torch
    n = n_blocks * 1024
    if kind == "noise":
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        x = torch.randint(-8192, 8192, (n_streams, n_blocks, 1024, 2), generator=g, device=device, dtype=torch.int32)
        return (x.to(torch.float32) / 32768.0).contiguous()
    out = torch.empty((n_streams, n, 2), dtype=torch.float32, device=device)
    t = torch.arange(n, dtype=torch.float64, device=device)
    chunk = 64
    for s0 in range(0, n_streams, chunk):       # float64 phases, bounded temporaries
        s1 = min(n_streams, s0 + chunk)
        sid = torch.arange(s0, s1, dtype=torch.float64, device=device)[:, None] + 7.0 * (seed % 13)
        if kind == "burst":
            tt = t[None, :] + 97.0 * sid
            amp = torch.where((torch.floor(tt / 3000.0) % 2) == 0, 0.02, 0.6)
            left = amp * torch.sin(2 * np.pi * 3000.0 * tt / 44100.0)
            right = 0.5 * left
        elif kind == "tones":
            left = torch.zeros((s1 - s0, n), dtype=torch.float64, device=device)
            right = torch.zeros_like(left)
            for i, f in enumerate((440.0, 1000.0, 3000.0, 7000.0, 11000.0)):
                ph = 2 * np.pi * f * t[None, :] / 44100.0 + 0.05 * sid
                left += 0.1 * torch.sin(ph)
                right += 0.1 * torch.sin(ph + 0.3 * i)
        else:
            raise ValueError(kind)
        x = torch.stack([left, right], dim=-1)
        x = torch.round(torch.clamp(x, -1.0, 32767.0 / 32768.0) * 32768.0) / 32768.0
        out[s0:s1] = x.to(torch.float32)
    return out.reshape(n_streams, n_blocks, 1024, 2).contiguous()


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


class DeviceJob:
    """One GPU's share: an at3hip context, its PCM batches resident in HBM, and the step loop."""

    def __init__(self, device, S, F, bitrate, no_gain, kind, seed):
        import torch
        import atracdenc_amd
        self.torch = torch
        self.device, self.S, self.F = device, S, F
        self.dev = torch.device("cuda", device)
        self.enc = atracdenc_amd.At3Hip(n_streams=S, max_blocks=F + 1, bitrate=bitrate, no_gain=no_gain, device_id=device)
        if DeviceJob.runs:
            from atracdenc_amd import binding as B
            self.enc.set_option(B.OPT_RUNS, DeviceJob.runs)
        if DeviceJob.gain_form:
            from atracdenc_amd import binding as B
            self.enc.set_option(B.OPT_GAIN_FORM, DeviceJob.gain_form)
        if DeviceJob.gain_wgs:
            from atracdenc_amd import binding as B
            self.enc.set_option(B.OPT_GAIN_WGS_PER_CU, DeviceJob.gain_wgs)
        if DeviceJob.timing_every != 1:
            from atracdenc_amd import binding as B
            self.enc.set_option(B.OPT_TIMING_EVERY, DeviceJob.timing_every)
        self.bitrate, self.no_gain = bitrate, no_gain
        self.fsz = self.enc.frame_size
        # synthetic PCM resident in HBM before timing: a priming look-ahead block + two alternating batches that are
        # re-fed (the encoder does full work on every step; distinct data for every step would only cost memory)
        pcm = synth_pcm_device(kind, S, 2 * F + 1, seed, self.dev)
        self.d_prime = pcm[:, :1].contiguous()
        self.d_batches = [pcm[:, 1 + i * F: 1 + (i + 1) * F].contiguous() for i in range(2)]
        del pcm
        self.d_out = torch.zeros((S, F, self.fsz), dtype=torch.uint8, device=self.dev)
        # The library reads and writes these buffers on ITS streams, which do not wait for torch's: everything torch has queued for them (the
        # synthesis, the copies above, the zero fill) must be done before the first call. Without this wait a fresh process on a crowded device
        # - eight ranks on one GPU - now and then encoded its first call from PCM that was not there yet.
        torch.cuda.synchronize(self.dev)
        self.calls = 0
        self.enc.encode_device(self.d_prime.data_ptr(), 1, self.d_out.data_ptr())     # LOOK_AHEAD call, emits nothing

    def step(self, asynchronous):
        n = self.enc.encode_device(self.d_batches[self.calls % 2].data_ptr(), self.F, self.d_out.data_ptr(),
                                   asynchronous=asynchronous)
        self.calls += 1
        return n

    def warmup(self, w):
        # queued like the timed steps (AT3HIP_ASYNC, one sync at the end): the same launch pattern the timed regions run, so the
        # clocks and the three-stage overlap are what they will be - a synchronous warm-up left the device idle between its steps
        for _ in range(w):
            assert self.step(not DeviceJob.sync_steps) == self.F
        self.enc.sync()
        self.torch.cuda.synchronize(self.dev)

    sync_steps = False   # --sync-steps (profiling aid)
    timing_every = 8     # --timing-every: which steps carry the stage-timing HIP events (AT3HIP_OPT_TIMING_EVERY)
    runs = 0             # --runs (tuning aid: AT3HIP_OPT_RUNS)
    gain_form = 0        # --gain-form (A/B aid: AT3HIP_OPT_GAIN_FORM)
    gain_wgs = 0         # --gain-wgs (tuning aid: AT3HIP_OPT_GAIN_WGS_PER_CU)

    def replay(self, n_steps):
        """Start of stream again (at3hip_reset + the LOOK_AHEAD call), then n_steps steps exactly as the warm-up and the timed
        regions queued them; returns the checksum of the last step's frames. The encoder is deterministic, so this equals
        the checksum after the same number of steps of the original run."""
        self.enc.reset()
        self.calls = 0
        self.enc.encode_device(self.d_prime.data_ptr(), 1, self.d_out.data_ptr())
        self.run_steps(n_steps)
        return self.checksum()

    def parity_sample(self, n_check=4):
        """Start of stream again, the first two steps through the same asynchronous path, each into its own buffer; the first
        n_check streams' 2 F frames against the CPU oracle (tests/at3_testlib: the checker, after the timed region, never
        timed) for the PCM that was actually fed - copied back from the device buffers the timed region read."""
        import at3_testlib as tl
        torch = self.torch
        tl.build_oracle()
        orc = tl.oracle()
        self.enc.reset()
        self.calls = 0
        self.enc.encode_device(self.d_prime.data_ptr(), 1, self.d_out.data_ptr())
        outs = [torch.zeros_like(self.d_out) for _ in range(2)]
        torch.cuda.synchronize(self.dev)   # (torch's zero fills before the library's streams write the same buffers)
        for i in range(2):
            self.enc.encode_device(self.d_batches[i].data_ptr(), self.F, outs[i].data_ptr(), asynchronous=not DeviceJob.sync_steps)
        self.enc.sync()
        torch.cuda.synchronize(self.dev)
        n_check = min(n_check, self.S)
        pcm = torch.cat([self.d_prime[:n_check], self.d_batches[0][:n_check], self.d_batches[1][:n_check]], dim=1).cpu().numpy()
        got = torch.cat([o[:n_check] for o in outs], dim=1).cpu().numpy()
        bad = 0
        for i in range(n_check):
            exp = orc.encode(pcm[i], self.bitrate, int(self.no_gain), 0)[0]
            bad += int((got[i] != exp).any(axis=1).sum())
        return {"streams": n_check, "frames": n_check * 2 * self.F, "mismatching_frames": bad}

    def run_steps(self, k):
        """K steps queued back to back (AT3HIP_ASYNC) and completed by one sync. Inside the context the front half of
        step i+1 (QMF, gain control, fused QMF+MDCT) runs beside the back half of step i (psychoacoustics, quantisation,
        rate loop, packing) on a second HIP stream - every step still does its full work on its own batch."""
        for _ in range(k):
            self.step(not DeviceJob.sync_steps)
        self.enc.sync()
        self.torch.cuda.synchronize(self.dev)

    def k1_stats(self, n_calls, first_ago):
        """The QMF + MDCT launch times and the stage spans of those of the last n_calls steps that carried timing events."""
        k1_ms, stage_ms, n = [], {}, 0
        for ago in range(first_ago, min(first_ago + n_calls, 32)):
            tm = self.enc.timings_ago(ago)
            if tm["qmf_mdct_launches"] == 0:
                continue                                      # a step without timing events (AT3HIP_OPT_TIMING_EVERY)
            n += 1
            k1_ms.append(tm["qmf_ms"] + tm["qmf_mdct_ms"])   # the QMF + MDCT work of one step, one or two kernels
            for k, v in tm.items():
                if k.endswith("_ms"):
                    stage_ms[k] = stage_ms.get(k, 0.0) + v
        return k1_ms, {k: v / max(1, n) for k, v in stage_ms.items()}

    def isolated_k1(self, reps=5):
        from atracdenc_amd import binding as B
        ms = []
        self.enc.set_option(B.OPT_TIMING_EVERY, 1)            # these steps are all timed
        wall, total = [], []
        for _ in range(reps):
            self.torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            self.step(False)                                  # a synchronous call: returns when its frames are in d_out
            wall.append((time.perf_counter() - t0) * 1e3)
            tm = self.enc.timings()
            ms.append(tm["qmf_ms"] + tm["qmf_mdct_ms"])
            total.append(tm["total_ms"])
            self.k1_launches = tm["qmf_mdct_launches"]
        # one call on an idle context: what a caller that does not keep the pipeline full waits for
        self.lone_call = {"host_wall_ms": round(float(np.median(wall)), 4), "device_ms": round(float(np.median(total)), 4),
                          "note": "one synchronous at3hip_encode of this workload on the idle context (median of %d): wall time around the call on the host, "
                                  "and first kernel to last kernel on the device" % reps}
        self.enc.set_option(B.OPT_TIMING_EVERY, DeviceJob.timing_every)
        return float(np.mean(ms))

    def checksum(self):
        return int(self.d_out.to(self.torch.int64).sum().item())

    def close(self):
        self.enc.close()


def _dbg_loud(job, tag):
    """BENCH_DEBUG_TAPS: the tracked loudness after a phase of the run (the eight-rank failure hunt)."""
    if not os.environ.get("BENCH_DEBUG_TAPS"):
        return
    from atracdenc_amd import binding as B
    lo = job.enc.read_tap(B.TAP_LOUDNESS, np.float32, (job.S, job.F))
    if not (lo.max() < 1.0):
        sys.stderr.write("LOUDBAD rank %s after %s (calls %d): max %g min %g\n" % (os.environ.get("RANK", "0"), tag, job.calls, float(lo.max()), float(lo.min())))


def timed_region(jobs, steps, dist):
    """Barrier + synchronize on both sides; returns wall seconds for `steps` steps on every job (max over devices)."""
    import torch
    for j in jobs:
        torch.cuda.synchronize(j.dev)
    if dist is not None:
        dist.barrier()
    if len(jobs) == 1:
        t0 = time.perf_counter()
        jobs[0].run_steps(steps)
        elapsed = time.perf_counter() - t0
    else:
        gate = threading.Barrier(len(jobs) + 1)
        errs = []

        def work(j):
            try:
                gate.wait()
                j.run_steps(steps)
            except Exception as ex:   # noqa: BLE001 - reported by the main thread
                errs.append(repr(ex))

        threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
        for t in threads:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        for t in threads:
            t.join()
        elapsed = time.perf_counter() - t0
        if errs:
            raise SystemExit("device thread failed: " + "; ".join(errs))
    for j in jobs:
        torch.cuda.synchronize(j.dev)
    if dist is not None:
        dist.barrier()
    return elapsed


def kernel_source_sha16():
    """Hash of the kernel sources of this tree (the same function as tools/summarize_prof.py stamps its profiles with)."""
    import hashlib
    csrc = os.path.join(ROOT, "atracdenc_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hpp", ".hip", ".cpp", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def roofline_of(k1_avg_ms, frames_per_launch):
    achieved = ALGO_BYTES_PER_FRAME_K1 * frames_per_launch / (max(k1_avg_ms, 1e-6) * 1e-3) / 1e9
    return round(achieved, 2), round(achieved / HBM_PEAK_GBS, 5)


# algorithmic flops of the QMF + MDCT work per stereo frame (QMF 2 x 100 352 + MDCT 2 x ~26 000), against the fp32
# vector peak WITHOUT FMA: 157.3 TFLOP/s counts a fused multiply-add as two flops; the bit-exactness contract (-ffp-contract=off)
# issues the multiply and the add separately, so the usable peak is half of it. Spec-derived: needs no micro-benchmark.
ALGO_FLOPS_PER_FRAME_K1 = 2 * 100352 + 2 * 26000
FP32_VECTOR_PEAK_TF = 157.3
FP32_NO_FMA_PEAK_TF = FP32_VECTOR_PEAK_TF / 2.0


def flops_of(k1_ms, frames_per_launch):
    tf = ALGO_FLOPS_PER_FRAME_K1 * frames_per_launch / (max(k1_ms, 1e-6) * 1e-3) / 1e12
    return round(tf, 3), round(tf / FP32_NO_FMA_PEAK_TF, 4)


def side_workload(name, S, F, bitrate, kind, steps, warmup, no_gain=False, prior=0):
    """One of SURVEY 8(d)'s other workloads on device 0, after the headline: whole-pipeline rate + K1 launch time.
    prior > 0: that many contexts of the same workload are created, run and closed first (later contexts of a process)."""
    out = {"workload": name}
    try:
        for k in range(prior):
            j0 = DeviceJob(0, S, F, bitrate, no_gain, kind, seed=11)
            j0.warmup(warmup)
            n0 = max(steps, 150)
            dt0 = float(np.median([timed_region([j0], n0, None) / n0 for _ in range(3)]))
            if k == 0:
                out["first_context_value"] = round(S * F / dt0, 1)
            j0.close()
        job = DeviceJob(0, S, F, bitrate, no_gain, kind, seed=11)
        job.warmup(warmup)
        # like the headline: the median of several regions of at least ~50 ms each (one 10 ms region read 3 - 5 % low)
        ms1 = timed_region([job], steps, None) / steps * 1e3
        reg_steps = max(steps, int(np.ceil(50.0 / max(ms1, 1e-6))))
        dts = [timed_region([job], reg_steps, None) / reg_steps for _ in range(3)]
        dt = float(np.median(dts)) * steps
        iso = job.isolated_k1()
        k1_ms, stage = job.k1_stats(min(reg_steps, 27), 5)   # (the five isolated steps are the latest calls)
        k1 = float(np.mean(k1_ms)) if k1_ms else iso
        ach, frac = roofline_of(k1, S * F)
        ach_i, frac_i = roofline_of(iso, S * F)
        out.update({"value": round(S * F * steps / dt, 1), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 4),
                    "frames_per_step": S * F, "frame_bytes": job.fsz, "input": kind,
                    "k1_avg_launch_ms": round(k1, 5), "k1_GBps": ach, "k1_frac": frac,
                    "k1_isolated_ms": round(iso, 5), "k1_isolated_GBps": ach_i, "k1_isolated_frac": frac_i,
                    "k1_isolated_TFLOPs": flops_of(iso, S * F)[0], "k1_isolated_flops_frac": flops_of(iso, S * F)[1],
                    "k1_kernels": "k_qmf_mdct8 (fused QMF + MDCT, one launch)" if getattr(job, "k1_launches", 2) == 1 else "k_qmf_sub8 + k_mdct_sub (two launches)",
                    "gain_control": not no_gain,
                    "stage_ms_per_step": {k: round(v, 4) for k, v in sorted(stage.items())}})
        if prior:
            out["contexts_before_this_one"] = prior
            out["vs_first_context"] = round(out["value"] / out["first_context_value"], 4)
        job.close()
        del job
        import torch
        torch.cuda.empty_cache()
    except Exception as ex:   # noqa: BLE001 - the headline line must not depend on these
        out["error"] = repr(ex)
    return out


def side_workload_fresh(name, S, F, bitrate, kind, steps, warmup, no_gain=False, prior=0):
    """side_workload in a process of its own, as the headline has: every figure of the line is a first context's. (Rounds 5 and 6 believed later contexts
    of a process ran `tones` 10 - 25 % slower; that was this file handing tensors over before torch had filled them - DeviceJob.__init__ waits now, and the
    "third context" entry of `other_workloads` reads what the first does. The fresh processes stay: they cost nothing and keep the entries independent.)
    Falls back to the in-process measurement, and says so, if the child fails."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one-side-workload", json.dumps([name, S, F, bitrate, kind, steps, warmup, bool(no_gain), int(prior)])],
                           capture_output=True, text=True, timeout=600)
        out = json.loads(r.stdout.strip().splitlines()[-1])
        out["fresh_process"] = True
        return out
    except Exception as ex:   # noqa: BLE001
        out = side_workload(name, S, F, bitrate, kind, steps, warmup, no_gain=no_gain, prior=prior)
        out["fresh_process"] = False
        out["fresh_process_error"] = repr(ex)
        return out


HOST_PIPELINE_DEPTH = {False: 2, True: 4}   # float32 / 16-bit samples: calls in flight (mirrors TAtrac3EncoderBatch::EncodePipelined)


def host_pipeline_workload(S, F, steps=400, warmup=24, s16=False, depth=0):
    """configs[1] fed from HOST memory the way the reference's caller hands it over (pcmengin.h:152-192), PCIe included: two
    page-locked PCM buffers and two frame buffers alternate, the calls are asynchronous, so the H2D copy of call k + 1, the
    kernels of call k and the D2H copy of call k - 1 overlap (at3hip_host_alloc / at3hip_wait_*). Never part of `value`."""
    out = {"workload": f"configs[1] from host memory: {S} x {F} frames per call, pinned buffers, {depth or HOST_PIPELINE_DEPTH[bool(s16)]} calls in flight, H2D + kernels + D2H overlapped"
                       + (", 16-bit samples (at3hip_encode_s16: converted on the device)" if s16 else ", float32 samples")}
    try:
        import torch
        import atracdenc_amd
        enc = atracdenc_amd.At3Hip(n_streams=S, max_blocks=F + 1, bitrate=LP2, device_id=0)
        # calls in flight (at3hip_wait_* reach three calls back; buffers alternate D ways): what TAtrac3EncoderBatch::EncodePipelined uses
        # for the sample format (kDepthOf in at3hip_host.hpp) unless the caller asks for another depth
        D = depth or HOST_PIPELINE_DEPTH[bool(s16)]
        ins = [enc.host_alloc((S, F, 1024, 2), np.int16 if s16 else np.float32) for _ in range(D)]
        outs = [enc.host_alloc((S, F, enc.frame_size), np.uint8) for _ in range(D)]
        rng = np.random.RandomState(5)
        for a in ins:
            a[...] = rng.randint(-8192, 8192, size=a.shape).astype(np.int16) if s16 else rng.randint(-8192, 8192, size=a.shape).astype(np.float32) / np.float32(32768.0)
        prime = synth_pcm(S, 1, 99)
        enc.encode(prime)                      # LOOK_AHEAD call
        for i in range(warmup):
            enc.encode_host_async(ins[i % D], outs[i % D])
        enc.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            if i >= D:
                enc.wait_input(D - 1)      # where the caller refills ins[i % D]
            enc.encode_host_async(ins[i % D], outs[i % D])
            if i >= D - 1:
                enc.wait_frames(D - 1)     # where the caller consumes the oldest call's frames: at most D calls in flight
        enc.sync()
        dt = (time.perf_counter() - t0) / steps
        # the same calls one at a time (copy, kernels, copy back, then the next): what the overlap buys
        t0 = time.perf_counter()
        for i in range(8):
            enc.encode_host_async(ins[i % D], outs[i % D])
            enc.sync()
        dt_serial = (time.perf_counter() - t0) / 8
        # the PCIe bound measured here: the same 32 MiB from page-locked memory, copies only
        nbytes = ins[0].nbytes
        x = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        d.copy_(x, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            d.copy_(x, non_blocking=True)
        torch.cuda.synchronize()
        h2d = (time.perf_counter() - t0) / 10
        for a in ins + outs:
            enc.host_free(a)
        enc.close()
        out.update({"value": round(S * F / dt, 1), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 4),
                    "serial_calls_ms_per_step": round(dt_serial * 1e3, 4),
                    "h2d_bytes_per_step": nbytes, "d2h_bytes_per_step": outs[0].nbytes,
                    "h2d_copy_alone_ms": round(h2d * 1e3, 4), "h2d_GBps": round(nbytes / h2d / 1e9, 2),
                    "pcie_bound_frames_per_s": round(S * F / h2d, 1), "frac_of_pcie_bound": round(h2d / dt, 4)})
    except Exception as ex:   # noqa: BLE001 - the headline line must not depend on this
        out["error"] = repr(ex)
    return out


def widened_rows(S):
    """SURVEY 8(f) rows f3 / f4 measured in the same process, after the headline (not part of `value`): the ATRAC1 encode
    path and ATRAC3plus PCM-to-frames (no tonal block) on the same audio shape (S stereo streams x 65536 samples, PCM resident in HBM)."""
    out = {}
    try:
        import torch
        import atracdenc_amd
        rng = np.random.RandomState(3)
        pcm = torch.from_numpy((rng.randint(-8192, 8192, size=(S, 65536, 2)).astype(np.float32) / np.float32(32768.0))).cuda()

        e1 = atracdenc_amd.At1Hip(n_streams=S, max_blocks=128)
        o1 = torch.zeros((S, 128, 2, 212), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()   # (the libraries' streams do not wait for torch's)
        for _ in range(2):
            e1.encode_device(pcm.data_ptr(), 128, o1.data_ptr())
        t0 = time.perf_counter()
        for _ in range(10):   # queued calls, one wait (AT3HIP_ASYNC), like the ATRAC3plus row below
            e1.encode_device(pcm.data_ptr(), 128, o1.data_ptr(), asynchronous=True)
        e1.sync()
        dt = (time.perf_counter() - t0) / 10
        e1.encode_device(pcm.data_ptr(), 128, o1.data_ptr())   # (a synchronous call: the one that carries stage timings)
        out["atrac1_encode"] = {"value": round(S * 128 / dt, 1), "unit": "512-sample stereo sound-unit pairs/s", "ms_per_step": round(dt * 1e3, 4),
                                "x_realtime": round(S * 65536 / 44100.0 / dt, 1), "device_ms": {k: round(v, 4) for k, v in e1.timings().items()}}
        e1.close()
        ep = atracdenc_amd.At3pHip(n_streams=S, max_frames=32)
        op = torch.zeros((S, 32, 2048), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            ep.encode_frames_device(pcm.data_ptr(), 32, op.data_ptr())
        t0 = time.perf_counter()
        for _ in range(10):   # queued calls, one wait: the writer of a call overlaps the next call's filter bank and transform
            ep.encode_frames_device(pcm.data_ptr(), 32, op.data_ptr(), asynchronous=True)
        ep.sync()
        dt = (time.perf_counter() - t0) / 10
        ep.encode_frames_device(pcm.data_ptr(), 32, op.data_ptr())
        tm = ep.timings()
        out["atrac3plus_encode_no_tonal"] = {"value": round(S * 32 / dt, 1), "unit": "2048-sample stereo frames/s", "ms_per_step": round(dt * 1e3, 4),
                                             "x_realtime": round(S * 65536 / 44100.0 / dt, 1), "device_ms": {k: round(v, 4) for k, v in tm.items()},
                                             "frontend_algorithmic_GBps": round(S * 32 * 2 * 2048 * 16 / ((tm["pqf_ms"] + tm["mdct_ms"]) * 1e-3) / 1e9, 1)}
        ep.close()
    except Exception as ex:   # the headline line must not depend on these
        out["error"] = repr(ex)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: 64 at N=1, 1024 at N>1)")
    ap.add_argument("--frames", type=int, default=0, help="frames per stream per step (default: 64 at N=1, 128 at N>1)")
    ap.add_argument("--bitrate", type=int, default=LP2)
    ap.add_argument("--input", choices=["noise", "burst", "tones"], default="noise")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-workloads", action="store_true")
    ap.add_argument("--no-gain", action="store_true")
    ap.add_argument("--regions", type=int, default=10, help="further timed regions after the contract's K-step one (N = 1): each at least "
                                                             "--region-ms long, value = the median over all of them")
    ap.add_argument("--region-ms", type=float, default=50.0)
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the frames after the timed regions")
    ap.add_argument("--runs", type=int, default=0, help="TUNING AID: AT3HIP_OPT_RUNS (runs per stream and channel of the QMF / MDCT kernels)")
    ap.add_argument("--gain-form", type=int, default=0, help="A/B AID: AT3HIP_OPT_GAIN_FORM (0 = the two-wavefront workgroups of the upsampler / "
                                                             "AnalyzeGain kernel, 1 = one wavefront per item; same results)")
    ap.add_argument("--timing-every", type=int, default=8, help="AT3HIP_OPT_TIMING_EVERY: every Nth step carries the stage-timing HIP events the "
                                                                 "roofline figure is read from (recording them on every step costs the step 3 %%)")
    ap.add_argument("--gain-wgs", type=int, default=0, help="TUNING AID: AT3HIP_OPT_GAIN_WGS_PER_CU")
    ap.add_argument("--sync-steps", action="store_true", help="PROFILING AID: run the timed steps synchronously (no overlap of "
                                                              "consecutive calls) so that rocprofv3 sees every kernel alone; "
                                                              "the line is marked and is not a valid throughput result")
    ap.add_argument("--device-map", default="", help="TEST AID: comma list of device ordinals to use instead of 0..N-1 (e.g. 0,0 runs the "
                                                     "two-device code path on one GPU); recorded in the JSON line, never a valid N-GPU result")
    ap.add_argument("--one-side-workload", default="", help="INTERNAL: JSON [name, S, F, bitrate, kind, steps, warmup, no_gain] - measure that one side workload in this "
                    "(fresh) process and print its record (what `other_workloads` calls for each entry)")
    args = ap.parse_args()
    if args.one_side_workload:
        name, S, F, br, kind, steps, warmup, no_gain, prior = (json.loads(args.one_side_workload) + [0])[:9]
        print(json.dumps(side_workload(name, S, F, br, kind, steps, warmup, no_gain=bool(no_gain), prior=int(prior))))
        return

    import torch

    DeviceJob.sync_steps = args.sync_steps
    DeviceJob.timing_every = max(0, args.timing_every)
    if args.sync_steps or (args.regions == 0 and args.steps < 8 * max(1, args.timing_every)):
        DeviceJob.timing_every = 1   # a run too short to sample (profiling passes): every step carries its events
    DeviceJob.runs = args.runs
    DeviceJob.gain_form = int(args.gain_form)
    DeviceJob.gain_wgs = args.gain_wgs
    from atracdenc_amd import dist as at3dist
    rank, local_rank, world = at3dist.env_world()
    if world > 1 and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU or none at all")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    n_gpus = args.gpus
    n_visible = torch.cuda.device_count()
    dist = None
    if world > 1:
        # one rank per GPU (torch.distributed.run): CPU-side gloo for the barrier and the MAX only - the data path has no
        # exchange step, and north_star asks for no RCCL
        dev = local_rank
        mode = f"{world} ranks x 1 GPU (torch.distributed.run, gloo barrier)"
        if args.device_map:
            dmap = [int(x) for x in args.device_map.split(",")]
            if len(dmap) != world:
                raise SystemExit("--device-map needs one entry per rank")
            dev = dmap[local_rank]
            mode = f"TEST AID --device-map {args.device_map}: {world} ranks on {len(set(dmap))} physical GPU(s) - not an N-GPU measurement"
        if dev >= n_visible:
            raise SystemExit(f"rank {rank}: device {dev} but only {n_visible} GPU(s) visible")
        dist = at3dist.init("gloo")
        devices = [dev]
    else:
        if args.device_map:
            devices = [int(x) for x in args.device_map.split(",")]
            if len(devices) != n_gpus or max(devices) >= n_visible:
                raise SystemExit("--device-map needs --gpus entries, all visible")
            mode = f"TEST AID --device-map {args.device_map}: {n_gpus} contexts on {len(set(devices))} physical GPU(s) - not an N-GPU measurement"
        else:
            if n_visible < n_gpus:
                raise SystemExit(f"--gpus {n_gpus} requested but only {n_visible} GPU(s) are visible: refusing to measure fewer")
            devices = list(range(n_gpus))
            mode = f"1 process x {n_gpus} GPU(s), one host thread and one at3hip context per device"
    torch.cuda.set_device(devices[0])

    S = args.streams or (64 if n_gpus == 1 else 1024)
    F = args.frames or (64 if n_gpus == 1 else 128)
    jobs = [DeviceJob(d, S, F, args.bitrate, args.no_gain, args.input, seed=1 + rank * 64 + i) for i, d in enumerate(devices)]
    fsz = jobs[0].fsz
    for j in jobs:
        j.warmup(args.warmup)

    one_gpu_ref = None
    if world == 1 and n_gpus > 1:
        # like-for-like reference for the scaling figure: the same per-GPU shard on device 0 ALONE, same step count
        timed_region(jobs[:1], min(args.steps, 5), None)
        dt1 = timed_region(jobs[:1], args.steps, None)
        one_gpu_ref = {"value": round(S * F * args.steps / dt1, 1), "unit": "frames/s", "ms_per_step": round(dt1 / args.steps * 1e3, 4),
                       "note": "device 0 alone on the same per-GPU shard, measured in this run before the N-GPU timed region"}

    _dbg_loud(jobs[0], "warm-up")
    # The contract's timed region: EXACTLY --steps steps between barrier + synchronize pairs, max over ranks.
    elapsed = timed_region(jobs, args.steps, dist)
    _dbg_loud(jobs[0], "contract region")
    elapsed = at3dist.max_over_ranks(elapsed, dist, device="cpu")
    j0 = jobs[0]
    try:
        sclk_contract = j0.enc.sclk_mhz()   # the shader clock under the rate loop of the contract region's last step
    except Exception:   # noqa: BLE001 - diagnostic only
        sclk_contract = None
    # median of >= 10 runs. Further regions bracketed the same way, each long enough (>= --region-ms) that a
    # launch hiccup or a clock ramp does not decide the figure; the steps they run are the same steps.
    region_ms = [elapsed / args.steps * 1e3]
    region_steps = args.steps
    k1_ms, stage_sum = [], {}

    def collect_k1(n_calls):
        # the timed steps of the region that just ended (the ctx keeps the events of its last 32 calls)
        ms, stage = j0.k1_stats(min(n_calls, 32), 0)
        k1_ms.extend(ms)
        for k, v in stage.items():
            stage_sum[k] = stage_sum.get(k, 0.0) + v * len(ms)
    collect_k1(args.steps)
    if args.regions > 0 and not args.sync_steps:
        region_steps = max(args.steps, int(np.ceil(args.region_ms / max(region_ms[0], 1e-6))))
        for _ in range(args.regions):
            dt = timed_region(jobs, region_steps, dist)
            dt = at3dist.max_over_ranks(dt, dist, device="cpu")
            region_ms.append(dt / region_steps * 1e3)
            _dbg_loud(jobs[0], "region")
            collect_k1(region_steps)
    # every step device 0's job has run since its LOOK_AHEAD call - warm-up, the one_gpu_same_workload regions of a
    # one-process multi-GPU run, the timed regions - is what the replay has to repeat to land on the same batch parity
    steps_done = j0.calls
    checksum = j0.checksum()
    try:
        sclk_mhz = j0.enc.sclk_mhz()
    except Exception:   # noqa: BLE001 - diagnostic only
        sclk_mhz = None
    stage_ms = {k: v / max(1, len(k1_ms)) for k, v in stage_sum.items()}   # the timed steps of all the regions
    iso_ms = j0.isolated_k1()                       # 5 synchronous steps after the timed regions
    _dbg_loud(j0, "isolated steps")
    parity = None
    contexts = []
    if not args.no_parity and not args.sync_steps:
        # EVERY context of the run (one per device; on every rank): (1) the whole timed sequence again from start of stream - same
        # final frames (determinism); (2) the start of that sequence against the CPU oracle. Both after the timing, on the buffers the
        # timed regions used. Seeds differ per context, so the checksums must all differ (a context fed another one's shard shows).
        for i, j in enumerate(jobs):
            rep = {"rank": rank, "device": j.device, "seed": 1 + rank * 64 + i, "steps": j.calls, "checksum": j.checksum()}
            same = None
            if j.calls * S * F <= 64 * 1024 * 1024:
                n_calls = j.calls
                timed_out = j.d_out.clone()
                dbg_taps = None
                if os.environ.get("BENCH_DEBUG_TAPS"):
                    from atracdenc_amd import binding as B
                    nbk = j.F + 1
                    sp = j.enc.read_tap(B.TAP_SPECTRA, np.float32, (j.S, j.F, 2, 1024))
                    ge = j.enc.read_tap(B.TAP_ENERGY_SCALE, np.float32, (j.S, nbk, 2, 4))
                    lo = j.enc.read_tap(B.TAP_LOUDNESS, np.float32, (j.S, j.F))
                    ps = j.enc.read_tap(B.TAP_PSY, B.At3Hip.PSY_DTYPE, (j.S, j.F, 2))
                    dbg_taps = {"spec_absmax": float(np.abs(sp).max()), "spec_nonzero": int((sp != 0).sum()), "ges_min": float(np.nanmin(ge)), "ges_max": float(np.nanmax(ge)),
                                "ges_nan": int(np.isnan(ge).sum()), "loud_min": float(np.nanmin(lo)), "loud_max": float(np.nanmax(lo)), "loud_nan": int(np.isnan(lo).sum()),
                                "loud_ch_max": float(np.nanmax(ps["loud_ch"])), "energy_max": float(np.nanmax(ps["energy"])), "sfi_max": int(ps["sfi"].max())}
                    cv = j.enc.read_tap(B.TAP_CURVES, np.uint8, (j.S, nbk, 2, 4, 16))
                    dbg_taps["curve_points_total"] = int(cv[..., 0].sum()); dbg_taps["ges_not_one"] = int((ge[:, 1:] != 1.0).sum())
                    dbg_taps["loud_first_stream"] = [float(x) for x in lo[0, :3]]
                    if os.environ.get("BENCH_DEBUG_TAPS") == "2":
                        sys.stderr.write("taps rank %d: %r\n" % (rank, dbg_taps))
                again = j.replay(n_calls)
                same = (again == rep["checksum"])
                if not same:   # which of the two runs is the odd one out: the sequence a third time, and where the frames differ
                    diff = (timed_out != j.d_out).any(dim=2)                     # [S, F]
                    third = j.replay(n_calls)
                    rep["replay_diagnostic"] = {"steps": n_calls, "timed": rep["checksum"], "replay": again, "replay_again": third,
                                                "frames_differing": int(diff.sum().item()), "streams_differing": int(diff.any(dim=1).sum().item()),
                                                "first": diff.nonzero()[:6].tolist(), "timed_nonzero_bytes": int((timed_out != 0).sum().item()),
                                                "replay_nonzero_bytes": int((j.d_out != 0).sum().item()),
                                                "taps_of_the_timed_run": dbg_taps,
                                                "timed_frame0_head": timed_out[diff.nonzero()[0][0], diff.nonzero()[0][1], :24].tolist() if diff.any() else None}
                    sys.stderr.write("replay mismatch on rank %d: %r\n" % (rank, rep["replay_diagnostic"]))
            try:
                chk = j.parity_sample(n_check=4 if n_gpus == 1 else 2)
                chk["timed_sequence_replayed_identically"] = same
            except Exception as ex:   # noqa: BLE001 - reported in the line, parity_in_run stays false
                chk = {"error": repr(ex)}
            rep["parity_check"] = chk
            rep["ok"] = bool(chk.get("mismatching_frames", 1) == 0 and same is not False)
            contexts.append(rep)
        contexts = [c for part in at3dist.gather_objects(contexts, dist) for c in part]
        if rank == 0:
            parity = contexts[0]["parity_check"]

    if rank == 0:
        med_ms = float(np.median(region_ms))
        value = n_gpus * S * F / (med_ms * 1e-3)
        k1_avg_ms = float(np.mean(k1_ms)) if k1_ms else iso_ms   # (no timed step in the regions: --timing-every 0)
        achieved, frac = roofline_of(k1_avg_ms, S * F)
        ach_iso, frac_iso = roofline_of(iso_ms, S * F)
        traffic, traffic_note = None, "not measured"
        prof = os.path.join(ROOT, "profiles", "k1_traffic.json")
        if os.path.exists(prof) and (S, F) == (64, 64):
            try:
                tj = json.load(open(prof))
                traffic = tj.get("bytes_per_launch")
                if tj.get("kernel_source_sha16") != kernel_source_sha16():
                    traffic_note = "STALE (profiles/k1_traffic.json was collected on other kernel sources than this tree's): "
                else:
                    traffic_note = ""
                traffic_note += ("read from profiles/k1_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel on "
                                "this workload, collected separately as the guide prescribes) - not measured in this run")
            except Exception:
                traffic = None
        # whole-pipeline HBM traffic and the vector-instruction counts of the QMF + MDCT kernels: PMC passes cannot run inside
        # the timed region, so these come from the committed profile of the same workload (tools/profile_gpu.sh) and say so
        pipe_traffic, valu_floor_ms, valu_note, pipe_valu = None, None, "not available", None
        pt = os.path.join(ROOT, "profiles", "pipeline_traffic.json")
        if os.path.exists(pt) and (S, F) == (64, 64) and not args.no_gain and fsz == 384:
            try:
                pj = json.load(open(pt))
                profile_stale = pj.get("kernel_source_sha16") != kernel_source_sha16()
                pipe_traffic = {"profile_is_of_these_kernel_sources": not profile_stale, "pipeline_bytes_per_frame": round(pj["pipeline_bytes_per_frame"], 1),
                                "algorithmic_bytes_per_frame": pj["algorithmic_bytes_per_frame"],
                                "source": "profiles/pipeline_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2) + WRITE_SIZE summed over the pipeline's "
                                          "kernels, separate passes, this workload - not measured in this run"}
                # Issue floors. SQ_INSTS_VALU per kernel (profiles/pipeline_traffic.json) priced with the issue costs tools/ubench/valu_issue
                # measured with four and eight wavefronts PROVABLY on every SIMD (profiles/r05_ubench_valu_issue.txt): a 32-bit encoded
                # vector instruction (VOP1 / VOP2 / VOPC) holds a SIMD for 2.15 cycles, a 64-bit encoded one (VOP3, the packed fp32 forms,
                # DPP / SDWA) for 4.2; a kernel's share of the second class comes from the compiler's assembly (profiles/valu_mix.json)
                mix, cyc32, cyc64, ghz = {}, 2.15, 4.2, 2.4
                mp = os.path.join(ROOT, "profiles", "valu_mix.json")
                if os.path.exists(mp):
                    mj = json.load(open(mp))
                    mix = {k: v.get("wide_share", v.get("packed_share", 0.0)) for k, v in mj.get("kernels", {}).items()}
                    cj = mj.get("cycles_per_wave_instruction_per_simd", {})
                    cyc32, cyc64, ghz = cj.get("encoded_32_bit", cyc32), cj.get("encoded_64_bit", cyc64), cj.get("clock_ghz", ghz)
                n_simd = 1024

                def floor_ms(counts):
                    return sum(n * (mix.get(k, 0.0) * cyc64 + (1.0 - mix.get(k, 0.0)) * cyc32) / ghz * 1e-6 for k, n in counts.items()) / n_simd

                k1_counts = pj.get("valu_wave_insts_per_launch", {})
                if k1_counts:
                    valu_floor_ms = floor_ms(k1_counts)
                    valu_note = (f"{int(sum(k1_counts.values()))} vector wave-instructions per launch pair (SQ_INSTS_VALU, profiles/pipeline_traffic.json) / "
                                 f"{n_simd} SIMDs, a 32-bit encoded instruction priced at {cyc32} cycles of its SIMD and a 64-bit encoded one (VOP3, packed fp32, DPP) at {cyc64} "
                                 f"(tools/ubench/valu_issue with 4 and 8 wavefronts per SIMD by construction, profiles/r05_ubench_valu_issue.txt), at {ghz} GHz; the share of 64-bit "
                                 "encodings per kernel from the compiler's assembly (profiles/valu_mix.json); the arithmetic contract forbids FMA, so the "
                                 "multiply-add pairs of the FIR are two packed instructions each")
                all_counts = pj.get("valu_wave_insts_per_launch_all", {})
                if all_counts:
                    step_floor = floor_ms(all_counts)
                    pipe_valu = {"floor_ms_per_step": round(step_floor, 4), "frac": round(step_floor / med_ms, 4),
                                 "vector_wave_instructions_per_step": int(sum(all_counts.values())),
                                 "note": "the whole step against the issue floor of its own vector instruction streams (every kernel of the pipeline, "
                                         "same pricing as roofline.valu_floor_ms; f64 instructions priced like the others, so the true floor is "
                                         "slightly higher): frac = floor / ms_per_step = the share of the step during which the vector ALUs would be busy "
                                         "if nothing else ever stalled a wavefront. A wavefront also issues at most one instruction of ANY kind every ~6.4 "
                                         "cycles (same table), and the scalar unit of a CU one instruction per cycle for its four SIMDs"}
            except Exception:
                pipe_traffic = None
        # the same launches as rocprofv3 saw them alone (kernel begin to kernel end, without the event and launch gaps of the
        # HIP-event span above): read from the committed profile of this workload, and labelled as such
        profile_isolated = {}
        if pipe_traffic is not None and getattr(j0, "k1_launches", 1) == 2:
            try:
                import glob
                import re
                prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_isolated_rocprof_summary.txt")))[-1]
                us = {}
                for ln in open(prof):
                    m = re.match(r"^(k_qmf_sub8|k_mdct_sub<false(?:, \d)?>)\s+\d+\s+[\d.]+\s+([\d.]+)\s", ln)
                    if m and m.group(1) not in us:
                        us[m.group(1)] = float(m.group(2))
                if len(us) == 2:
                    pms = sum(us.values()) * 1e-3
                    pa, pf = roofline_of(pms, S * F)
                    profile_isolated = {"rocprofv3_avg_launch_ms": round(pms, 5), "rocprofv3_achieved": pa, "rocprofv3_frac": pf,
                                        "rocprofv3_source": os.path.relpath(prof, ROOT) + " (kernel durations of synchronous steps of this "
                                                            "workload; not measured in this run)"}
            except Exception:
                profile_isolated = {}
        cfgname = {384: "LP2 132 kbps", 192: "LP4 66 kbps joint stereo"}.get(fsz, f"{fsz} B/frame")
        line = {
            "metric": "ATRAC3 1024-sample stereo frames/sec", "value": round(value, 1), "unit": "frames/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(med_ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_is": "median of the timed regions (SURVEY 8(d)); the contract's exactly --steps region alone is timing.value_contract_region",
            "x_realtime": round(value * 1024 / 44100.0, 1),
            "config": {"workload": f"ATRAC3 {cfgname} stereo, {S} streams x {F} frames = {S * F} frames per GPU per step "
                                   f"({'BASELINE configs[1]' if (S, F) == (64, 64) else 'per-GPU shard of BASELINE configs[2]' if (S, F) == (1024, 128) else 'custom'}), "
                                   f"'{args.input}' PCM resident in HBM, frames written to HBM (PCIe excluded)",
                       "streams_per_gpu": S, "frames_per_stream_per_step": F, "frames_per_step_all_gpus": n_gpus * S * F,
                       "frame_bytes": fsz, "input": args.input,
                       "gain_control": not args.no_gain, "tonal_components": True, "parallelism": f"streams/{n_gpus}",
                       "launch": mode, "collectives": "none (no RCCL); start/stop barrier only"},
            "per_gpu_value": round(value / n_gpus, 1),
            "roofline": {"bound": "hbm",
                         "kernel": ("k_qmf_sub8 + k_mdct_sub: the batched QMF tree and the windowed MDCT-512 as the two kernels on either side of "
                                    "the gain analysis (which needs every block's subbands before any curve exists); durations summed")
                         if getattr(j0, "k1_launches", 1) == 2 else "k_qmf_mdct8 (fused QMF + windowed MDCT-512)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": frac, "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME_K1 * S * F,
                         "avg_launch_ms": round(k1_avg_ms, 5), "launches_per_step": getattr(j0, "k1_launches", 1),
                         "launches_timed": len(k1_ms), "timing_every": DeviceJob.timing_every,
                         "timing": "HIP events on the context's stream around the kernel(s) of every timing_every-th step of the timed regions "
                                   "(AT3HIP_OPT_TIMING_EVERY: recorded on every step the events cost the step 3 %); they include the launch gaps "
                                   "that rocprofv3's kernel durations (profiles/) do not",
                         "limiter": "the vector pipe (FMA-free fp32 arithmetic contract: the FIR's multiplies and adds are two packed instructions "
                                    "each), not HBM - and for the fused kernel running back to back on white noise the board's 1400 W power cap, which holds "
                                    "the shader clock at 2.0 - 2.1 GHz instead of 2.4 (profiles/r06_k1_power_probe.txt, not measured in this run): see "
                                    "DESIGN.md section 5; the HBM fraction is reported because it is the roofline north_star names",
                         "note": "launches of the timed region (device 0). In the timed region the two kernels run on two of the context's "
                                 "three streams and share the GPU with the neighbouring steps' gain analysis and rate loop - on purpose: "
                                 "that overlap is what shortens the step - so each launch takes longer than it does alone; `isolated` is "
                                 "the same launches with the GPU to themselves",
                         "flops": {"algorithmic_flops_per_frame": ALGO_FLOPS_PER_FRAME_K1, "peak_TF": round(FP32_NO_FMA_PEAK_TF, 2),
                                   "peak_is": "fp32 vector peak 157.3 TFLOP/s / 2: the arithmetic contract forbids FMA contraction, so a multiply-add is two "
                                              "instructions (SURVEY 8(d)); spec-derived, no micro-benchmark involved",
                                   "achieved_TF": flops_of(k1_avg_ms, S * F)[0], "frac": flops_of(k1_avg_ms, S * F)[1],
                                   "isolated_achieved_TF": flops_of(iso_ms, S * F)[0], "isolated_frac": flops_of(iso_ms, S * F)[1],
                                   "hbm_frac_at_this_ceiling": round(FP32_NO_FMA_PEAK_TF * 1e12 / ALGO_FLOPS_PER_FRAME_K1 * ALGO_BYTES_PER_FRAME_K1 / 1e9 / HBM_PEAK_GBS, 4),
                                   "note": "the compute-side view of the same launches (HIP-event durations as for `achieved`): at 100 % of this peak the QMF + "
                                           "MDCT work would run at hbm_frac_at_this_ceiling of the HBM peak, so north_star's 50 % of HBM asks for 78 % of "
                                           "the FMA-free vector peak"},
                         "sclk_mhz_observed": None if sclk_mhz is None else round(sclk_mhz, 1),
                         "sclk_note": "shader clock under the rate loop of the last timed step: s_memtime cycles / s_memrealtime (100 MHz) ticks over the "
                                      "life of k_alloc_pack's workgroup 0 (AT3HIP_TAP_CLOCK)",
                         "valu_floor_ms": None if valu_floor_ms is None else round(valu_floor_ms, 5),
                         "valu_frac": None if valu_floor_ms is None else round(valu_floor_ms / max(iso_ms, 1e-6), 4),
                         "valu_floor_note": valu_note + "; valu_frac = valu_floor_ms / isolated.avg_launch_ms",
                         "isolated": dict({"avg_launch_ms": round(iso_ms, 5), "achieved": ach_iso, "frac": frac_iso,
                                           "note": "same kernel, same batch, launched alone (5 synchronous steps after the timed region)"},
                                          **profile_isolated)},
            "pipeline_traffic": pipe_traffic,
            "pipeline_valu": pipe_valu,
            "stage_ms_per_step": {k: round(v, 4) for k, v in sorted(stage_ms.items())},
            "pipelining": "three HIP streams inside the context: the heavy front stage (QMF, gain spectra, envelopes) of step i+1, the "
                          "light front stage (curves, energy scales, MDCT) of step i and the back half (psychoacoustics, rate loop, "
                          "packing) of step i overlap; stage_ms are per-step HIP-event spans and overlap in time, total_ms is one step's latency",
            "timing": {"value_is": "median over the timed regions (SURVEY 8(d)): region 0 is the contract's exactly --steps steps, the others "
                                   f"are {region_steps} steps each (>= {args.region_ms:g} ms); every region sits between barrier + synchronize pairs, max over ranks",
                       "regions": len(region_ms), "steps_per_region": [args.steps] + [region_steps] * (len(region_ms) - 1),
                       "ms_per_step_median": round(med_ms, 4), "ms_per_step_min": round(min(region_ms), 4), "ms_per_step_max": round(max(region_ms), 4),
                       "ms_per_step_contract_region": round(region_ms[0], 4),
                       "sclk_mhz_contract_region": None if sclk_contract is None else round(sclk_contract, 1),
                       "contract_region_note": "every timed region starts with an empty pipeline and ends with a drain (barrier + synchronize on both sides): "
                                               "about one step's time is paid once per region on top of its steps (30-step regions read 0.258 ms per step where 182-step "
                                               "ones read 0.249); region 0 also starts --warmup steps after an idle device, while the part is still stepping its clock up "
                                               "(sclk_mhz_contract_region against roofline.sclk_mhz_observed, taken after the last region) - with the defaults (50 warm-up "
                                               "steps, 200 timed) neither matters",
                       "value_contract_region": round(n_gpus * S * F / (region_ms[0] * 1e-3), 1),
                       "lone_call": getattr(j0, "lone_call", None)},
            "checksum": checksum,
        }
        if parity is not None:
            sums = [c["checksum"] for c in contexts]
            line["parity_in_run"] = bool(len(contexts) == n_gpus and all(c["ok"] for c in contexts) and len(set(sums)) == len(sums))
            line["parity_check"] = parity
            if n_gpus > 1:
                line["contexts"] = contexts   # one record per device / rank: seed, checksum, replay and oracle check of ITS shard
        if args.sync_steps:
            line["INVALID_profiling_run"] = "--sync-steps: calls were not pipelined; not a throughput measurement"
        if one_gpu_ref is not None:
            line["one_gpu_same_workload"] = one_gpu_ref
        for j in jobs:
            j.close()
        jobs = []
        torch.cuda.empty_cache()
        if n_gpus == 1 and world == 1:
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline()
            if not args.no_side_workloads:
                line["other_workloads"] = [
                    side_workload_fresh("configs[1] shape, 'burst' input (gain-control path busy)", 64, 64, LP2, "burst", 30, 3),
                    side_workload_fresh("configs[1] shape, 'tones' input (tonal extraction busy)", 64, 64, LP2, "tones", 30, 3),
                    side_workload_fresh("configs[1] shape, 'tones', as the THIRD context of its process (DESIGN section 7: open)", 64, 64, LP2, "tones", 30, 3, prior=2),
                    side_workload_fresh("configs[3]: LP4 66 kbps joint stereo, configs[1] shape, 'noise'", 64, 64, LP4, "noise", 30, 3),
                    side_workload_fresh("shard_1024x128: per-GPU shard of configs[2] (1024 streams x 128 frames) on one GPU, 'noise'", 1024, 128, LP2, "noise", 10, 2),
                    side_workload_fresh("configs[1] shape, 'noise', --nogaincontrol: the FUSED QMF + MDCT kernel k_qmf_mdct8 (north_star's kernel)", 64, 64, LP2, "noise", 30, 3, no_gain=True),
                    side_workload_fresh("shard_1024x128, 'noise', --nogaincontrol: k_qmf_mdct8 at the per-GPU shard", 1024, 128, LP2, "noise", 10, 2, no_gain=True),
                ]
                line["host_pipeline"] = host_pipeline_workload(64, 64)
                line["host_pipeline_s16"] = host_pipeline_workload(64, 64, s16=True)
                line["widened_rows"] = widened_rows(64)
        print(json.dumps(line))
    for j in jobs:
        j.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
