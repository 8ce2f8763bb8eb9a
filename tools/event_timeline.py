"""GPU box, a -DAT3HIP_DEBUG_EVENTS build (AT3HIP_LIB: the release kernels plus at3hip_debug_event_ms): the pipelined step's stage timeline from the context's own HIP events, no tracer.
For the last few calls of a run of asynchronous steps: when each stage began / ended relative to the oldest call's first event.
usage: AT3HIP_LIB=build_ab/lib_dbg.so python tools/event_timeline.py [bench-like args: --input tones --bitrate 66150]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/..")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../tests")
import numpy as np
import bench
from atracdenc_amd import binding as B

args = sys.argv[1:]
kind = args[args.index("--input") + 1] if "--input" in args else "noise"
br = int(args[args.index("--bitrate") + 1]) if "--bitrate" in args else bench.LP2
bench.DeviceJob.timing_every = 1   # (every call carries its events here)
for _ in range(int(args[args.index("--prior") + 1]) if "--prior" in args else 0):   # contexts created, run and closed before the measured one
    j0 = bench.DeviceJob(0, 64, 64, br, False, kind, seed=1)
    j0.warmup(3); j0.run_steps(20); j0.close()
job = bench.DeviceJob(0, 64, 64, br, False, kind, seed=1)
job.warmup(3)
job.run_steps(40)
lib = job.enc.lib
lib.at3hip_debug_event_ms.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_float)]
names = ["QMF start", "QMF end", "analysis end", "curves+ges end", "MDCT end", "back start", "psy end", "rate loop end"]
N = 6
def at(ago, i):
    ms = ctypes.c_float()
    rc = lib.at3hip_debug_event_ms(job.enc.ctx, N - 1, 0, ago, i, ctypes.byref(ms))
    assert rc == 0, rc
    return ms.value * 1e3
print(f"input={kind} bitrate={br}: microseconds after the first event of the call {N - 1} calls back")
rows = []
for ago in range(N - 1, -1, -1):
    t = [at(ago, i) for i in range(8)]
    rows.append(t)
    print(f"call -{ago}: " + "  ".join(f"{names[i]} {t[i]:8.1f}" for i in range(8)))
per = np.diff([r[7] for r in rows])
print("period (rate loop end to rate loop end): %s us" % np.round(per, 1))
for k in range(1, N):
    a, b = rows[k - 1], rows[k]
    print(f"call -{N - 1 - k}: rate loop starts ~{b[6]:.1f} (psy end) after the previous rate loop's end {a[7]:.1f} (+{b[6] - a[7]:.1f}); its analysis ended {b[2]:.1f}; "
          f"the NEXT call's analysis end vs this rate loop's start: {rows[k + 1][2] - b[6] if k + 1 < N else float('nan'):.1f}")
