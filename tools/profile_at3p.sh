#!/bin/bash
# Run on the GPU box: per-kernel durations of the ATRAC3plus path (rocprofv3 --kernel-trace --stats over tools/at3p_bench.py), the
# kernels' resources and the bench line of the same command without the profiler -> gpurun_out/at3p_summary.txt
REPO=$(pwd)
export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out
cd /tmp
rm -rf /tmp/at3pprof
rocprofv3 --kernel-trace --stats -d /tmp/at3pprof -o at3p -- python $REPO/tools/at3p_bench.py --steps 20 --warmup 3 > /dev/null 2>&1
python3 - > $REPO/gpurun_out/at3p_summary.txt <<'PY'
import glob, sqlite3
print("== ATRAC3plus PCM-to-frames path without the tonal block (SURVEY 8(f) row f4): rocprofv3 --kernel-trace --stats -- python tools/at3p_bench.py")
print("   workload: 64 stereo streams x 32 frames of 2048 samples (the audio of BASELINE configs[1]), PCM and frames in HBM")
print("%-48s %6s %12s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "pct"))
for f in glob.glob("/tmp/at3pprof/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start) from kernels group by name order by sum(end-start) desc"))
    tot = sum(r[2] for r in rows)
    for name, calls, total, avg, mn in rows:
        print("%-48s %6d %12.1f %10.2f %10.2f %6.2f" % (name.split("(")[0][:48], calls, total / 1e3, avg / 1e3, mn / 1e3, 100.0 * total / tot))
    print("-- resources --")
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    want = [c for c in ("workgroup_size", "grid_size", "lds_size", "scratch_size", "vgpr_count", "sgpr_count") if c in cols]
    if want:
        for r in db.execute("select name, %s from kernels group by name" % ", ".join("max(%s)" % c for c in want)):
            print("%-48s %s" % (r[0].split("(")[0][:48], " ".join("%s=%s" % (c, v) for c, v in zip(want, r[1:]))))
PY
echo "-- bench line of the same command without the profiler --" >> $REPO/gpurun_out/at3p_summary.txt
python $REPO/tools/at3p_bench.py 2>/dev/null | tail -1 >> $REPO/gpurun_out/at3p_summary.txt
cat $REPO/gpurun_out/at3p_summary.txt
