#!/usr/bin/env python3
"""Condense rocprofv3 outputs (kernel stats + PMC passes) into a short text summary for profiles/."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]

def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))

print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("*kernel_stats.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows:
        name = r.get("Name", "")[:70]
        print(f"{name:70s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")

for tag in ("pmc_fetch", "pmc_write", "pmc_sq"):
    files = find(f"{tag}/**/*counter_collection.csv") or glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "")[:60]
                agg[k][r.get("Counter_Name")] += float(r.get("Counter_Value", 0) or 0)
                cnt[k].add(r.get("Dispatch_Id"))
    if agg:
        print(f"== {tag}: per-dispatch averages ==")
        for k, d in sorted(agg.items()):
            n = max(1, len(cnt[k]))
            print(f"{k:60s} dispatches={n} " + " ".join(f"{c}={v / n:.4g}" for c, v in sorted(d.items())))
