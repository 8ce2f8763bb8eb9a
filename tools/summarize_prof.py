#!/usr/bin/env python3
"""Condense rocprofv3 rocpd databases (kernel stats + PMC passes) into a short text summary for profiles/.
usage: summarize_prof.py <dir with stats/ pmc_fetch/ pmc_write/ pmc_sq/>"""
import glob, os, sqlite3, sys

root = sys.argv[1]

def dbs(sub):
    return sorted(glob.glob(os.path.join(root, sub, "**", "*.db"), recursive=True))

def short(n):
    n = n.split("(")[0]
    return n.replace("void ", "").replace("at3::", "")[:48]

for f in dbs("stats"):
    db = sqlite3.connect(f)
    print("== kernel stats: rocprofv3 --kernel-trace --stats (top_kernels view) ==")
    print(f"{'kernel':48s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    for name, calls, total, avg in rows:
        print(f"{short(name):48s} {calls:6d} {total/1e3:12.1f} {avg/1e3:10.2f} {100.0*total/tot:6.2f}")
    print("-- resources --")
    for r in db.execute("select distinct name, workgroup_x, grid_x, lds_size, scratch_size, vgpr_count, sgpr_count from kernels group by name"):
        print(f"{short(r[0]):48s} wg={r[1]} grid={r[2]} lds={r[3]} scratch={r[4]} vgpr={r[5]} sgpr={r[6]}")

for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in dbs(sub):
        db = sqlite3.connect(f)
        print(f"== {sub}: per-dispatch average counter values ==")
        q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
             "group by kernel_name, counter_name order by kernel_name, counter_name")
        for k, c, v, n in db.execute(q):
            print(f"{short(k):48s} {c:24s} avg={v:.6g} (n={n})")
