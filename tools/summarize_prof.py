#!/usr/bin/env python3
"""Condense rocprofv3 rocpd databases (kernel stats + PMC passes) into a short text summary for profiles/.
usage: summarize_prof.py <dir with stats/ pmc_fetch/ pmc_write/ pmc_sq/>"""
import glob, hashlib, os, sqlite3, sys

root = sys.argv[1]


def kernel_source_sha16():
    """Hash of the kernel sources the profiled library was built from: bench.py compares it with the tree it runs in and marks
    figures it derives from a profile of OTHER sources as stale."""
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "atracdenc_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hpp", ".hip", ".cpp", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


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

# ---- HBM traffic of the fused QMF+MDCT kernel -> k1_traffic.json (read by bench.py for roofline.traffic) ----
import json

def avg_counter(sub, counter, like):
    for f in dbs(sub):
        db = sqlite3.connect(f)
        row = db.execute("select kernel_name, avg(value) from counters_collection where counter_name=? and kernel_name like ? "
                         "group by kernel_name", (counter, like)).fetchone()
        if row:
            return row[0], row[1]
    return None, None

# the QMF + MDCT work: two kernels around the gain analysis (k_qmf_sub8 + k_mdct_sub), or the fused k_qmf_mdct8
parts = []
for like in ("%k_qmf_sub8%", "%k_mdct_sub%", "%k_qmf_mdct8%"):
    kn, fetch = avg_counter("pmc_fetch", "FETCH_SIZE", like)
    _, write = avg_counter("pmc_write", "WRITE_SIZE", like)
    if fetch is not None and write is not None:
        parts.append({"kernel": short(kn), "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB": write,
                      "bytes_per_launch": (2.0 * fetch + write) * 1024.0})
if parts:
    algo = 16384 * 4096
    total = sum(q["bytes_per_launch"] for q in parts)
    out = {
        "kernels": parts, "workload": "64 streams x 64 frames (4096 frames per step), run lengths automatic",
        "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section)",
        "bytes_per_launch": total, "algorithmic_bytes_per_launch": algo, "ratio": total / algo,
        "note": "bytes_per_launch = the kernels' sum: with gain control the subbands cross HBM between the two kernels (8 KB written + 8 KB "
                "read per frame on top of the 16 KB of PCM in and spectra out) because the gain analysis needs them there anyway",
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/profile_gpu.sh)",
        "kernel_source_sha16": kernel_source_sha16(),
    }
    json.dump(out, open(os.path.join(root, "k1_traffic.json"), "w"), indent=1)
    print("== k1 traffic ==", json.dumps(out))

# ---- whole-pipeline traffic per frame and the vector-instruction counts of the QMF + MDCT kernels -> pipeline_traffic.json ----
def per_kernel(sub, counter):
    res = {}
    for f in dbs(sub):
        db = sqlite3.connect(f)
        for k, v, n in db.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name", (counter,)):
            res[short(k)] = v
    return res

fetch, write, valu = per_kernel("pmc_fetch", "FETCH_SIZE"), per_kernel("pmc_write", "WRITE_SIZE"), per_kernel("pmc_sq", "SQ_INSTS_VALU")
ours = [k for k in fetch if k.startswith("k_") and k in write]
if ours:
    frames = 4096
    # launches per step (k_state_update runs twice per step)
    rows = {k: {"FETCH_SIZE_KiB_raw": fetch[k], "WRITE_SIZE_KiB": write[k], "bytes_per_launch": (2.0 * fetch[k] + write[k]) * 1024.0} for k in sorted(ours)}
    mult = {k: (2 if k.startswith("k_state_update") else 1) for k in rows}
    total = sum(rows[k]["bytes_per_launch"] * mult[k] for k in rows)
    out = {"workload": "64 streams x 64 frames (4096 frames per step), synchronous steps", "kernels": rows, "kernel_source_sha16": kernel_source_sha16(),
           "bytes_per_step": total, "pipeline_bytes_per_frame": total / frames, "algorithmic_bytes_per_frame": 8192 + 384,
           "correction": "FETCH_SIZE x2 + WRITE_SIZE per kernel (MI355X_MICROARCH.md, HBM section), separate --pmc passes",
           "valu_wave_insts_per_launch": {k: v for k, v in valu.items() if k.startswith(("k_qmf", "k_mdct_sub"))},
           # every kernel of the pipeline, for the whole-step issue floor (bench.py prices them with profiles/valu_mix.json)
           "valu_wave_insts_per_launch_all": {k: v * mult.get(k, 1) for k, v in sorted(valu.items()) if k.startswith("k_")}}
    json.dump(out, open(os.path.join(root, "pipeline_traffic.json"), "w"), indent=1)
    print("== pipeline traffic == %.1f KB per frame (%.1f MB per step)" % (total / frames / 1024.0, total / 1e6))
