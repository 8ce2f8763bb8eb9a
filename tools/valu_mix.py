#!/usr/bin/env python3
"""Static instruction mix of every kernel of the ATRAC3 translation unit -> profiles/valu_mix.json.

Why: the arithmetic contract forbids fused multiply-add, so the kernels are bound by vector-instruction ISSUE, and an
issue floor needs a price per instruction. tools/ubench measured (profiles/r02_ubench_instruction_rates.txt) 1.29 ns per
plain fp32 wave-instruction and SIMD, 2.37 ns per packed one (v_pk_mul_f32 / v_pk_add_f32 carry two operations but
occupy the pipe nearly twice as long). rocprofv3's SQ_INSTS_VALU counts both kinds alike; this script reads the
share of packed (and f64) instructions off the compiler's assembly of the same sources and flags. The kernels' hot
parts are unrolled straight-line code, so the static share stands in for the dynamic one (stated in the JSON).

usage: tools/valu_mix.py            (needs hipcc; no GPU)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"]


def main():
    src = os.path.join(ROOT, "atracdenc_amd", "csrc", "at3hip.hip")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "at3hip.s")
        subprocess.check_call(["hipcc", *FLAGS, "-S", "--cuda-device-only", "-o", asm, src], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    cur, stats = None, {}
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            stats[cur] = {"valu": 0, "packed_f32": 0, "f64": 0}
            continue
        if cur is None:
            continue
        if ".Lfunc_end" in ln:
            cur = None
            continue
        m = re.match(r"^\s+(v_\w+)", ln)
        if m:
            op = m.group(1)
            stats[cur]["valu"] += 1
            if op.startswith("v_pk_") and "f32" in op:
                stats[cur]["packed_f32"] += 1
            if "f64" in op:
                stats[cur]["f64"] += 1
    out = {}
    for k, v in stats.items():
        if v["valu"] == 0:
            continue
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
        name = name.replace("void ", "").replace("int ", "").replace("at3::", "")
        v["packed_share"] = round(v["packed_f32"] / v["valu"], 4)
        out[name] = v
    res = {"source": "hipcc " + " ".join(FLAGS) + " -S atracdenc_amd/csrc/at3hip.hip: vector instructions per kernel in the compiler's assembly",
           "note": "static counts; the share of packed instructions stands in for the dynamic share (unrolled straight-line hot loops)",
           "ns_per_wave_instruction_per_simd": {"plain": 1.29, "packed_f32": 2.37, "source": "tools/ubench/valu_lds_rates.hip, profiles/r02_ubench_instruction_rates.txt"},
           "kernels": out}
    path = os.path.join(ROOT, "profiles", "valu_mix.json")
    json.dump(res, open(path, "w"), indent=1)
    for k, v in out.items():
        print(f"{k:32s} valu {v['valu']:6d} packed {v['packed_f32']:5d} ({v['packed_share']:.2f}) f64 {v['f64']}")
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
