#!/usr/bin/env python3
"""Static instruction mix of every kernel of the ATRAC3 translation unit -> profiles/valu_mix.json.

Why: the arithmetic contract forbids fused multiply-add, so the kernels are bound by vector-instruction ISSUE, and an
issue floor needs a price per instruction. tools/ubench/valu_issue (profiles/r05_ubench_valu_issue.txt; four and eight
wavefronts per SIMD by construction) measured two classes: a 32-bit encoded instruction (VOP1 / VOP2 / VOPC: v_mul_f32,
v_add_f32, v_add_u32 ...) occupies a SIMD for 2.15 cycles, a 64-bit encoded one (VOP3: v_fma_f32, v_and_or_b32, a compare into
an SGPR pair, v_readlane ...; VOP3P: v_pk_mul_f32 / v_pk_add_f32, which carry two operations) for 3.8 - 4.2. rocprofv3's
SQ_INSTS_VALU counts both alike; this script reads the share of each class (and of packed / f64 instructions) off the
compiler's assembly of the same sources and flags. The kernels' hot parts are unrolled straight-line code, so the static
share stands in for the dynamic one (stated in the JSON).

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
            stats[cur] = {"valu": 0, "packed_f32": 0, "f64": 0, "wide": 0}
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
            # 64-bit encodings: everything that is not a plain _e32 form (VOP3 / VOP3P, and the DPP / SDWA forms, which append a
            # second dword to the 32-bit encoding)
            if not op.endswith("_e32") or re.search(r"\b(dpp|sdwa|quad_perm|row_\w+|dst_sel|src0_sel)\b", ln) or "dpp" in op or "sdwa" in op:
                stats[cur]["wide"] += 1
    out = {}
    for k, v in stats.items():
        if v["valu"] == 0:
            continue
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
        name = name.replace("void ", "").replace("int ", "").replace("at3::", "")
        v["packed_share"] = round(v["packed_f32"] / v["valu"], 4)
        v["wide_share"] = round(v["wide"] / v["valu"], 4)
        out[name] = v
    res = {"source": "hipcc " + " ".join(FLAGS) + " -S atracdenc_amd/csrc/at3hip.hip: vector instructions per kernel in the compiler's assembly",
           "note": "static counts; the share of packed instructions stands in for the dynamic share (unrolled straight-line hot loops)",
           "cycles_per_wave_instruction_per_simd": {"encoded_32_bit": 2.15, "encoded_64_bit": 4.2, "clock_ghz": 2.4,
                                                    "source": "tools/ubench/valu_issue.hip, profiles/r05_ubench_valu_issue.txt (residency by construction; SIMD's own busy span)"},
           "kernels": out}
    path = os.path.join(ROOT, "profiles", "valu_mix.json")
    json.dump(res, open(path, "w"), indent=1)
    for k, v in out.items():
        print(f"{k:32s} valu {v['valu']:6d} 64-bit encoded {v['wide']:5d} ({v['wide_share']:.2f}) packed {v['packed_f32']:5d} ({v['packed_share']:.2f}) f64 {v['f64']}")
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
