#!/bin/bash
# Registers, scratch and LDS of every kernel of one translation unit (default: the ATRAC3 one), as the compiler reports them.
# usage: tools/kernel_resources.sh [file.hip] [extra hipcc flags]
cd "$(dirname "$0")/.."
SRC=${1:-atracdenc_amd/csrc/at3hip.hip}
shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -c "$SRC" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys
cur = None
rows = {}
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs Spill|SGPRs Spill|TotalSGPRs|VGPRs|AGPRs|ScratchSize|Occupancy|LDS Size)[^:]*: (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
import subprocess
print("%-44s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "occ"))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0][:44]
    print("%-44s %5d %5d %7d %6d %4d" % (name, v.get("VGPRs", 0), v.get("TotalSGPRs", 0), v.get("ScratchSize", 0), v.get("LDS Size", 0), v.get("Occupancy", 0)))
'
