#!/bin/bash
# register / scratch / occupancy report of the patch kernels (hipcc's own resource report); usage: tools/regs.sh [name filter] [extra flags...]
f=${1:-k_patch_lean}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I include -I voronoirt_amd/csrc "$@" \
  -c voronoirt_amd/csrc/vrt_patch.hip -o /tmp/p.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
f = sys.argv[1]; name = None; row = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        if name and f in name: print(name, row)
        name, row = m.group(1), {}
    for k in ("TotalSGPRs", "VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]"):
        m = re.search(r" " + re.escape(k) + r": (\d+)", line)
        if m: row[k.split()[0]] = int(m.group(1))
    if "error" in line: print(line, end="")
if name and f in name: print(name, row)
' "$f"
