#!/bin/bash
# kernel_resources.sh <file.hip> [extra hipcc flags]: VGPRs / spills / scratch / occupancy / LDS of every kernel in a translation unit
# (hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel)
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -c "$f" -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  python3 -c '
import sys, re, subprocess
rows = []; cur = None
for line in sys.stdin:
    m = re.search(r"remark: [^ ]* +(?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None: cur[m.group(1)] = int(m.group(2))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    print("%-90s vgpr %3d  sgpr %3d  spill %3d  scratch %4d  occ %d  lds %6d" % (name[:90], r.get("VGPRs", -1), r.get("SGPRs", -1), r.get("VGPRs Spill", -1),
          r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))
'
rm -f /tmp/kr_$$.o
