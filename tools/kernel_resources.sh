#!/bin/bash
# VGPR / spill / scratch / LDS / occupancy of every kernel in libcpnative (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
# usage: tools/kernel_resources.sh [filter-regex]
cd /tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared /root/repo/contrastiveprosthetics_amd/csrc/api.hip -o /tmp/_res.so \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys, subprocess
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark: (.*) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    if flt and not re.search(flt, name): continue
    print("%-70s VGPR %3s AGPR %3s spill %3s scratch %5s LDS %6s occ %s" % (name[:70], v.get("VGPRs"), v.get("AGPRs"), v.get("VGPRs Spill"), v.get("ScratchSize [bytes/lane]"), v.get("LDS Size [bytes/block]"), v.get("Occupancy [waves/SIMD]")))
' "$1"
