#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin or file): one line per kernel."""
import re, sys, subprocess
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
cur = None; rows = []
for line in txt.splitlines():
    m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (.*?) \[-R", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
for r in rows:
    try:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        name = r["name"]
    name = re.sub(r"\(.*", "", name).replace("pcpx::(anonymous namespace)::", "")
    print(f"{name:40s} sgpr={r.get('TotalSGPRs'):>4} vgpr={r.get('VGPRs'):>4} scratch={r.get('ScratchSize'):>4} occ={r.get('Occupancy'):>2} spillV={r.get('VGPRs Spill')} lds={r.get('LDS Size')}")
