#!/bin/bash
# usage: tools/kres2.sh point-cloud-processing_amd/csrc/pcpx_query.hip [pattern]  -- per-kernel register/scratch/occupancy summary
f=$1; pat=${2:-.}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -I/root/repo/include -c "$f" -o /tmp/kres2.o -Rpass-analysis=kernel-resource-usage > /tmp/kres2.log 2>&1
python3 - "$pat" <<'PY'
import re,subprocess,sys
pat=sys.argv[1]
txt=open('/tmp/kres2.log').read()
if 'error' in txt: print(txt[:3000])
rows=[];cur=None
for line in txt.splitlines():
    m=re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (.*?) \[-R", line)
    if not m: continue
    k,v=m.groups()
    if k=="Function Name":
        name=subprocess.run(["c++filt",v],capture_output=True,text=True).stdout.strip()
        name=re.sub(r"\(anonymous namespace\)::","",name)
        name=name[5:] if name.startswith("void ") else name
        cur=[name.split("(")[0]];rows.append(cur)
    else: cur.append(k.split(" [")[0].split()[0][:7]+"="+v)
for r in rows:
    if re.search(pat,r[0]): print(r[0][:50].ljust(50)," ".join(r[1:]))
PY
