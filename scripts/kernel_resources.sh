#!/bin/bash
# usage: scripts/kernel_resources.sh gorse_amd/csrc/<file>.hip -- per-kernel VGPR/SGPR/scratch/occupancy table (cross-compiled, no GPU)
f=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics \
  -I"$(dirname "$0")/../include" -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re,sys
cur=None
for line in sys.stdin:
    m=re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m: continue
    t=m.group(1)
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur=t.split(": ",1)[1][:70].ljust(72)
    else:
        k,v=t.split(":")
        k=k.strip()
        if k in ("VGPRs","AGPRs","TotalSGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","VGPRs Spill","LDS Size [bytes/block]"):
            cur+=" %s=%s"%(k.split(" ")[0]+("Spill" if "Spill" in k else ""),v.strip())
if cur: print(cur)
'
