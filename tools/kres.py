#!/usr/bin/env python3
"""Print a per-kernel resource table (VGPR/AGPR/spill/scratch/LDS/occupancy) for a .hip file."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                      "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur); rows[cur] = {}
        continue
    m = re.search(r"remark: .*?\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    print(f"{k[:70]:70s} vgpr={v.get('VGPRs')} agpr={v.get('AGPRs')} spill={v.get('VGPRs Spill')} scratch={v.get('ScratchSize [bytes/lane]')} occ={v.get('Occupancy [waves/SIMD]')} lds={v.get('LDS Size [bytes/block]')} sgpr={v.get('SGPRs')}")
