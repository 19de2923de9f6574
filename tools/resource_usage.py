#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: make -C nerfshop_amd/csrc resource-usage 2>&1 | python tools/resource_usage.py [filter]"""
import re
import subprocess
import sys

flt = sys.argv[1] if len(sys.argv) > 1 else ""
rows, cur = [], None
for ln in sys.stdin:
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for k in ("VGPRs", "AGPRs", "ScratchSize", "Occupancy", "SGPRs Spill", "VGPRs Spill", "LDS Size"):
        m = re.search(r"remark:\s+" + re.escape(k) + r"(?: \[[^\]]*\])?: (\d+)", ln)
        if m and k not in cur:
            cur[k] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(nrs::DeviceModel.*", "", n).replace("void nrs::", "")
    if flt and flt not in n:
        continue
    print(f"{n:78s} VGPR {r.get('VGPRs', '?'):>3} scratch {r.get('ScratchSize', '?'):>4} occ {r.get('Occupancy', '?')} sgpr-spill {r.get('SGPRs Spill', '?'):>3} vgpr-spill {r.get('VGPRs Spill', '?'):>3} lds {r.get('LDS Size', '?')}")
