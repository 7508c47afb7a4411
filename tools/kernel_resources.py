#!/usr/bin/env python
"""Table of the per-kernel resource remarks hipcc prints with -Rpass-analysis=kernel-resource-usage
(usage: hipcc ... -c file.hip -Rpass-analysis=kernel-resource-usage 2> res.txt; python tools/kernel_resources.py res.txt [filter])."""
import re
import subprocess
import sys

rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/(?:lane|block)\])?: (\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print(f"{name[:110]:110s} VGPR {r.get('VGPRs', '?'):>3} AGPR {r.get('AGPRs', '?'):>3} scratch {r.get('ScratchSize', '?'):>4} "
          f"spill V{r.get('VGPRs Spill', '?')}/S{r.get('SGPRs Spill', '?')} occ {r.get('Occupancy', '?')} LDS {r.get('LDS Size', '?')}")
