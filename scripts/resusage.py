#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (build log) as a table."""
import re, subprocess, sys
log = open(sys.argv[1]).read()
rows = []
cur = None
for line in log.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void bazmusic::", "")
        cur = {"name": name}
        rows.append(cur)
        continue
    for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print("%-60s %5s %5s %5s %8s %4s %6s" % ("kernel", "sgpr", "vgpr", "agpr", "scratch", "occ", "lds"))
for r in rows:
    print("%-60s %5s %5s %5s %8s %4s %6s" % (r["name"][:60], r.get("sgpr"), r.get("vgpr"), r.get("agpr"), r.get("scratch"), r.get("occ"), r.get("lds")))
