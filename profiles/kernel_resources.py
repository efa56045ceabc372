#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python profiles/kernel_resources.py attention16.hip [substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act3d-chained-diffuser_amd"))
import build as B  # noqa: E402

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc"] + B.FLAGS + B.EXTRA_FLAGS.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, src), "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print("%-70s %5s %5s %5s %4s %7s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "occ", "LDS", "spill"))
for r in rows:
    if flt in r["name"]:
        print("%-70s %5d %5d %5d %4d %7d %6d" % (r["name"][-70:], r.get("vgpr", -1), r.get("agpr", -1), r.get("sgpr", -1), r.get("occ", -1),
                                                  r.get("lds", -1), r.get("spill", 0) + r.get("scratch", 0)))
