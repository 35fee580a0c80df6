#!/usr/bin/env python3
"""Per-function register / scratch budget from hipcc's -Rpass-analysis=kernel-resource-usage remarks (kernels AND the
device functions they call): tools/remarks_table.py remarks.txt [regex] [top]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "."
top = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rows = []
for b in re.split(r"(?=remark: [^\n]*Function Name:)", txt):
    m = re.search(r"Function Name: (\S+)", b)
    if not m:
        continue
    g = lambda k: int((re.search(k + r": (\d+)", b) or [0, 0])[1])
    rows.append((g(r"ScratchSize \[bytes/lane\]"), g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g(r"Occupancy \[waves/SIMD\]"), m.group(1)))
dem = subprocess.run(["c++filt"] + [r[5] for r in rows], capture_output=True, text=True).stdout.split("\n")
out = []
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", re.sub(r"bbs::", "", d))
    if re.search(pat, d):
        out.append((r, d))
for r, d in sorted(out, reverse=True)[:top]:
    print("scratch=%5d vgpr=%3d agpr=%3d spill=%3d occ=%d  %s" % (r[0], r[1], r[2], r[3], r[4], d[:130]))
