#!/usr/bin/env python3
"""Which frames make a kernel's scratch: the compiler's own `.private_seg_size` expressions of the device assembly
(own frame + max over callees), resolved and printed as a tree per kernel.
usage: hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S bbs_sign_amd/csrc/tu_pv_bls.hip -o /tmp/pv.s
       tools/frames.py /tmp/pv.s [kernel regex] [depth]"""
import re
import subprocess
import sys

own, callees = {}, {}
for line in open(sys.argv[1]):
    m = re.match(r"\s*\.set\s+(\S+)\.private_seg_size,\s*(\d+)(?:\+max\((.*)\))?\s*$", line)
    if not m:
        continue
    own[m.group(1)] = int(m.group(2))
    callees[m.group(1)] = [c.strip()[:-len(".private_seg_size")] for c in (m.group(3) or "").split(",") if c.strip()]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_stage|k_pip"
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
names = list(own)
dem = dict(zip(names, subprocess.run(["c++filt"] + [n[2:] if n.startswith(".L_Z") else n for n in names],
                                     capture_output=True, text=True).stdout.split("\n")))


def total(f, seen=()):
    if f in seen or f not in own:
        return 0
    return own[f] + max([total(c, seen + (f,)) for c in callees[f]] or [0])


def show(f, ind, d):
    nm = re.sub(r"\(.*", "", dem.get(f, f)).replace("bbs::", "")
    print("%s%5d own %5d total  %s" % ("  " * ind, own.get(f, 0), total(f), nm[:110]))
    if d > 0:
        for c in sorted(callees.get(f, []), key=total, reverse=True):
            if total(c) > 0:
                show(c, ind + 1, d - 1)


for f in sorted(own, key=total, reverse=True):
    if re.search(pat, dem.get(f, f)) and not f.startswith(".L"):
        show(f, 0, depth)
