#!/usr/bin/env python3
"""Lists the loops of every device function that contain scratch accesses or calls (development aid).
usage: hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S bbs_sign_amd/csrc/tu_pv_bls.hip -o /tmp/pv.s
       tools/scan_loops.py /tmp/pv.s [min_loop_len]
A loop-carried value that lives in scratch shows up as a cluster of scratch_load at the loop head and scratch_store at
its end (DESIGN.md 5 rule 7b)."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
min_len = int(sys.argv[2]) if len(sys.argv) > 2 else 200
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z[\w]+:", l)]
for a in starts:
    e = next((i for i in range(a, len(lines)) if lines[i].strip().startswith(".Lfunc_end")), len(lines))
    body = lines[a:e]
    labels = {}
    for k, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    for k, l in enumerate(body):
        m = re.match(r"\s*(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] < k:
            seg = body[labels[m.group(2)]:k]
            sc = sum(1 for x in seg if x.split() and x.split()[0].startswith("scratch_"))
            calls = sum(1 for x in seg if x.split() and x.split()[0].startswith("s_swappc"))
            if (sc or calls) and len(seg) >= min_len:
                print("%-90s loop of %6d lines: %3d scratch ops, %2d calls" % (lines[a][:-1][:90], len(seg), sc, calls))
