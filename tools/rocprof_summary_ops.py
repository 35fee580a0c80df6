#!/usr/bin/env python3
"""Condense what tools/run_profile_ops.sh left under gpurun_out/<tag>/ (per operation: four rocprofv3 --pmc passes over
tools/prof_ops.py) into  <dst>_pmc_ops.csv : op,kernel,counter,mean_per_launch  -- the mean over the last launches of every
kernel the operation's job launched in its measured runs (kernels of the set-up, launched once, are left out).
usage: tools/rocprof_summary_ops.py gpurun_out/r05_o profiles/r05_o [runs=6]"""
import csv
import glob
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 6


def short(name):
    return name.replace("void rt::k_stage<bbs::", "").split(",")[0].replace("bbs::", "").strip('"')


rows = []
for op in ("sign", "verify", "proof_gen"):
    vals = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(src, op + "_p[0-9]*", "*_counter_collection.csv"))):
        seen = set()
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                vals[(k, r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                if (k, r["Dispatch_Id"]) not in seen:
                    seen.add((k, r["Dispatch_Id"]))
                    vals[(k, "duration_ns")].append((int(r["Dispatch_Id"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    for (k, c), v in sorted(vals.items()):
        if k.startswith("__amd") or "Tab" in k or len(v) < runs:
            continue
        v.sort()
        last = [x for _, x in v[-runs:]]
        rows.append((op, k, c, sum(last) / len(last)))
with open(dst + "_pmc_ops.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --pmc <set> (4 separate passes per operation, tools/run_profile_ops.sh) -- python3 tools/prof_ops.py <op> %d\n" % runs)
    o.write("# one resident 4096-item BLS12-381 job (L = 32, R = 8, 20-bit windows, throughput form) at a time; mean of its last %d launches.\n" % runs)
    o.write("# FETCH_SIZE / WRITE_SIZE are KiB as reported (FETCH_SIZE x2 on gfx950 for wide reads); duration_ns from the dispatch timestamps of the same passes\n")
    o.write("op,kernel,counter,mean_per_launch\n")
    for r in rows:
        o.write("%s,%s,%s,%g\n" % r)
print("wrote", dst + "_pmc_ops.csv", len(rows), "rows")
