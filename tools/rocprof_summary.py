#!/usr/bin/env python3
"""Condense what tools/run_profile.sh left under gpurun_out/<tag>/ into the small files kept under profiles/.
usage: tools/rocprof_summary.py gpurun_out/r01_e profiles/r01_e
  -> profiles/r01_e_kernel_stats.csv   (rocprofv3 --kernel-trace --stats of the default bench command)
     profiles/r01_e_pmc.csv            (mean per launch of every counter, last 3 launches of each kernel)
     profiles/r01_e_bench_inflight{1,8}.json, profiles/r01_e_pytest_gpu.log
and prints the constants bench.py carries (VALU wave-instructions and HBM bytes per launch)."""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]


def short(name):
    return name.replace("void rt::k_stage<bbs::", "").split(",")[0].replace("bbs::", "").strip('"')


with open(os.path.join(src, "stats", "stats_kernel_stats.csv")) as f, open(dst + "_kernel_stats.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras  (MI355X, 8 batches of 4096 in flight)\n")
    o.write("kernel,calls,total_ns,avg_ns,percent,min_ns,max_ns\n")
    for r in csv.DictReader(f):
        o.write('"%s",%s,%s,%d,%s,%s,%s\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                                              r["Percentage"], r["MinNs"], r["MaxNs"]))

vals = defaultdict(list)
for path in sorted(glob.glob(os.path.join(src, "p*", "*_counter_collection.csv"))):
    with open(path) as f:
        for r in csv.DictReader(f):
            vals[(short(r["Kernel_Name"]), r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
mean = {}
for k, v in vals.items():
    v.sort()
    last = [x for _, x in v[-3:]]
    mean[k] = sum(last) / len(last)
with open(dst + "_pmc.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --pmc <set> (4 separate passes, tools/run_profile.sh) -- python3 bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras\n")
    o.write("# mean of the last 3 launches of each kernel (one 4096-item batch, BLS12-381, L=32, R=8).\n")
    o.write("# SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles; FETCH_SIZE / WRITE_SIZE are KiB as reported (FETCH_SIZE x2 on gfx950 for wide reads)\n")
    o.write("kernel,counter,mean_per_launch\n")
    for (k, c) in sorted(mean):
        if k.startswith("__amd") or "TabEntry" in k or "TabWin" in k:
            continue
        o.write("%s,%s,%g\n" % (k, c, mean[(k, c)]))

for a, b in (("bench.json", "_bench_inflight8.json"), ("bench_inflight1.json", "_bench_inflight1.json"), ("pytest_gpu.log", "_pytest_gpu.log")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), dst + b)

print("bench.py constants:")
for k in sorted({k for k, _ in mean}):
    if ("SQ_INSTS_VALU" in {c for kk, c in mean if kk == k}) and not k.startswith("__amd") and "Tab" not in k:
        print("  %-28s valu=%.4g fetch_KiB=%.6g write_KiB=%.6g  active_valu/wave_cycles=%.2f wait_any/wave_cycles=%.2f" % (
            k, mean[(k, "SQ_INSTS_VALU")], mean.get((k, "FETCH_SIZE"), 0), mean.get((k, "WRITE_SIZE"), 0),
            mean.get((k, "SQ_ACTIVE_INST_VALU"), 0) / max(mean.get((k, "SQ_WAVE_CYCLES"), 1), 1),
            mean.get((k, "SQ_WAIT_ANY"), 0) / max(mean.get((k, "SQ_WAVE_CYCLES"), 1), 1)))
