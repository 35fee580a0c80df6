#!/usr/bin/env python3
"""Condense what tools/run_profile.sh left under gpurun_out/<tag>/ into the small files kept under profiles/.
usage: tools/rocprof_summary.py gpurun_out/r02_c profiles/r02_c
  -> <dst>_kernel_stats.csv       rocprofv3 --kernel-trace --stats of the default bench command (8 batches in flight)
     <dst>_pmc.csv                mean per launch of every counter over the last launches of each kernel, one batch at a
                                  time (p* passes); plus duration_ns from the dispatch timestamps of the same passes
     <dst>_pmc_inflight8.csv      the same from the q* passes (bench --inflight 8)
     <dst>_ubench_valu_int.csv    tools/ubench/valu_int output, if present
then run tools/isa_histogram.py and tools/valu_model.py <dst>."""
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
    o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 64  (MI355X, the default number of distinct 4096-item batches in flight -- 6 since round 3 --, host-inclusive loop)\n")
    o.write("kernel,calls,total_ns,avg_ns,percent,min_ns,max_ns\n")
    for r in csv.DictReader(f):
        o.write('"%s",%s,%s,%d,%s,%s,%s\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                                              r["Percentage"], r["MinNs"], r["MaxNs"]))


def summarise(prefix, out, header):
    vals = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(src, prefix + "[0-9]*", "*_counter_collection.csv"))):
        seen = set()
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                vals[(k, r["Counter_Name"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                if (k, r["Dispatch_Id"]) not in seen:
                    seen.add((k, r["Dispatch_Id"]))
                    vals[(k, "duration_ns")].append((int(r["Dispatch_Id"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    mean = {}
    for k, v in vals.items():
        v.sort()
        last = [x for _, x in v[-6:]]
        mean[k] = sum(last) / len(last)
    with open(out, "w") as o:
        o.write(header)
        o.write("# SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles; FETCH_SIZE / WRITE_SIZE are KiB as reported (FETCH_SIZE x2 on gfx950 for wide reads);\n")
        o.write("# duration_ns = End_Timestamp - Start_Timestamp of the dispatch in the same passes (rocprofv3 serialises dispatches while it collects counters)\n")
        o.write("kernel,counter,mean_per_launch\n")
        for (k, c) in sorted(mean):
            if k.startswith("__amd") or "TabEntry" in k or "TabWin" in k:
                continue
            o.write("%s,%s,%g\n" % (k, c, mean[(k, c)]))
    return mean


mean = summarise("p", dst + "_pmc.csv",
                 "# rocprofv3 --kernel-trace --pmc <set> (5 separate passes, tools/run_profile.sh) -- python3 bench.py --steps 4 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras\n"
                 "# mean of the last 6 launches of each kernel (one 4096-item batch, BLS12-381, L=32, R=8).\n")
if glob.glob(os.path.join(src, "q[0-9]*")):
    summarise("q", dst + "_pmc_inflight8.csv",
              "# rocprofv3 --kernel-trace --pmc <set> (3 separate passes) -- python3 bench.py --steps 24 --warmup 8 --inflight 8 --no-cpu-baseline --no-extras\n"
              "# mean of the last 6 launches of each kernel.  Counter collection serialises the dispatches: these equal the one-at-a-time values.\n")
for a, b in (("ubench_valu_int.csv", "_ubench_valu_int.csv"), ("bench_under_rocprof.json", "_bench_under_rocprof.json")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), dst + b)
for k in sorted({k for k, _ in mean}):
    if ("SQ_INSTS_VALU" in {c for kk, c in mean if kk == k}) and not k.startswith("__amd") and "Tab" not in k:
        print("  %-28s valu=%.4g fetch_KiB=%.6g write_KiB=%.6g wait_any/wave_cycles=%.3f dur=%.3f ms" % (
            k, mean[(k, "SQ_INSTS_VALU")], mean.get((k, "FETCH_SIZE"), 0), mean.get((k, "WRITE_SIZE"), 0),
            mean.get((k, "SQ_WAIT_ANY"), 0) / max(mean.get((k, "SQ_WAVE_CYCLES"), 1), 1), mean.get((k, "duration_ns"), 0) / 1e6))
