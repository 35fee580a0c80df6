#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace CSV: per kernel name the number of launches, mean / max duration, and per
hardware queue the busy share of the traced interval.  usage: tools/trace_summary.py <kernel_trace.csv> [last_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)


def short(n):
    return n.replace("void rt::k_stage<bbs::", "").replace("void bbs::", "").split("<")[0].split("(")[0]


by = defaultdict(list)
q = defaultdict(int)
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    by[short(r["Kernel_Name"])].append(d)
    q[r.get("Queue_Id", "?")] += d
print("interval %.2f ms, %d dispatches, %d queues" % ((t1 - t0) / 1e6, len(rows), len(q)))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s n=%5d mean %8.3f ms  max %8.3f ms  sum %9.2f ms" % (k, len(v), sum(v) / len(v) / 1e6, max(v) / 1e6, sum(v) / 1e6))
for k, v in sorted(q.items()):
    print("queue %s busy %.0f %%" % (k, 100.0 * v / (t1 - t0)))
