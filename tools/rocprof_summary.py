#!/usr/bin/env python3
"""rocprofv3 results (.db from `rocprofv3 --kernel-trace --stats`) -> a small CSV summary for profiles/.
usage: tools/rocprof_summary.py gpurun_out/prof_x/x_results.db profiles/r01_x_kernel_stats.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w") as f:
    f.write("kernel,calls,total_us,avg_us,percent\n")
    for name, calls, tot, avg, pct in rows:
        short = name.replace("void rt::k_stage<bbs::", "").split(",")[0].replace("bbs::", "")
        f.write('"%s",%d,%.3f,%.3f,%.2f\n' % (short, calls, tot / 1e3 if tot > 1e6 else tot, avg / 1e3 if tot > 1e6 else avg, pct))
print(open(sys.argv[2]).read())
