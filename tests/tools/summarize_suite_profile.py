#!/usr/bin/env python3
"""Per-kernel statistics (calls, total / average / min / max duration) of a rocprofv3 --kernel-trace rocpd database
as a CSV in the format of rocprofv3's own kernel_stats.   Usage: summarize_suite_profile.py <results.db> <out.csv>"""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                   "group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1
with open(sys.argv[2], "w") as f:
    f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for name, calls, tot, avg, mn, mx in rows:
        short = name if len(name) <= 200 else name[:197] + "..."
        f.write(f'"{short}",{calls},{tot},{avg:.3f},{100.0 * tot / total:.2f},{mn},{mx}\n')
print(f"{len(rows)} kernels, {total / 1e6:.1f} ms of kernel time -> {sys.argv[2]}")
