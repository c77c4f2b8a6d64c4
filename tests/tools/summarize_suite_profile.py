#!/usr/bin/env python3
"""Per-kernel statistics (calls, total / average / min / max duration) of a rocprofv3 --kernel-trace rocpd database
as a CSV in the format of rocprofv3's own kernel_stats -- one row per (kernel, GRID SIZE): a tool that launches the same
kernel at several shapes (warm-up shapes, pre-roll, the measured shape) gets one row per shape, so an average can be
turned into a roofline fraction (the categorical kernels' averages mixed shapes).
Usage: summarize_suite_profile.py <results.db> <out.csv>"""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
print("columns of `kernels`:", cols)


def dims(stem):
    """SQL expression for the product of the x / y / z extents named <stem>[_x|_y|_z] (whatever subset exists)."""
    have = [c for c in (f"{stem}_x", f"{stem}_y", f"{stem}_z") if c in cols]
    if have:
        return "(" + " * ".join(f"max({c}, 1)" if False else c for c in have) + ")"
    return stem if stem in cols else None


grid = dims("grid_size") or dims("grid")
wg = dims("workgroup_size") or dims("workgroup")
sel = "name" + (f", {grid}" if grid else ", 0") + (f", {wg}" if wg else ", 0")
rows = con.execute(f"select {sel}, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                   f"group by {sel} order by sum(duration) desc").fetchall()
total = sum(r[4] for r in rows) or 1
with open(sys.argv[2], "w") as f:
    f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","GridSize","WorkgroupSize"\n')
    for name, g, w, calls, tot, avg, mn, mx in rows:
        short = name if len(name) <= 200 else name[:197] + "..."
        f.write(f'"{short}",{calls},{tot},{avg:.3f},{100.0 * tot / total:.2f},{mn},{mx},{g},{w}\n')
print(f"{len(rows)} (kernel, grid) rows, {total / 1e6:.1f} ms of kernel time -> {sys.argv[2]}")
