#!/usr/bin/env python3
"""Kernel timeline of a few LSTM steps from a rocprofv3 --kernel-trace database (rocpd SQLite): start / end / queue of
every launch, to see which kernels actually overlap.   Usage: lstm_timeline.py <results.db> [first] [count]"""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
print("columns:", cols)
pick = [c for c in ("name", "start", "end", "duration", "queue_id", "stream_id", "grid_size", "workgroup_size") if c in cols]
rows = con.execute(f"select {', '.join(pick)} from kernels order by start").fetchall()
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = rows[first][pick.index("start")]
for r in rows[first:first + count]:
    d = dict(zip(pick, r))
    nm = d["name"].split("(")[0].split("::")[-1][:44]
    print(f"{(d['start'] - t0) / 1e3:10.1f} .. {(d['end'] - t0) / 1e3:10.1f} us  q={d.get('queue_id')} s={d.get('stream_id')} "
          f"grid={d.get('grid_size')} {nm}")
