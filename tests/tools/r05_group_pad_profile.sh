#!/bin/bash
# Round 5: kernel trace of the grouped packed pad of 2^20 rows (tests/tools/r03_group_pad_time.py) -> per-kernel averages
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
python "$REPO/tests/tools/r03_group_pad_time.py" 2>&1 | grep grouped
(export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$OUT/gp_prof" -o trace -- python "$REPO/tests/tools/r03_group_pad_time.py" > "$OUT/r05_group_pad_under_rocprof.txt" 2>&1)
find "$OUT/gp_prof" -type f | head
DB=$(find "$OUT/gp_prof" -name "*.db" | head -1)
python - "$DB" <<'P'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t == 'kernels'] or [t for t in tabs if t.startswith('kernels')]
print("tables:", kt[:3])
rows = con.execute(f"select name, count(*), avg(end-start), min(end-start) from {kt[0]} group by name order by sum(end-start) desc").fetchall()
for n, c, a, m in rows[:16]:
    print(f"{c:5d} calls  avg {a/1e3:9.2f} us  min {m/1e3:9.2f} us  {n[:90]}")
P
rm -rf "$OUT/gp_prof"
