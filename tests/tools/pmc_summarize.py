#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: mean counter value per kernel name (last launch of each group)."""
import csv
import collections
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if "gemm" not in k and (len(sys.argv) < 3 or sys.argv[2] not in k):
        continue
    acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print(f"    {c:32s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
