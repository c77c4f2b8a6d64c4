#!/usr/bin/env python3
"""(round 6: with a third argument, also writes {kernel: {fetch_mb, write_mb, launches}} as JSON -- what bench_suite.py quotes as measured traffic)
HBM traffic per kernel from two rocprofv3 --pmc passes (rocpd SQLite): FETCH_SIZE and WRITE_SIZE are KiB per dispatch,
summed over the counter's instances.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of a
wide coalesced streaming read -> x2, calibrated in the same run on the device copy of a known size.  The LARGEST launch
of every kernel is listed.   Usage: summarize_suite_pmc.py <fetch.db> <write.db>"""
import collections
import re
import sqlite3
import sys


def load(db, counter):
    con = sqlite3.connect(db)
    per = collections.defaultdict(float)
    names = {}
    for did, k, c, v in con.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
        if c == counter:
            per[did] += float(v)
            names[did] = k
    best = {}
    count = collections.Counter()
    for did, v in per.items():
        k = names[did]
        k = re.sub(r"hpc_rll::\(anonymous namespace\)::|hpc_rll::|void ", "", k).split("(")[0]
        k = k if len(k) <= 66 else k[:63] + "..."
        count[k] += 1
        best[k] = max(best.get(k, 0.0), v)
    return best, count


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
cal = [v for k, v in fetch.items() if "copyBuffer" in k]
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tests/tools/pmc_suite.py, MI355X")
print("# columns: kernel | 2 x FETCH_SIZE in MB (gfx950 correction) | WRITE_SIZE in MB | launches (largest launch shown)")
if cal:
    print(f"# calibration: the device copy of 2147.5 MB reads 2 x FETCH_SIZE = {2 * max(cal) * 1024 / 1e6:.1f} MB")
for k in sorted(set(fetch) | set(write)):
    if k.startswith("at::") or "elementwise" in k or "distribution" in k:
        continue
    print(f"{k:68s} {2 * fetch.get(k, 0.0) * 1024 / 1e6:10.1f} {write.get(k, 0.0) * 1024 / 1e6:10.1f} {nf.get(k, 0)}")

if len(sys.argv) > 3:
    import json
    rec = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tests/tools/pmc_suite.py; FETCH_SIZE x 2 (gfx950 correction, "
                     "MI355X_MICROARCH.md), calibrated in the same run on a device copy of 2147.5 MB",
           "calibration_copy_fetch_mb": (2 * max(cal) * 1024 / 1e6) if cal else None, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith("at::") or "elementwise" in k or "distribution" in k or not k:
            continue
        rec["kernels"][k] = {"fetch_mb": 2 * fetch.get(k, 0.0) * 1024 / 1e6, "write_mb": write.get(k, 0.0) * 1024 / 1e6, "launches": nf.get(k, 0)}
    json.dump(rec, open(sys.argv[3], "w"), indent=1)
