#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of the headline bench (gpurun_out/prof_<tag>/{trace,fetch,write}, collected by
tests/tools/collect_profiles.sh) into the committed
summaries under profiles/: kernel stats of `python bench.py`, per-launch HBM traffic from the PMC passes (FETCH_SIZE
and WRITE_SIZE collected in separate runs; FETCH_SIZE x2 per MI355X_MICROARCH.md, checked against the calibration copy
inside the same run), and profiles/gae_traffic.json which bench.py reads for roofline.traffic."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
out = os.path.join(ROOT, "profiles")
import sqlite3


def kernel_stats():
    """Per-kernel statistics of the --kernel-trace run: rocprofv3 of ROCm 7.2 writes a rocpd SQLite database by default
    (older builds a trace_kernel_stats.csv, which is copied as is)."""
    csv_path = os.path.join(src, "trace", "trace_kernel_stats.csv")
    dst = os.path.join(out, f"{tag}_gae_bench_kernel_stats.csv")
    if os.path.exists(csv_path):
        shutil.copy(csv_path, dst)
        return
    con = sqlite3.connect(os.path.join(src, "trace", "trace_results.db"))
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(dst, "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
        for name, calls, tot, avg, mn, mx in rows:
            short = name if len(name) <= 200 else name[:197] + "..."
            f.write(f'"{short}",{calls},{tot},{avg:.3f},{100.0 * tot / total:.2f},{mn},{mx}\n')


def counter_rows(name):
    csv_path = os.path.join(src, name, f"{name}_counter_collection.csv")
    if os.path.exists(csv_path):
        for r in csv.DictReader(open(csv_path)):
            yield r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])
        return
    con = sqlite3.connect(os.path.join(src, name, f"{name}_results.db"))
    for k, c, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
        yield k, c, float(v)


kernel_stats()
agg = collections.defaultdict(list)
full = {}
for name in ("fetch", "write"):
    for k, cname, val in counter_rows(name):
        short = ("gae_fwd_kernel" if "gae_fwd_" in k else "gae_bwd_kernel" if "gae_bwd_" in k
                 else "copyBuffer(calibration)" if "copyBuffer" in k else None)
        if short:
            agg[(short, cname)].append(val)
            full[short] = k.split("(")[0][-60:] if "gae" in k else short
T, B = 1024, 65536
alg = 12 * T * B + 4 * B
res = {k: sum(v) / len(v) for k, v in agg.items()}
cal = res[("copyBuffer(calibration)", "FETCH_SIZE")] * 1024 / (T * B * 4)
lines = ["# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tests/tools/pmc_probe.py, T=1024 B=65536, MI355X",
         "# counters are KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of a",
         f"# wide coalesced streaming read -> x2; the calibration copy of {T*B*4} B in this run reads FETCH_SIZE = {cal:.4f}x,",
         "# WRITE_SIZE is exact on it.", "kernel,instantiation,counter,mean_KiB_per_launch,launches"]
for (s, c), v in sorted(agg.items()):
    lines.append(f"{s},{full[s]},{c},{sum(v)/len(v):.1f},{len(v)}")
traffic = {}
for k in ("gae_fwd_kernel", "gae_bwd_kernel"):
    tb = (2 * res[(k, "FETCH_SIZE")] + res[(k, "WRITE_SIZE")]) * 1024
    traffic[k] = tb
    lines.append(f"# {k}: HBM traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 = {tb:.0f} B per launch; algorithmic {alg} B; ratio {tb/alg:.4f}")
open(os.path.join(out, f"{tag}_gae_pmc_traffic.csv"), "w").write("\n".join(lines) + "\n")
cfg_path = os.path.join(src, "gae_config.json")
config = json.load(open(cfg_path)) if os.path.exists(cfg_path) else None
json.dump({"T": T, "B": B, **traffic, "config": config, "source": f"profiles/{tag}_gae_pmc_traffic.csv (rocprofv3 --pmc, FETCH_SIZE x2 per MI355X_MICROARCH.md)"},
          open(os.path.join(out, "gae_traffic.json"), "w"), indent=1)
prof_log = os.path.join(src, "trace.log")
if os.path.exists(prof_log):   # the bench line printed by the PROFILED command itself (its live kernel timings agree with
    for line in open(prof_log):   # the trace; un-profiled the same binary runs at a higher clock)
        if line.startswith('{"metric"'):
            open(os.path.join(out, f"{tag}_bench_under_rocprof.json"), "w").write(line)
bench = os.path.join(ROOT, "gpurun_out", "bench.log")
if os.path.exists(bench):
    open(os.path.join(out, f"{tag}_bench_n1.json"), "w").write(open(bench).read().strip().splitlines()[-1] + "\n")
print("\n".join(lines))
