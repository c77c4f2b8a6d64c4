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
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(out, f"{tag}_gae_bench_kernel_stats.csv"))
agg = collections.defaultdict(list)
full = {}
for name in ("fetch", "write"):
    for r in csv.DictReader(open(os.path.join(src, name, f"{name}_counter_collection.csv"))):
        k = r["Kernel_Name"]
        short = ("gae_fwd_kernel" if "gae_fwd_kernel" in k else "gae_bwd_kernel" if "gae_bwd_kernel" in k
                 else "copyBuffer(calibration)" if "copyBuffer" in k else None)
        if short:
            agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
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
json.dump({"T": T, "B": B, **traffic, "source": f"profiles/{tag}_gae_pmc_traffic.csv (rocprofv3 --pmc, FETCH_SIZE x2 per MI355X_MICROARCH.md)"},
          open(os.path.join(out, "gae_traffic.json"), "w"), indent=1)
bench = os.path.join(ROOT, "gpurun_out", "bench.log")
if os.path.exists(bench):
    open(os.path.join(out, f"{tag}_bench_n1.json"), "w").write(open(bench).read().strip().splitlines()[-1] + "\n")
print("\n".join(lines))
