#!/usr/bin/env python3
"""profiles/<tag>_gemm_pmc.txt from gpurun_out/prof_<tag>/gemm_pmc (tests/tools/collect_gemm_pmc.sh): per GEMM kernel
MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), plus the wave-time split
(GRBM_GUI_ACTIVE is the sum over the 8 XCDs: divided by 8 and by the dispatch duration it gives the shader clock, which
is printed as the cross-check).  The MFMA counter
counts matrix-pipe cycles per SIMD (64 per v_mfma_f32_32x32x2_f32), so it is cross-checked against the instruction count
the shape implies (2*M*N*K / 4096 flop per MFMA x 64)."""
import collections
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
db = os.path.join(ROOT, "gpurun_out", "prof_" + tag, "gemm_pmc", "gemm_pmc_results.db")
con = sqlite3.connect(db)
rows = con.execute("select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value, duration from "
                   "counters_collection where kernel_name like '%gemm_f32_%kernel%'").fetchall()
disp = collections.OrderedDict()
for did, k, grid, wg, c, v, dur in rows:
    disp.setdefault(did, {"kernel": k, "grid": grid, "wg": wg, "dur_ns": float(dur)})[c] = disp.get(did, {}).get(c, 0.0) + float(v)
shapes = [("NN rec 4096x4096x1024", 4096, 4096, 1024), ("NT dh 4096x1024x4096", 4096, 1024, 4096), ("TN dW 1024x4096x65536", 1024, 4096, 65536),
          ("NT rec 4096x4096x1024 (LDS-DMA 256x128x16, 8 waves)", 4096, 4096, 1024), ("TN 4096x4096x4096 (LDS-DMA k-major tiles)", 4096, 4096, 4096)]
lines = ["# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY",
         "# tests/tools/gemm_pmc_probe.py (3 launches per shape; the LAST launch of each shape is listed), MI355X, counters only",
         "# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); expected = 2*M*N*K/4096 MFMAs * 64 cycles",
         "# counter runs are slower than plain runs (the dispatches are serialised and instrumented): compare busy %, not ms"]
groups = collections.OrderedDict()
for did, d in disp.items():
    groups.setdefault((d["kernel"].split("(")[0][-90:], d["grid"]), []).append(d)
for i, ((name, grid), ds) in enumerate(groups.items()):
    d = ds[-1]
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    act = d.get("GRBM_GUI_ACTIVE", 0.0)
    label, exp = ("?", 0.0)
    if i < len(shapes):
        label = shapes[i][0]
        exp = 2.0 * shapes[i][1] * shapes[i][2] * shapes[i][3] / 4096 * 64
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    lines.append(f"{label}: grid={grid} wg={d['wg']} kernel=...{name[-60:]}")
    ghz = act / 8 / d["dur_ns"] if d["dur_ns"] else 0.0
    lines.append(f"    MFMA_BUSY={busy:.4g} (expected {exp:.4g}, ratio {busy / exp if exp else 0:.3f})  GUI_ACTIVE={act:.4g}  "
                 f"duration {d['dur_ns'] / 1e3:.1f} us -> clock {ghz:.2f} GHz  mfma_busy={busy / (act / 8 * 1024) * 100 if act else 0:.1f} %")
    if wc:
        lines.append(f"    wave time: parked (SQ_WAIT_ANY) {d.get('SQ_WAIT_ANY', 0) / wc * 100:.1f} %, issue-stalled (SQ_WAIT_INST_ANY) "
                     f"{d.get('SQ_WAIT_INST_ANY', 0) / wc * 100:.1f} %, issuing (SQ_ACTIVE_INST_ANY) {d.get('SQ_ACTIVE_INST_ANY', 0) / wc * 100:.1f} %")
open(os.path.join(ROOT, "profiles", f"{tag}_gemm_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
